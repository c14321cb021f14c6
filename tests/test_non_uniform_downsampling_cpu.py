"""SampleOptions.use_non_uniform_downsampling in the product (deepvariant_amd/csrc/sampling.cpp,
packing.non_uniform_sample; reference: deepvariant/pileup_image_native.cc:242-294,326-341 and
deepvariant/sampling_util.h): every allele keeps a minimum of its supporting reads, the rest of the image is filled
from what is left.

  * the reference's own two tests (deepvariant/sampling_util_test.cc:73-152) are DISTRIBUTION tests over injected
    index providers; they are re-run here by enumerating every sequence of draws through `forced_draws`;
  * the product's pile-ups equal the oracle's and -- where it is built -- the reference's own code (oracle/_ref) with
    the option on.  The device encoder is replaced by the oracle's packed adapter here (no GPU in this suite;
    tests/test_hip_sampling.py is the device form): what is under test is which reads the host hands over.
Parity of the bit stream itself (absl::Uniform over std::mt19937_64) is UNPINNED: abseil is not in the image and no
reference test fixes a draw; product, oracle and reference build share one restatement of its published algorithm.
"""
import itertools

import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from deepvariant_amd import packing
from oracle import oracle as O
from tests import fuzz_inputs as FZ


class _Table:
  def __init__(self, n):
    self.keys = ['r%d/0' % i for i in range(n)]


def _sample(n, alleles, max_reads, min_per, draws):
  call = T.DeepVariantCall(variant=T.Variant('c', 0, 1, 'A', sorted(alleles)),
                           allele_support={a: T.SupportingReads(['r%d/0' % i for i in idx]) for a, idx in alleles.items()})
  return packing.non_uniform_sample(call, _Table(n), np.arange(n), max_reads, min_per, 1, forced_draws=draws)


def test_reservoir_sample_is_uniform():
  """SamplingUtilTest.ReservoirSampleIsUniform: 3 of {0..6}; every index provider value equally likely -> every
  3-subset equally likely.  (One partition element -- no allele lists a read -- with a minimum of 3 and an image of 3
  rows: the first reservoir is the sample, the second, of size 0, still consumes its draws.)"""
  counts = {}
  for draws in itertools.product(*[range(i + 1) for i in range(3, 7)]):
    got = _sample(7, {}, 3, 3, list(draws) + [0, 0, 0, 0])
    counts[tuple(got.tolist())] = counts.get(tuple(got.tolist()), 0) + 1
  assert len(counts) == 35 and set(counts.values()) == {840 // 35}
  assert all(len(k) == 3 and list(k) == sorted(k) for k in counts)


def test_sample_with_partition_mins_distribution():
  """SamplingUtilTest.CheckSampleWithPartitionMinsDistribution: partitions {0,1,2} | {3,4,5}, 4 of 6 with at least 1
  per partition: balanced (2 + 2) two times out of three."""
  balanced = total = 0
  first = list(itertools.product(range(2), range(3)))                  # 1 of 3: draws for indices 1, 2
  rest = list(itertools.product(range(3), range(4)))                   # 2 of the 4 left: draws for indices 2, 3
  for a, b, c in itertools.product(first, first, rest):
    got = _sample(6, {'A': [0, 1, 2]}, 4, 1, list(a) + list(b) + list(c)).tolist()
    assert len(got) == 4 and any(x < 3 for x in got) and any(x >= 3 for x in got)
    balanced += sum(x < 3 for x in got) == 2
    total += 1
  assert total == 432 and balanced * 3 == total * 2


def test_partition_rules_and_the_fall_back():
  # thresholds that cannot fit: None = the caller keeps the uniform shuffle (pileup_image_native.cc:333-337)
  assert _sample(30, {'A': range(0, 12), 'C': range(12, 24)}, 10, 6, None) is None
  # fewer reads than rows: everybody stays, whatever the draws
  assert _sample(9, {'A': [1, 2], 'C': [5]}, 25, 3, None).tolist() == list(range(9))
  # a read listed twice belongs to the allele that comes first in key order; unknown names are ignored
  call = T.DeepVariantCall(variant=T.Variant('c', 0, 1, 'A', ['C', 'G']),
                           allele_support={'G': T.SupportingReads(['r0/0', 'r1/0', 'nobody/0']), 'C': T.SupportingReads(['r1/0'])})
  got = packing.non_uniform_sample(call, _Table(40), np.arange(40), 3, 1, 7)
  assert got is not None and len(got) == 3 and 1 in got.tolist() and 0 in got.tolist()      # C = {1}, G = {0}: both stay
  # of several reads with one key only the LAST is ever found (the reference maps names to indices): the others
  # belong to no partition element and are never drawn
  class Dup:
    keys = ['a/0', 'b/0', 'a/0', 'c/0', 'b/0', 'd/0']
  call = T.DeepVariantCall(variant=T.Variant('c', 0, 1, 'A', ['C']), allele_support={'C': T.SupportingReads(['a/0'])})
  got = packing.non_uniform_sample(call, Dup, np.arange(6), 25, 1, 3)
  assert got.tolist() == [2, 3, 4, 5]
  # seeded draws are a function of the seed
  a = _sample(60, {'A': range(0, 60, 7)}, 20, 2, None).tolist()
  assert a == _sample(60, {'A': range(0, 60, 7)}, 20, 2, None).tolist() and len(a) == 20 and a == sorted(a)
  assert sum(x % 7 == 0 for x in a) >= 2


@pytest.fixture
def oracle_drawn_encoder(monkeypatch):
  """PileupImageEncoderNative with the oracle's packed adapter where the device encoder would be."""
  from deepvariant_amd import pileup_image_native as pin
  from tests.test_reference_examples_cpu import OracleDeviceEncoder
  monkeypatch.setattr(pin.PileupImageEncoderNative, '_encoder', lambda self, width: OracleDeviceEncoder(self._options))


def _case(seed, height=30):
  name, channels, width, _, okw, ckw = FZ.CONFIGS[0]
  opts = FZ.options(channels, width, height, **dict(okw))
  rng = np.random.default_rng(seed)
  depth = int(rng.choice([5, 26, 60, 140]))
  call, ref_window, reads, image_start, combo = FZ.make_case(rng, width, depth, n_alts=int(rng.integers(1, 4)), **dict(ckw))
  for k, r in enumerate(reads):
    r.fragment_name, r.read_number = 'f%03d' % k, 0
    r.alignment.mapping_quality = 60
  keys = ['%s/0' % r.fragment_name for r in reads]
  for allele in list(call.allele_support):
    pick = rng.choice(len(keys), size=int(rng.integers(0, max(len(keys) // 3, 1) + 1)), replace=False)
    call.allele_support[allele] = T.SupportingReads([keys[int(j)] for j in pick])
  return opts, call, ref_window, reads, image_start, combo, height


@pytest.mark.parametrize('threshold', [0, 1, 4, 40])
def test_product_pileups_equal_the_oracle_and_the_reference_build(oracle_drawn_encoder, threshold):
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  differs_from_uniform = 0
  for seed in range(8):
    opts, call, ref_window, reads, image_start, combo, height = _case(9000 + 31 * threshold + seed)
    enc = PileupImageEncoderNative(opts)
    so = T.SampleOptions(pileup_height=height, use_non_uniform_downsampling=True,
                         non_uniform_downsampling_threshold=threshold)
    got = enc.build_pileup_for_one_sample(call, ref_window, reads, image_start, combo, so)
    want = O.build_pileup(opts, call, ref_window, reads, image_start, combo, pileup_height=height,
                          non_uniform_downsampling_threshold=threshold)
    assert np.array_equal(got, want), (threshold, seed)
    if O.reference_available():
      with O.reference_backend():
        ref = O.build_pileup(opts, call, ref_window, reads, image_start, combo, pileup_height=height,
                             non_uniform_downsampling_threshold=threshold)
      assert np.array_equal(got, ref), (threshold, seed)
    uniform = O.build_pileup(opts, call, ref_window, reads, image_start, combo, pileup_height=height)
    differs_from_uniform += not np.array_equal(got, uniform)
  assert (differs_from_uniform >= 3) if threshold < 40 else (differs_from_uniform == 0)


def test_allele_sample_probability_pixels():
  """channels/allele_sample_probability_channel.cc: the host pixel (packing.allele_sample_probability_pixels, alleles
  in key order; it travels to the device in list_aux like the allele-frequency pixel) is what the oracle and the
  reference build draw along the read.  (tests/test_hip_sampling.py compares whole pile-ups on the device.)"""
  name, channels, width, height, okw, ckw = FZ.SAMPLE_PROBABILITY_CONFIGS[0]
  opts = FZ.options(channels, width, height, min_bq=0, min_mapq=0)
  seen = set()
  n = 0
  for seed in range(10):
    rng = np.random.default_rng(300 + seed)
    call, ref_window, reads, image_start, combo = FZ.make_case(rng, width, int(rng.choice([3, 12, 40])), **dict(ckw))
    table = packing.ReadTable.from_reads(reads)
    mine = packing.allele_sample_probability_pixels(call, table, np.arange(len(reads)))
    for i, r in enumerate(reads):
      rows = [O.encode_read(opts, call, ref_window, r, image_start, combo, None)]
      if O.reference_available():
        with O.reference_backend():
          rows.append(O.encode_read(opts, call, ref_window, r, image_start, combo, None))
      for row in rows:
        if row is None or not row[0, :, 0].any():
          continue                                         # rejected, or no base inside the window
        drawn = row[0, :, len(channels) - 1][row[0, :, 0] > 0]
        assert set(drawn.tolist()) == {int(mine[i])}, (seed, i, drawn.tolist(), int(mine[i]))
        n += 1
    seen.update(mine.tolist())
  assert n > 200 and len(seen) > 8
