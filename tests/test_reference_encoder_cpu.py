"""The REFERENCE's own pileup encoder, compiled here from its sources (oracle/_ref/libdvref.so: deepvariant/
pileup_image_native.cc, pileup_channel_lib.cc and channels/*.cc, unmodified, behind oracle/dvo.h -- see
oracle/ref_build/), against

  * its own unit-test expectations: every case of tests/test_oracle_known_answers.py (lifted from
    deepvariant/pileup_image_test.py, pileup_image_native_test.cc, pileup_channel_lib_test.cc) is collected
    again in this module and runs on the reference build -- which shows the build (mini_protoc structs, abseil
    stand-ins) behaves like a real one;
  * the oracle restatement (oracle/encoder_oracle.cpp): pixel for pixel on seeded fuzz inputs over every
    channel set of tests/fuzz_inputs.py (every CIGAR operator, HP tags, methylation, fuzzy support, deep
    pile-ups through the shuffle, blanked channels, mean coverage, trimmed-read positions), read by read
    (EncodeRead) and pile-up by pile-up (BuildPileupForOneSample), and on packed batches of the three bench
    workloads through the packed adapter -- so the oracle is validated by the reference ITSELF, not only by the
    vectors its tests hold;
  * the reference's golden TFRecords (the 84 Illumina images, the PacBio rows), drawn by the reference build.

CPU only; skipped where neither /root/reference nor a prebuilt oracle/_ref/libdvref.so exists.
"""
import numpy as np
import pytest

from oracle import oracle as O

if not O.reference_available():
  pytest.skip('oracle/_ref/libdvref.so is not built and the reference tree is not here', allow_module_level=True)

from deepvariant_amd import dv_types as T      # noqa: E402
from tests import fuzz_inputs as FZ            # noqa: E402
from tests import test_oracle_known_answers as _KA_TESTS      # noqa: E402

# Every known-answer test of the oracle, collected once more here; the autouse fixture below makes this
# module's tests run on the reference build.  Two of them assert the ORACLE's wording of a fatal error.
_ORACLE_WORDING = {'test_avg_base_quality_out_of_bounds_is_fatal', 'test_unknown_cigar_op_is_fatal'}
for _name in dir(_KA_TESTS):
  if _name.startswith('test_') and _name not in _ORACLE_WORDING:
    globals()[_name] = getattr(_KA_TESTS, _name)


# The golden TFRecords drawn by the reference build: the 84 Illumina images (reference band exact, the rows the
# reference's realigner left alone), the PacBio rows, and -- with the product's window realigner, candidate
# caller and phasing in front, as in tests/test_oracle_golden.py -- ALL 84 Illumina / 401 PacBio images bit-exact.
from tests import test_oracle_golden as _GOLDEN_TESTS      # noqa: E402
golden = _GOLDEN_TESTS.golden
for _name in ('test_golden_illumina_images', 'test_golden_pacbio_rows', 'test_golden_illumina_chain_with_realigner',
              'test_golden_pacbio_chain_with_phasing'):
  globals()[_name] = getattr(_GOLDEN_TESTS, _name)


@pytest.fixture(autouse=True)
def _on_the_reference_build():
  with O.reference_backend():
    assert O.is_reference_backend()
    yield
  assert not O.is_reference_backend()


def _both(fn):
  """fn() on the reference build (active in this module) and on the oracle restatement."""
  ref = fn()
  saved = O._lib      # pylint: disable=protected-access
  try:
    O._lib = None     # pylint: disable=protected-access
    assert not O.is_reference_backend()
    mine = fn()
  finally:
    O._lib = saved    # pylint: disable=protected-access
  return ref, mine


@pytest.mark.parametrize('name,channels,width,height,okw,ckw', FZ.CONFIGS + FZ.ULTIMA_CONFIGS + FZ.SAMPLE_PROBABILITY_CONFIGS,
                         ids=[c[0] for c in FZ.CONFIGS + FZ.ULTIMA_CONFIGS + FZ.SAMPLE_PROBABILITY_CONFIGS])
def test_oracle_equals_the_reference_on_fuzz_inputs(name, channels, width, height, okw, ckw):
  opts = FZ.options(channels, width, height, **dict(okw))
  enums = [O.channel_str_to_enum(c) for c in channels]
  n_cases = 0
  for seed in range(12):
    rng = np.random.default_rng(1000 * len(name) + seed)
    depth = int(rng.choice([0, 1, 7, 30, height + 25]))      # incl. deeper than the image: the shuffle
    call, ref_window, reads, image_start, combo = FZ.make_case(rng, width, depth, **dict(ckw))
    blank = [enums[int(rng.integers(0, len(enums)))]] if seed % 4 == 3 else None
    positions = [int(r.alignment.position.position) - int(rng.integers(0, 30)) for r in reads] if seed % 5 == 4 else None
    kw = dict(pileup_height=(height if seed % 3 else 0), mean_coverage=float(rng.integers(0, 60)),
              alignment_positions=positions, channels_to_blank=blank)
    ref, mine = _both(lambda: O.build_pileup(opts, call, ref_window, reads, image_start, combo, **kw))
    assert ref.shape == mine.shape == (height, width, len(channels))
    assert np.array_equal(ref, mine), (name, seed, np.argwhere(ref != mine)[:5])
    for r in reads[:12]:
      a, b = _both(lambda: O.encode_read(opts, call, ref_window, r, image_start, combo, blank))
      assert (a is None) == (b is None)
      assert a is None or np.array_equal(a, b), (name, seed, r.fragment_name)
    n_cases += 1
  assert n_cases == 12


def test_fuzzy_support_codes_equal_the_reference():
  rng = np.random.default_rng(77)
  seen = set()
  for _ in range(60):
    call, _, reads, _, combo = FZ.make_case(rng, 61, 12, with_hp=True, fuzzy=True, n_alts=3)
    for r in reads:
      a, b = _both(lambda: O.fuzzy_read_supports_alt(call, r, combo))
      assert a == b
      seen.add(a)
  assert len(seen) >= 3      # several of 0 / 1 / 2 / 9 / 10 occur


@pytest.mark.parametrize('n,max_reads,seed', [(5, 95, 1), (96, 95, 2101079370), (300, 95, 2101079370), (1500, 95, 7)])
def test_downsample_indices_equal_the_reference(n, max_reads, seed):
  a, b = _both(lambda: O.downsample_indices(n, max_reads, seed))
  assert np.array_equal(a, b) and sorted(a.tolist()) == list(range(n))


@pytest.mark.parametrize('kind', ['illumina30', 'hifi35', 'ont50'])
def test_packed_batches_of_the_bench_workloads(kind):
  """The packed adapter (what bench.py's cpu_baseline and the full-size parity runs use) on the three bench
  workloads: oracle == reference, byte for byte."""
  from deepvariant_amd import synth
  if kind == 'illumina30':
    opts = synth.illumina_options()
    batch = synth.make_illumina_batch(48, seed=5, options=opts)
  else:
    lr = 'hifi' if kind == 'hifi35' else 'ont'
    opts = synth.longread_options(lr)
    batch = synth.make_longread_batch(32, lr, seed=5, options=opts)
  (ref, ref_rows), (mine, mine_rows) = _both(lambda: O.encode_packed(opts, batch, n_threads=2))
  assert ref.size == mine.size > 0 and np.array_equal(ref, mine)
  assert np.array_equal(ref_rows, mine_rows) and int(mine_rows.max()) > 20


@pytest.mark.parametrize('threshold', [0, 1, 3, 8, 40])
def test_non_uniform_downsampling_equals_the_reference(threshold):
  """SampleOptions.use_non_uniform_downsampling (pileup_image_native.cc:242-294,326-341; deepvariant/sampling_util.h):
  the allele partition, the per-allele minimum, the reservoir samples and the fall-back to the uniform shuffle when
  the thresholds alone exceed the image -- the oracle restatement == the reference's own code, rows and pixels.  Both
  draw through the same restatement of absl::Uniform (oracle/absl_uniform_restated.h; abseil is not in the image), so
  this pins everything around the bit stream, not the bit stream."""
  name, channels, width, height, okw, ckw = FZ.CONFIGS[0]
  height = 30
  opts = FZ.options(channels, width, height, **dict(okw))
  n_sampled = n_fallback = 0
  for seed in range(10):
    rng = np.random.default_rng(4000 + 17 * threshold + seed)
    depth = int(rng.choice([5, 24, 26, 60, 140]))
    call, ref_window, reads, image_start, combo = FZ.make_case(rng, width, depth, n_alts=int(rng.integers(1, 4)), **dict(ckw))
    for k, r in enumerate(reads):       # every read usable, distinct keys: the sample decides the image
      r.fragment_name, r.read_number = 'f%03d' % k, 0
      r.alignment.mapping_quality = 60
    keys = ['%s/0' % r.fragment_name for r in reads]
    for allele in list(call.allele_support):
      pick = rng.choice(len(keys), size=int(rng.integers(0, max(len(keys) // 3, 1) + 1)), replace=False) if keys else []
      call.allele_support[allele] = T.SupportingReads([keys[int(j)] for j in pick])
    kw = dict(pileup_height=height, non_uniform_downsampling_threshold=threshold, return_row_reads=True)
    (ref, _, _), (mine, n_mine, rows_mine) = _both(
        lambda: O.build_pileup(opts, call, ref_window, reads, image_start, combo, **kw))
    # (the reference does not report which read a row shows: the pixels say it -- random reads, no two rows alike)
    assert np.array_equal(ref, mine), (threshold, seed)
    _, (uniform, _, rows_uniform) = _both(lambda: O.build_pileup(opts, call, ref_window, reads, image_start, combo,
                                                                 pileup_height=height, return_row_reads=True))
    if depth > height - 5:
      same_as_uniform = rows_uniform.tolist() == rows_mine.tolist()
      n_fallback += same_as_uniform
      n_sampled += not same_as_uniform
  assert n_sampled + n_fallback >= 4
  assert (n_sampled >= 3) if threshold <= 3 else True
  assert (n_fallback >= 3) if threshold == 40 else True       # 40 reads per allele cannot fit 25 rows: uniform again
