"""SampleOptions.use_non_uniform_downsampling on the device path: the host picks the reads (csrc/sampling.cpp), the
HIP encoder draws the shorter list -- pile-ups equal the oracle's and, where it is built, the reference's own code
(oracle/_ref), byte for byte.  tests/test_non_uniform_downsampling_cpu.py holds the sampler's own tests."""
import numpy as np
import pytest

from deepvariant_amd import dv_types as T
from tests.test_non_uniform_downsampling_cpu import _case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('threshold', [0, 2, 40])
def test_non_uniform_downsampling_on_the_device(threshold):
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  for seed in range(6):
    opts, call, ref_window, reads, image_start, combo, height = _case(7000 + 13 * threshold + seed)
    enc = PileupImageEncoderNative(opts)
    so = T.SampleOptions(pileup_height=height, use_non_uniform_downsampling=True,
                         non_uniform_downsampling_threshold=threshold)
    got = enc.build_pileup_for_one_sample(call, ref_window, reads, image_start, combo, so)
    want = O.build_pileup(opts, call, ref_window, reads, image_start, combo, pileup_height=height,
                          non_uniform_downsampling_threshold=threshold)
    np.testing.assert_array_equal(got, want, err_msg='threshold %d seed %d (oracle)' % (threshold, seed))
    if O.reference_available():
      with O.reference_backend():
        ref = O.build_pileup(opts, call, ref_window, reads, image_start, combo, pileup_height=height,
                             non_uniform_downsampling_threshold=threshold)
      np.testing.assert_array_equal(got, ref, err_msg='threshold %d seed %d (reference build)' % (threshold, seed))


def test_allele_sample_probability_on_the_device():
  """channels/allele_sample_probability_channel.cc: the host-computed pixel (alleles in key order) drawn through
  list_aux; whole pile-ups against the oracle and the reference build."""
  import zlib
  from deepvariant_amd.pileup_image_native import PileupImageEncoderNative
  from oracle import oracle as O
  from tests import fuzz_inputs as F
  name, channels, width, height, okw, ckw = F.SAMPLE_PROBABILITY_CONFIGS[0]
  opts = F.options(channels, width, height, **dict(okw))
  enc = PileupImageEncoderNative(opts)
  so = T.SampleOptions(pileup_height=height)
  rng = np.random.default_rng(zlib.crc32(name.encode()))
  for trial in range(8):
    call, ref, reads, start, combo = F.make_case(rng, width, int(rng.choice([0, 2, 10, height + 20])), **dict(ckw))
    got = enc.build_pileup_for_one_sample(call, ref, reads, start, combo, so)
    want = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height)
    np.testing.assert_array_equal(got, want, err_msg='trial %d (oracle)' % trial)
    if O.reference_available():
      with O.reference_backend():
        ref_img = O.build_pileup(opts, call, ref, reads, start, combo, pileup_height=height)
      np.testing.assert_array_equal(got, ref_img, err_msg='trial %d (reference build)' % trial)
