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
