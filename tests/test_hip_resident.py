"""conv_resident_kernel (weights resident in LDS, persistent eight-wave blocks) against
conv_mfma_kernel (GPU): same K order, same MFMA sequence, same epilogue -- the stem output and
the probabilities must be BIT-identical.  600 images per forward so that the launch is large
enough for the resident path to be chosen (it needs several tiles per wave of a full grid)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _forward(shape, weights, x, mode):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.get('DV_RESIDENT')
  os.environ['DV_RESIDENT'] = str(mode)
  os.environ['DV_NO_POOL2_IN_CONV'] = '1'   # round 4 pools inside the 3x3 80->192 (test_hip_convpool.py); this file
  try:                                      # keeps checking the plain resident kernel the round-3 way
    m = InceptionV3(shape, max_batch=x.shape[0])
    m.load_flat_weights(weights)
    probs = m(x).cpu().numpy()          # the graph is captured under this setting
    stem = m.debug_tensor(-2, x.shape[0])
    feat = m.debug_tensor(-1, x.shape[0])
  finally:
    os.environ.pop('DV_NO_POOL2_IN_CONV', None)
    if old is None:
      os.environ.pop('DV_RESIDENT', None)
    else:
      os.environ['DV_RESIDENT'] = old
  return probs, stem, feat


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 147, 8)])
def test_resident_weight_kernel_is_bit_identical(shape):
  from oracle import inception_ref as R
  h, w, c = shape
  n = 600 if w == 221 else 900
  ref = R.make_random_model(c, seed=29)
  weights = ref.export_flat()
  rng = np.random.default_rng(7)
  x = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
  x[: n // 2, 40:] = 0                     # pileup-like: zero rows below the reads
  xd = torch.from_numpy(x).cuda()
  p0, s0, f0 = _forward(shape, weights, xd, 0)   # conv_mfma_kernel everywhere
  p1, s1, f1 = _forward(shape, weights, xd, 1)   # resident 3x3 80->192
  p2, s2, f2 = _forward(shape, weights, xd, 2)   # every eligible 96-cout-tile layer
  np.testing.assert_array_equal(s1, s0)
  np.testing.assert_array_equal(p1, p0)
  np.testing.assert_array_equal(s2, s0)
  np.testing.assert_array_equal(f2, f0)
  np.testing.assert_array_equal(p2, p0)
