"""chain.hip (the factorised-7x7 branches of the 17x17 blocks as fused launches, intermediates in
LDS, loader / computing waves) against the per-layer kernels (GPU): same K order, same fp16
rounding of every intermediate, skipped taps multiply zeros only -- the 2048 features and the
probabilities must be BIT-identical.  Shapes: WGS 221-wide (4x12 maps, 4 images per tile), PacBio
147-wide (4x7 maps, 6 per tile, tile not full), ONT 199-wide (4x10 maps); batch sizes that are not
multiples of the tile, smaller than one tile, and large enough for several tiles per workgroup."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _forward(shape, weights, x, chain, max_batch=None):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.get('DV_NO_CHAIN')
  if chain:
    os.environ.pop('DV_NO_CHAIN', None)
  else:
    os.environ['DV_NO_CHAIN'] = '1'
  try:
    m = InceptionV3(shape, max_batch=max_batch or x.shape[0])
    m.load_flat_weights(weights)
    probs = m(x).cpu().numpy()
    feat = m.debug_tensor(-1, x.shape[0])
  finally:
    if old is None:
      os.environ.pop('DV_NO_CHAIN', None)
    else:
      os.environ['DV_NO_CHAIN'] = old
  return probs, feat


def _images(n, shape, seed):
  h, w, c = shape
  rng = np.random.default_rng(seed)
  x = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
  x[: n // 2, 40:] = 0                     # pileup-like: zero rows below the reads
  return x


@pytest.mark.parametrize('shape,n', [((100, 221, 7), 1203), ((100, 221, 7), 3), ((100, 221, 6), 64),
                                     ((100, 147, 10), 601), ((100, 199, 9), 130), ((100, 147, 10), 5)])
def test_fused_chains_are_bit_identical_to_the_per_layer_kernels(shape, n):
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=31)
  weights = ref.export_flat()
  xd = torch.from_numpy(_images(n, shape, 11)).cuda()
  p0, f0 = _forward(shape, weights, xd, chain=False)
  p1, f1 = _forward(shape, weights, xd, chain=True)
  assert np.isfinite(p1).all()
  np.testing.assert_array_equal(f1, f0)
  np.testing.assert_array_equal(p1, p0)


def test_fused_chains_with_a_batch_smaller_than_the_model():
  """max_batch above the batch: the last tile is shifted back, images past the batch stay untouched."""
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  ref = R.make_random_model(shape[2], seed=5)
  weights = ref.export_flat()
  xd = torch.from_numpy(_images(37, shape, 3)).cuda()
  p0, f0 = _forward(shape, weights, xd, chain=False, max_batch=64)
  p1, f1 = _forward(shape, weights, xd, chain=True, max_batch=64)
  np.testing.assert_array_equal(f1, f0)
  np.testing.assert_array_equal(p1, p0)


def test_fused_chains_against_the_oracle():
  """...and within the 1e-3 bar of the fp32 restatement (oracle/inception_ref.py)."""
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  ref = R.make_random_model(shape[2], seed=13)
  weights = ref.export_flat()
  x = _images(24, shape, 17)
  p1, _ = _forward(shape, weights, torch.from_numpy(x).cuda(), chain=True)
  with torch.no_grad():
    want = ref(torch.from_numpy(x)).numpy()
  assert np.abs(p1 - want).max() <= 1e-3
