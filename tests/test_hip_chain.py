"""chain.hip (the factorised-7x7 branches of the 17x17 blocks and the 3x3 / 5x5 branches of the 35x35
blocks as fused launches, intermediates in LDS, loader / computing waves) against the per-layer kernels (GPU): same K order, same fp16
rounding of every intermediate, skipped taps multiply zeros only -- the 2048 features and the
probabilities must be BIT-identical.  Shapes: WGS 221-wide (4x12 maps, 4 images per tile), PacBio
147-wide (4x7 maps, 6 per tile, tile not full), ONT 199-wide (4x10 maps); batch sizes that are not
multiples of the tile, smaller than one tile, and large enough for several tiles per workgroup."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


_KNOBS = ('DV_NO_CHAIN', 'DV_NO_CHAIN2D', 'DV_CHAIN2D_MIN_LEN')


def _forward(shape, weights, x, chain, max_batch=None):
  """chain: False = per-layer kernels; True = every fused chain; '1d' = only the 1 x k / k x 1
  chains of the small maps; '2d-singles' = also the 35x35 stage's single 3x3 / 5x5 layers (default: its 3x3 -> 3x3 pairs only)."""
  from deepvariant_amd.inception_v3 import InceptionV3
  old = {k: os.environ.pop(k, None) for k in _KNOBS}
  if chain is False:
    os.environ['DV_NO_CHAIN'] = '1'
  elif chain == '1d':
    os.environ['DV_NO_CHAIN2D'] = '1'
  elif chain == '2d-singles':
    os.environ['DV_CHAIN2D_MIN_LEN'] = '1'
  try:
    m = InceptionV3(shape, max_batch=max_batch or x.shape[0])
    m.load_flat_weights(weights)
    probs = m(x).cpu().numpy()
    feat = m.debug_tensor(-1, x.shape[0])
  finally:
    for k in _KNOBS:
      os.environ.pop(k, None)
      if old[k] is not None:
        os.environ[k] = old[k]
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
  for mode in ('1d', '2d-singles', True):
    p1, f1 = _forward(shape, weights, xd, chain=mode)
    assert np.isfinite(p1).all()
    np.testing.assert_array_equal(f1, f0, err_msg=str(mode))
    np.testing.assert_array_equal(p1, p0, err_msg=str(mode))


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
