"""conv_pool_resident_kernel: the stem's 3x3 80->192 with the 3x3 / stride-2 max-pool that follows it
taken inside the kernel (GPU).

Graph: tf_keras InceptionV3 stem -> MaxPooling2D(3, 2) -> mixed0 (deepvariant/keras_modeling.py:268-274,
SURVEY.md App. B).  Max-pooling commutes with shift + ReLU + fp16 rounding and the convolution keeps
conv_resident_kernel's K order, so the pooled tensor, the 2048 features and the probabilities must
be BIT-identical to the round-3 arrangement (conv -> 21x51x192 tensor -> pool on load in mixed0's
heads, DV_NO_POOL2_IN_CONV=1) -- for batches smaller than one persistent grid, for several
fragments per wave, and for map widths whose fragments straddle examples differently.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _forward(shape, weights, x, fused):
  from deepvariant_amd.inception_v3 import InceptionV3
  os.environ.pop('DV_NO_POOL2_IN_CONV', None)
  if not fused:
    os.environ['DV_NO_POOL2_IN_CONV'] = '1'
  try:
    m = InceptionV3(shape, max_batch=x.shape[0])
    m.load_flat_weights(weights)
    probs = m(x).cpu().numpy()
    stem = m.debug_tensor(-2, x.shape[0])
    feat = m.debug_tensor(-1, x.shape[0])
  finally:
    os.environ.pop('DV_NO_POOL2_IN_CONV', None)
  return probs, stem, feat


def _maxpool3s2(t):
  return torch.nn.functional.max_pool2d(torch.from_numpy(t.astype(np.float32)).permute(0, 3, 1, 2), 3, 2
                                        ).permute(0, 2, 3, 1).numpy().astype(np.float16)


@pytest.mark.parametrize('shape,n', [((100, 221, 7), 3), ((100, 221, 7), 70), ((100, 221, 7), 700),
                                     ((100, 147, 10), 90), ((100, 199, 9), 33), ((75, 75, 1), 5),
                                     ((120, 301, 5), 7)])
def test_pool_inside_the_conv_is_bit_identical(shape, n):
  from oracle import inception_ref as R
  h, w, c = shape
  ref = R.make_random_model(c, seed=31)
  weights = ref.export_flat()
  rng = np.random.default_rng(n)
  x = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
  x[: n // 2, 40:] = 0                     # pileup-like: zero rows below the reads
  xd = torch.from_numpy(x).cuda()
  p0, s0, f0 = _forward(shape, weights, xd, fused=False)
  p1, s1, f1 = _forward(shape, weights, xd, fused=True)
  # the round-3 path's last stem tensor is the unpooled conv output; pool it here (max is exact)
  assert s0.shape[1] == 2 * s1.shape[1] + 1 or s0.shape[1] == 2 * s1.shape[1] + 2, (s0.shape, s1.shape)
  np.testing.assert_array_equal(s1, _maxpool3s2(s0))
  np.testing.assert_array_equal(f1, f0)
  np.testing.assert_array_equal(p1, p0)
  with torch.no_grad():
    want = ref(torch.from_numpy(x)).numpy()
  assert np.abs(p1 - want).max() <= 2e-3   # uniform-noise images: the tolerance test_hip_stem_fused.py gives them


def test_pooled_tensor_against_the_fp32_oracle():
  """The pooled stem output itself against the oracle's stem (relative tolerance of the fp16 grid)."""
  from oracle import inception_ref as R
  n = 40
  ref = R.make_random_model(7, seed=3)
  rng = np.random.default_rng(11)
  x = rng.integers(0, 256, (n, 100, 221, 7), dtype=np.uint8)
  x[:, 45:] = 0
  _, s1, _ = _forward((100, 221, 7), ref.export_flat(), torch.from_numpy(x).cuda(), fused=True)
  pre = ((torch.from_numpy(x).float() - 128.0) / 128.0).permute(0, 3, 1, 2)
  with torch.no_grad():
    s = ref.stem
    want = R._maxpool(s[4](s[3](R._maxpool(s[2](s[1](s[0](pre))))))).permute(0, 2, 3, 1).numpy()
  assert s1.shape == want.shape == (n, 10, 25, 192)
  rel = np.abs(s1.astype(np.float32) - want).max() / np.abs(want).max()
  assert rel <= 1e-2, rel


def _forward_env(shape, weights, x, env):
  from deepvariant_amd.inception_v3 import InceptionV3
  for k in env:
    os.environ[k] = '1'
  try:
    m = InceptionV3(shape, max_batch=x.shape[0])
    m.load_flat_weights(weights)
    probs = m(x).cpu().numpy()
    feat = m.debug_tensor(-1, x.shape[0])
  finally:
    for k in env:
      os.environ.pop(k, None)
  return probs, feat


@pytest.mark.parametrize('shape,n', [((100, 221, 7), 5), ((100, 221, 7), 300), ((100, 147, 10), 64),
                                     ((100, 199, 9), 40), ((75, 75, 1), 9), ((120, 301, 5), 6)])
def test_side_max_pool_of_mixed3_is_bit_identical(shape, n):
  """mixed3's MaxPooling2D(3, 2) taken on the side by the 3x3 / stride-2 convolution of the same tensor
  (model.hip choose_side_pool: the convolution's nine tap fragments of a channel chunk ARE the pool's window)
  against the separate max-pool launch (DV_NO_SIDE_POOL=1): the maximum is exact, so the 2048 features and
  the probabilities must agree bit for bit -- for the <4,1> tile shape of small batches and the <4,2> shape
  of large ones, and for map sizes whose last pixel block is partial."""
  from oracle import inception_ref as R
  h, w, c = shape
  ref = R.make_random_model(c, seed=37)
  weights = ref.export_flat()
  x = np.random.default_rng(n + w).integers(0, 256, (n, h, w, c), dtype=np.uint8)
  x[: n // 2, 45:] = 0
  xd = torch.from_numpy(x).cuda()
  p0, f0 = _forward_env(shape, weights, xd, ['DV_NO_SIDE_POOL'])
  p1, f1 = _forward_env(shape, weights, xd, [])
  np.testing.assert_array_equal(f1, f0)
  np.testing.assert_array_equal(p1, p0)
