"""The pooled projections' average pool in the heads launch's epilogue (model.hip choose_avg_epilogue,
conv_common.h conv_epilogue_avg) against the separate conv -> avgpool3s1_kernel path (DV_NO_AVG_EPI=1): the
same arithmetic in the same order, so every block output and the probabilities must agree BIT FOR BIT (GPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(shape, weights, max_batch, no_avg_epi):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.pop('DV_NO_AVG_EPI', None)
  if no_avg_epi:
    os.environ['DV_NO_AVG_EPI'] = '1'
  try:
    m = InceptionV3(shape, max_batch=max_batch)
  finally:
    os.environ.pop('DV_NO_AVG_EPI', None)
    if old is not None:
      os.environ['DV_NO_AVG_EPI'] = old
  m.load_flat_weights(weights)
  return m


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 147, 10), (100, 199, 9)])
@pytest.mark.parametrize('n', [3, 61, 700])
def test_epilogue_pool_is_bit_identical_to_the_separate_kernel(shape, n):
  """n = 3: one partly filled block per stage; 61: map counts that are no multiple of 5 / 51 maps per block;
  700: many blocks.  Random uint8 images exercise every map position (the borders' divisors included)."""
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=9)
  w = ref.export_flat()
  rng = np.random.default_rng(n)
  x = torch.from_numpy(rng.integers(0, 256, (n,) + shape, dtype=np.uint8)).cuda()
  fused = _model(shape, w, n, no_avg_epi=False)
  plain = _model(shape, w, n, no_avg_epi=True)
  a = fused(x).cpu().numpy()
  b = plain(x).cpu().numpy()
  fa, fb = fused.debug_tensor(-1, n), plain.debug_tensor(-1, n)
  sa, sb = fused.debug_tensor(-2, n), plain.debug_tensor(-2, n)
  assert np.array_equal(sa, sb)
  assert np.array_equal(fa, fb), float(np.abs(fa.astype(np.float32) - fb.astype(np.float32)).max())
  assert np.array_equal(a, b)
  with torch.no_grad():
    want = ref(x[:min(n, 8)].cpu()).numpy()
  assert np.abs(a[:min(n, 8)] - want).max() <= 2e-3      # both are the classifier, not merely equal
