"""Opt-in blank-row skipping (DV_BLANK_SKIP=1, HISTORY.md 7) against the default kernels (GPU).

Stem outputs whose receptive field sees only the zero rows below the pile-up are copied from
the all-blank image's response instead of being computed.  That must be invisible: the stem
output tensor and the probabilities are compared BIT FOR BIT with the default path on encoded
pileups of every depth, on images with hand-placed last rows around every threshold parity,
on all-zero and completely filled images, and with a nonzero byte in the very last row."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(shape, weights, max_batch, skip):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.pop('DV_BLANK_SKIP', None)
  if skip:
    os.environ['DV_BLANK_SKIP'] = '1'
  try:
    m = InceptionV3(shape, max_batch=max_batch)
  finally:
    os.environ.pop('DV_BLANK_SKIP', None)
    if old is not None:
      os.environ['DV_BLANK_SKIP'] = old
  m.load_flat_weights(weights)
  return m


def _images(shape, n_pileups, seed):
  h, w, c = shape
  rng = np.random.default_rng(seed)
  imgs = []
  if (h, w) == (100, 221) and c in (6, 7):
    from deepvariant_amd import synth
    from deepvariant_amd.pileup_image_native import _Encoder
    opts = synth.illumina_options(c)
    batch = synth.make_illumina_batch(n_pileups, seed=seed, options=opts)
    out, _ = _Encoder(opts, opts.width).encode(batch, c)
    imgs.extend(out.reshape(-1, h, w, c))
  # hand-placed pile-up heights: every row count around the thresholds, 0 and the full image
  for r in list(range(0, 12)) + list(range(30, 60)) + [h - 3, h - 2, h - 1, h]:
    im = np.zeros((h, w, c), np.uint8)
    im[:r] = rng.integers(0, 256, (r, w, c), dtype=np.uint8)
    imgs.append(im)
  # a single nonzero byte far below the pile-up (what the mean-coverage paint does)
  for row, col, ch in ((h - 1, w - 1, c - 1), (h - 1, 0, 0), (70, w // 2, 0), (45, 3, c - 1)):
    im = np.zeros((h, w, c), np.uint8)
    im[:20] = rng.integers(0, 256, (20, w, c), dtype=np.uint8)
    im[row, col, ch] = 1
    imgs.append(im)
  return np.stack(imgs)


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 221, 6), (100, 147, 8), (76, 199, 4)])
def test_blank_skip_is_bit_identical(shape):
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=23)
  w = ref.export_flat()
  x = _images(shape, 300, seed=5)
  n = x.shape[0]
  plain = _model(shape, w, n, skip=False)
  skip = _model(shape, w, n, skip=True)
  xd = torch.from_numpy(x).cuda()
  want = plain(xd).cpu().numpy()
  got = skip(xd).cpu().numpy()
  # the stem's last tensor (3x3 80->192 output) is where the copies land.  Round 4: the default path
  # pools that tensor inside the convolution (10 x 25), the blank-skipping path keeps the unpooled
  # one (21 x 51) for its copies -- pool it here before comparing (max is exact)
  idx = -2
  a, b = skip.debug_tensor(idx, n), plain.debug_tensor(idx, n)
  if a.shape != b.shape:
    a = torch.nn.functional.max_pool2d(torch.from_numpy(a.astype(np.float32)).permute(0, 3, 1, 2), 3, 2
                                       ).permute(0, 2, 3, 1).numpy().astype(np.float16)
  np.testing.assert_array_equal(a, b)
  np.testing.assert_array_equal(got, want)
  # a second forward through the captured graph, different images in the same buffer
  xd.copy_(torch.from_numpy(x[::-1].copy()).cuda())
  np.testing.assert_array_equal(skip(xd).cpu().numpy(), plain(xd).cpu().numpy())
