"""Blank-row skipping through the stem (round 6: on by default) against the dense kernels (GPU).

Tiles of conv2 / stem_b / the 3x3 80->192 whose receptive field sees only the zero rows below the pile-up are copied
from the all-blank image's response instead of being computed.  That must be invisible: the stem's tensors and the
probabilities are compared BIT FOR BIT with the dense path (DV_BLANK_SKIP=0, and the run-time switch of one model) on
encoded pileups of every depth, on images with hand-placed last rows around every threshold parity, on all-zero and
completely filled images, with a nonzero byte in the very last row, at the long-read shapes (per-layer conv1 / conv2),
at sizes that do not fill the persistent grids and at the bench's batch size.  Semantics being exploited:
deepvariant/pileup_image_native.cc:405-447 (images are zero-padded to `height` rows), deepvariant/dv_utils.py:343-366."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(shape, weights, max_batch, skip):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.pop('DV_BLANK_SKIP', None)
  if not skip:
    os.environ['DV_BLANK_SKIP'] = '0'
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
  if n_pileups:
    from deepvariant_amd import calibration_set
    drawn = calibration_set.draw(shape, n_pileups, seed=seed)
    if drawn is not None:
      imgs.extend(drawn.cpu().numpy())
  # hand-placed pile-up heights: every row count around the thresholds, 0 and the full image
  for r in list(range(0, 12)) + list(range(30, 60)) + [h - 3, h - 2, h - 1, h]:
    r = min(r, h)
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


def _expected_thresholds(x):
  n, h = x.shape[0], x.shape[1]
  nz = x.reshape(n, h, -1).any(axis=2)
  r = np.where(nz.any(axis=1), h - np.argmax(nz[:, ::-1], axis=1), 0)
  t2 = (r + 1) // 2
  t4 = (t2 + 2) // 2
  return np.stack([r, t2, t4, t4, (t4 + 1) // 2]).astype(np.int32)


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 221, 6), (100, 147, 8), (76, 199, 4), (100, 199, 9), (100, 147, 10)])
def test_blank_skip_is_bit_identical(shape):
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=23)
  w = ref.export_flat()
  x = _images(shape, 192, seed=5)
  n = x.shape[0]
  dense = _model(shape, w, n, skip=False)
  skip = _model(shape, w, n, skip=True)
  xd = torch.from_numpy(x).cuda()
  want = dense(xd).cpu().numpy()
  got = skip(xd).cpu().numpy()
  assert dense.blank_thresholds(n) is None
  thr = skip.blank_thresholds(n)
  np.testing.assert_array_equal(thr, _expected_thresholds(x))
  assert (thr[4] < 10).mean() > 0.5                  # the test does skip: most images have blank pooled rows
  # the stem's output (the pooled 3x3 80->192), where the last copies land -- every later layer reads all of it.  (The
  # tensors between the skipping kernels are NOT compared: blank tiles that no computed tile of the consumer reads
  # are not even copied there, StemAArgs::blank_need.)
  np.testing.assert_array_equal(skip.debug_tensor(-2, n), dense.debug_tensor(-2, n))
  np.testing.assert_array_equal(got, want)
  # dv_model_infer_rows: the caller states the rows used (exactly, or generously) instead of the scan
  exact = torch.from_numpy(thr[0].copy()).cuda()
  np.testing.assert_array_equal(skip(xd, rows_used=exact).cpu().numpy(), want)
  np.testing.assert_array_equal(skip.blank_thresholds(n), thr)
  np.testing.assert_array_equal(skip(xd, rows_used=torch.clamp(exact - 3, min=0), rows_add=5).cpu().numpy(), want)
  np.testing.assert_array_equal(skip(xd).cpu().numpy(), want)          # and the scan again, through the same graph
  # the run-time switch on ONE model: dense, then skipping again -- the same bits every time
  skip.set_blank_skip(False)
  np.testing.assert_array_equal(skip(xd).cpu().numpy(), want)
  assert skip.blank_thresholds(n) is None
  skip.set_blank_skip(True)
  np.testing.assert_array_equal(skip(xd).cpu().numpy(), want)
  # a second forward through the captured graph, different images in the same buffer; and a small batch that does
  # not fill the persistent grids
  xd.copy_(torch.from_numpy(x[::-1].copy()).cuda())
  np.testing.assert_array_equal(skip(xd).cpu().numpy(), dense(xd).cpu().numpy())
  np.testing.assert_array_equal(skip(xd[:7]).cpu().numpy(), dense(xd[:7]).cpu().numpy())
  # calibration moves the shifts: the blank responses are recomputed with them
  from deepvariant_amd import calibration_set
  if calibration_set.supported(shape):
    skip.calibrate_for_checkpoint(64)
    dense.calibrate_for_checkpoint(64)
    np.testing.assert_array_equal(skip(xd).cpu().numpy(), dense(xd).cpu().numpy())


def test_blank_skip_at_the_bench_batch_size():
  """8,104 ILLUMINA30 pileups per forward (the bench's step): every probability and the stem's output identical."""
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  w = R.make_random_model(7, seed=31).export_flat()
  x = T.illumina_pileups_gpu(8104, seed=77)
  skip = _model(shape, w, 8104, skip=True)
  got = skip(x).cpu().numpy()
  a = skip.debug_tensor(-2, 64)
  skip.set_blank_skip(False)
  want = skip(x).cpu().numpy()
  np.testing.assert_array_equal(skip.debug_tensor(-2, 64), a)
  np.testing.assert_array_equal(got, want)
