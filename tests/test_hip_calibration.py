"""dv_model_calibrate (include/dvhip.h, deepvariant_amd/csrc/calib.hip) on the GPU: the native two-pipeline
walk against its torch restatement (tests/calib_emulation.py), determinism, and what it buys against the fp32
oracle on pileups it was NOT calibrated on."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(shape, n, seed):
  from tests import cnn_tail as T
  if shape == (100, 221, 7):
    return T.illumina_pileups_gpu(n, seed=seed, chunk=4096)
  return T.longread_images_gpu('hifi' if shape[2] == 10 else 'ont', n, seed=seed)


def _model(shape, weights, max_batch):
  from deepvariant_amd.inception_v3 import InceptionV3
  m = InceptionV3(shape, max_batch=max_batch)
  m.load_flat_weights(weights)
  return m


@pytest.mark.parametrize('shape', [(100, 221, 7), (100, 147, 10)])
def test_native_corrections_equal_the_torch_restatement(shape):
  from tests import calib_emulation as E
  from oracle import inception_ref as R
  ref = R.make_random_model(shape[2], seed=23)
  x = _images(shape, 48, seed=515)
  m = _model(shape, ref.export_flat(), 64)
  got = m.calibrate(x)
  torch.set_num_threads(min(64, os.cpu_count() or 1))
  want = E.corrections(ref, x.cpu())
  assert got.shape == want.shape
  rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
  print('%s: |corr| rms %.3g, max %.3g; native vs torch restatement: relative L2 difference %.3g, max abs %.3g' % (
      shape, float(np.sqrt((want ** 2).mean())), float(np.abs(want).max()), rel, float(np.abs(got - want).max())))
  off = 0
  rows = []
  for i, cb in enumerate(ref.convs):       # where the two walks part: per-layer relative difference
    c = cb.conv.out_channels
    g, w = got[off:off + c], want[off:off + c]
    rows.append((float(np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-12)), i, c, float(np.abs(w).max())))
    off += c
  print('  per layer (relative L2, layer, couts, max |corr|): first 6 %s' % (
      ' '.join('%d:%.3f' % (i, r) for r, i, _, _ in rows[:6])))
  print('  worst 8: %s' % ' '.join('%d(c%d):%.3f' % (i, c, r) for r, i, c, _ in sorted(rows, reverse=True)[:8]))
  assert np.isfinite(got).all()
  # The two walks round the same numbers in different summation orders, so individual fp16 roundings differ and
  # the difference of two nearly equal means is noisy where a layer has few samples: measured (48 images)
  # 0.0-0.1 % on the stem, 3-4 % at 35x35, ~20 % on the 1x5 maps of mixed8-10 (240 samples per channel),
  # 12-14 % over the whole vector.  Bars: the stem layers (millions of samples) to 1 %, everything to 25 %,
  # and the two vectors pointing the same way.
  stem = sum(cb.conv.out_channels for cb in ref.convs[:3])
  rel_stem = float(np.linalg.norm(got[:stem] - want[:stem]) / np.linalg.norm(want[:stem]))
  cos = float(np.dot(got, want) / (np.linalg.norm(got) * np.linalg.norm(want)))
  assert rel_stem <= 0.01, rel_stem
  assert rel <= 0.25, rel
  assert cos >= 0.97, cos


def test_calibration_is_deterministic_and_replaces_the_previous_one():
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  ref = R.make_random_model(7, seed=31)
  x = _images(shape, 96, seed=616)
  m = _model(shape, ref.export_flat(), 96)
  p0 = m(x).cpu().numpy()
  a = m.calibrate(x[:64])
  p1 = m(x).cpu().numpy()
  b = m.calibrate(x[:64])
  p2 = m(x).cpu().numpy()
  assert np.array_equal(a, b)
  assert np.array_equal(p1, p2)          # a second calibration starts from the loaded shifts again
  assert not np.array_equal(p0, p1)
  assert np.abs(p0 - p1).max() < 5e-3


@pytest.mark.parametrize('seed', [17, 404])
def test_calibrated_model_is_closer_to_the_fp32_oracle_on_other_pileups(seed):
  from tests import cnn_tail as T
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  n = 4096
  ref = R.make_random_model(7, seed=seed)
  ref_gpu = R.make_random_model(7, seed=seed).cuda()
  x = _images(shape, n, seed=7000 + seed)
  want = T.oracle_probs_gpu(ref_gpu, x)
  old = os.environ.get('DV_SPLIT_FROM')
  os.environ['DV_SPLIT_FROM'] = '94'     # plain fp16 weights in every layer
  try:
    m = _model(shape, ref.export_flat(), n)
  finally:
    os.environ.pop('DV_SPLIT_FROM', None)
    if old is not None:
      os.environ['DV_SPLIT_FROM'] = old
  before = T.tail_stats(m(x).cpu().numpy(), want)
  m.calibrate(_images(shape, 256, seed=880000 + seed))
  after = T.tail_stats(m(x).cpu().numpy(), want)
  print('seed %d uncalibrated: %s' % (seed, T.fmt(before)))
  print('seed %d calibrated:   %s' % (seed, T.fmt(after)))
  assert after['mean_abs_dp'] <= 0.9 * before['mean_abs_dp'], (before, after)
  assert after['max_abs_dp'] <= 1e-3, after


def test_calibrate_argument_errors():
  import ctypes as C
  from deepvariant_amd import _lib
  from deepvariant_amd.inception_v3 import InceptionV3
  m = InceptionV3((100, 221, 7), max_batch=8)
  x = torch.zeros((4, 100, 221, 7), dtype=torch.uint8, device='cuda')
  with pytest.raises(ValueError):
    m.calibrate(x)                        # no weights yet
  w = np.zeros(m.num_params, np.float32)
  rc = _lib.lib().dv_model_calibrate(m._handle, w.ctypes.data, w.size, x.data_ptr(), 4, None, 0)
  assert rc == _lib.DV_ERR_INVALID_ARGUMENT and b'load weights first' in _lib.lib().dv_last_error()
  m.init_random(seed=3)
  rc = _lib.lib().dv_model_calibrate(m._handle, w.ctypes.data, w.size - 1, x.data_ptr(), 4, None, 0)
  assert rc == _lib.DV_ERR_INVALID_ARGUMENT
  with pytest.raises(ValueError):
    m.calibrate(x.cpu())


def test_checkpoint_calibration_is_the_manual_calibration_on_the_fixed_set():
  """What call_variants / make_examples do after loading weights (--calibration_examples): calibrate on
  calibration_set.draw(shape) -- the same images whenever and wherever it is drawn."""
  from deepvariant_amd import calibration_set
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  ref = R.make_random_model(7, seed=57)
  x = _images(shape, 320, seed=818)
  cal = calibration_set.draw(shape, 256)
  assert cal.shape == (256,) + shape and cal.dtype == torch.uint8
  assert torch.equal(cal, calibration_set.draw(shape, 256))          # deterministic
  assert torch.equal(cal[:64], calibration_set.draw(shape, 64))      # and prefix-stable
  assert not torch.equal(cal[:64].cpu(), x[:64].cpu())
  manual = _model(shape, ref.export_flat(), 320)
  plain = manual(x).cpu().numpy()
  manual.calibrate(cal)
  want = manual(x).cpu().numpy()
  auto = _model(shape, ref.export_flat(), 320)
  corr = auto.calibrate_for_checkpoint(256)
  assert auto.calibration == {'images': 256, 'set_version': calibration_set.SET_VERSION, 'cached': False}
  assert np.array_equal(auto(x).cpu().numpy(), want)
  assert np.array_equal(auto(x[:40]).cpu().numpy(), want[:40])       # small forwards are calibrated too
  assert not np.array_equal(want, plain)
  off = _model(shape, ref.export_flat(), 320)
  assert off.calibrate_for_checkpoint(0) is None
  assert np.array_equal(off(x).cpu().numpy(), plain)
  assert corr.size == sum(co for _, _, _, co, _ in auto.layer_table())


def test_probabilities_are_a_pure_function_of_checkpoint_and_image():
  """The reference's call_variants maps (weights, image) -> probabilities (deepvariant/call_variants.py:904-932).
  The same 512 examples (a) in one batch, (b) shuffled, in batches of 100, (c) split between two models that each
  loaded and calibrated for themselves: identical bits per example."""
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  flat = R.make_random_model(7, seed=59).export_flat()
  x = _images(shape, 512, seed=4242)
  a = _model(shape, flat, 512)
  a.calibrate_for_checkpoint(256)
  one = a(x).cpu().numpy()
  perm = torch.randperm(512, generator=torch.Generator().manual_seed(5)).to(x.device)
  got = np.zeros_like(one)
  for i in range(0, 512, 100):
    idx = perm[i:i + 100]
    got[idx.cpu().numpy()] = a(x[idx]).cpu().numpy()
  assert np.array_equal(got, one)
  b = _model(shape, flat, 256)
  b.calibrate_for_checkpoint(256)
  c = _model(shape, flat, 300)
  c.calibrate_for_checkpoint(256)
  halves = np.concatenate([b(x[:256]).cpu().numpy(), c(x[256:]).cpu().numpy()])
  assert np.array_equal(halves, one)


@pytest.mark.parametrize('shape', [(100, 199, 9), (100, 147, 10), (100, 147, 8), (100, 221, 6), (75, 75, 4)])
def test_calibration_set_of_every_supported_shape(shape):
  from deepvariant_amd import calibration_set
  x = calibration_set.draw(shape, 48)
  assert x.shape == (48,) + shape and x.dtype == torch.uint8 and x.is_cuda
  assert torch.equal(x, calibration_set.draw(shape, 48))
  assert int((x.reshape(48, -1).max(1).values > 0).sum()) == 48         # every image holds a pile-up
  assert calibration_set.draw((100, 221, 12), 8) is None                # no set: the model stays plain fp16


def test_applied_corrections_equal_the_calibration_that_measured_them(tmp_path, monkeypatch):
  """dv_model_apply_corrections, and the cache of a checkpoint's corrections next to it."""
  from oracle import inception_ref as R
  shape = (100, 221, 7)
  ref = R.make_random_model(7, seed=58)
  x = _images(shape, 300, seed=919)
  a = _model(shape, ref.export_flat(), 300)
  corr = a.calibrate(x[:256])
  want = a(x).cpu().numpy()
  b = _model(shape, ref.export_flat(), 300)
  plain = b(x).cpu().numpy()
  b.apply_corrections(corr)
  assert np.array_equal(b(x).cpu().numpy(), want) and not np.array_equal(plain, want)
  with pytest.raises(Exception):
    b.apply_corrections(corr[:-1])
  # the cache next to a checkpoint: named by everything the corrections depend on; the second model applies the file
  prefix = str(tmp_path / 'ckpt.f32')
  first = _model(shape, ref.export_flat(), 300)
  c1 = first.calibrate_for_checkpoint(128, cache_prefix=prefix)
  files = [f for f in os.listdir(tmp_path) if '.dvcal-' in f]
  assert len(files) == 1 and first.calibration['cached'] is False
  second = _model(shape, ref.export_flat(), 300)
  c2 = second.calibrate_for_checkpoint(128, cache_prefix=prefix)
  assert second.calibration['cached'] is True and np.array_equal(c1, c2)
  assert np.array_equal(first(x).cpu().numpy(), second(x).cpu().numpy())
  other = _model(shape, R.make_random_model(7, seed=60).export_flat(), 300)      # other weights: another file
  other.calibrate_for_checkpoint(128, cache_prefix=prefix)
  assert other.calibration['cached'] is False and len([f for f in os.listdir(tmp_path) if '.dvcal-' in f]) == 2
  (tmp_path / files[0]).write_bytes(b'\0' * 10)                                   # a damaged file is measured again
  third = _model(shape, ref.export_flat(), 300)
  assert np.array_equal(third.calibrate_for_checkpoint(128, cache_prefix=prefix), c1)
  assert third.calibration['cached'] is False
