"""Fused stem kernels (deepvariant_amd/csrc/stem.hip) vs the fp32 oracle and vs the
per-layer HIP path (GPU).

The fused launches compute the same chain as the first five layers of
`oracle/inception_ref.InceptionV3.stem` (tf_keras InceptionV3 as built by
deepvariant/keras_modeling.py:268-274): fp16 operands, fp32 accumulation, fp16
hand-offs -- so against the per-layer path only the accumulation order differs
(an fp16 ulp now and then), and against the fp32 oracle the tolerance is the fp16
grid of the activations.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(n, channels, seed):
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(channels)
  k = max(1, (2 * n) // 3)
  batch = synth.make_illumina_batch(k, seed=seed, options=opts, multi_allelic=False)
  out, _ = _Encoder(opts, opts.width).encode(batch, channels)
  real = out.reshape(-1, 100, 221, channels)[:k]
  noise = np.random.default_rng(seed).integers(0, 256, (n - k, 100, 221, channels), dtype=np.uint8)
  return np.concatenate([real, noise])


def _model(shape, weights, max_batch, fused):
  from deepvariant_amd.inception_v3 import InceptionV3
  old = os.environ.pop('DV_NO_STEM_FUSE', None)
  if not fused:
    os.environ['DV_NO_STEM_FUSE'] = '1'
  try:
    m = InceptionV3(shape, max_batch=max_batch)
  finally:
    os.environ.pop('DV_NO_STEM_FUSE', None)
    if old is not None:
      os.environ['DV_NO_STEM_FUSE'] = old
  m.load_flat_weights(weights)
  # this file checks the kernels' arithmetic on whole tensors: blank-row skipping (tests/test_hip_blank_skip.py) leaves
  # the rows of conv2's / the 1x1's output that no computed tile reads unwritten, so it is switched off here
  m.set_blank_skip(False)
  return m


def _interior(t, halo):
  return t[:, halo:t.shape[1] - halo, halo:t.shape[2] - halo] if halo else t


def _report(name, got, want, atol, rtol):
  diff = np.abs(got - want)
  bad = diff > atol + rtol * np.abs(want)
  if bad.any():
    idx = np.argwhere(bad)
    lines = ['%s: %d / %d elements differ (max |d| = %g)' % (name, bad.sum(), bad.size, diff.max())]
    for axis, label in enumerate(('example', 'row', 'col', 'channel')):
      vals, counts = np.unique(idx[:, axis], return_counts=True)
      lines.append('  bad %ss: %s' % (label, ', '.join(
          '%d(x%d)' % (v, c) for v, c in list(zip(vals, counts))[:24])))
    for i in idx[:8]:
      lines.append('  at %s got %g want %g' % (tuple(i), got[tuple(i)], want[tuple(i)]))
    pytest.fail('\n'.join(lines))


@pytest.mark.parametrize('channels', [7, 6])
def test_fused_stem_stage_by_stage(channels):
  """conv2's and the 1x1's outputs (the two tensors the fused launches write) against
  the fp32 oracle and against the per-layer HIP kernels, on more tiles than the persistent
  grids have workgroups (every block walks 2-3 tiles)."""
  from oracle import inception_ref as R
  n = 80
  ref = R.make_random_model(channels, seed=11)
  w = ref.export_flat()
  x = torch.from_numpy(_images(n, channels, seed=31))
  fused = _model((100, 221, channels), w, n, fused=True)
  plain = _model((100, 221, channels), w, n, fused=False)
  fused(x.cuda())
  plain(x.cuda())
  pre = ((x.float() - 128.0) / 128.0).permute(0, 3, 1, 2)
  with torch.no_grad():
    c2 = ref.stem[1](ref.stem[0](pre))
    c4 = ref.stem[3](R._maxpool(ref.stem[2](c2)))
  want2 = c2.permute(0, 2, 3, 1).numpy()
  want4 = c4.permute(0, 2, 3, 1).numpy()
  # buffer 2 = conv2 output (halo 1: conv3 is 'same'), buffer 4 = 1x1 output (no halo)
  got2 = _interior(fused.debug_tensor(2, n).astype(np.float32), 1)
  got4 = fused.debug_tensor(4, n).astype(np.float32)
  old2 = _interior(plain.debug_tensor(2, n).astype(np.float32), 1)
  old4 = plain.debug_tensor(4, n).astype(np.float32)
  assert got2.shape == want2.shape == (n, 47, 108, 32)
  assert got4.shape == want4.shape == (n, 23, 53, 80)
  _report('conv2 vs per-layer HIP', got2, old2, 4e-3, 4e-3)
  _report('conv2 vs fp32 oracle', got2, want2, 2e-2, 2e-2)
  _report('1x1 vs per-layer HIP', got4, old4, 8e-3, 8e-3)
  _report('1x1 vs fp32 oracle', got4, want4, 3e-2, 3e-2)
  # the halo ring of the conv2 buffer must still be zero (conv3 reads it as padding)
  full2 = fused.debug_tensor(2, n)
  assert not full2[:, 0].any() and not full2[:, -1].any()
  assert not full2[:, :, 0].any() and not full2[:, :, -1].any()


@pytest.mark.parametrize('shape', [(100, 147, 8), (75, 75, 1), (120, 301, 5), (100, 199, 9), (100, 147, 10), (100, 221, 12),
                                   (90, 151, 13)])
def test_fused_stem_other_shapes(shape):
  """Tile edges at other image sizes (partial tiles in both directions); round 6: the 9..12-channel build of
  stem_a (one tap x 16 channels per chunk, uint8 patch in LDS) at the long-read model shapes, and 13 channels,
  which keep conv_first_u8 + a per-layer conv2 in front of stem_b."""
  from oracle import inception_ref as R
  h, w_, c = shape
  n = 6
  ref = R.make_random_model(c, seed=13)
  x = torch.from_numpy(np.random.default_rng(c).integers(0, 256, (n, h, w_, c), dtype=np.uint8))
  fused = _model(shape, ref.export_flat(), n, fused=True)
  plain = _model(shape, ref.export_flat(), n, fused=False)
  got = fused(x.cuda()).cpu()
  old = plain(x.cuda()).cpu()
  with torch.no_grad():
    want = ref(x)
  got4 = fused.debug_tensor(4, n).astype(np.float32)
  old4 = plain.debug_tensor(4, n).astype(np.float32)
  _report('1x1 vs per-layer HIP %s' % (shape,), got4, old4, 8e-3, 8e-3)
  assert (got - want).abs().max().item() <= 1e-3
  # two valid fp16 pipelines (different accumulation orders flip individual fp16 roundings): measured 1-6e-4 on these
  # uniform-noise images, each within 1e-3 of the oracle
  assert (got - old).abs().max().item() <= 1e-3


def test_benchmark_configuration_against_the_oracle():
  """600 images in ONE forward (400 ILLUMINA30 pileups + 200 uniform-noise images): every
  kernel variant the bench batch uses (fused stem, imgconv tiles, <NB,2>, <NB,4>) meets the
  fp32 oracle directly.  Softmax within 1e-3 on the pileups (BASELINE.json); the noise
  images (every channel uniform in 0..255 -- far outside what the encoder can draw) get 2e-3."""
  from oracle import inception_ref as R
  n = 600
  ref = R.make_random_model(7, seed=17)
  x = torch.from_numpy(_images(n, 7, seed=41))
  model = _model((100, 221, 7), ref.export_flat(), n, fused=True)
  model.calibrate_for_checkpoint(256)     # the product's model preparation (the 1e-3 bar is the product's)
  got = model(x.cuda()).cpu()
  torch.set_num_threads(min(32, os.cpu_count() or 1))
  with torch.no_grad():
    want = torch.cat([ref(x[i:i + 50]) for i in range(0, n, 50)])
  err = (got - want).abs().max(1).values
  k = (2 * n) // 3
  print('max |dp|: pileups %.3g (mean %.3g), noise %.3g (mean %.3g)' %
        (err[:k].max(), err[:k].mean(), err[k:].max(), err[k:].mean()))
  assert err[:k].max().item() <= 1e-3, err[:k].max().item()
  assert err[k:].max().item() <= 2e-3, err[k:].max().item()
  assert (want.max(0).values - want.min(0).values).max() > 1e-2


def test_features_and_logits_stage_by_stage():
  """Beyond the softmax: the stem's output, the 2048 pooled features and the logits of
  the HIP forward against the fp32 oracle with RELATIVE tolerances (a softmax of random
  weights can hide a wrong feature map behind a saturated class)."""
  from oracle import inception_ref as R
  n = 24
  ref = R.make_random_model(7, seed=23)
  w = ref.export_flat()
  x = torch.from_numpy(_images(n, 7, seed=5))
  model = _model((100, 221, 7), w, n, fused=True)
  probs = model(x.cuda()).cpu()
  pre = ((x.float() - 128.0) / 128.0).permute(0, 3, 1, 2)
  with torch.no_grad():
    s = ref.stem
    stem_out = R._maxpool(s[4](s[3](R._maxpool(s[2](s[1](s[0](pre)))))))
    feats = ref.features(pre)
    logits = ref.classification(feats)
  got_stem = model.debug_tensor(-2, n).astype(np.float32)
  if got_stem.shape[1] >= 21:
    # the stem's last max-pool is taken on the fly by mixed0's heads: the stem's last TENSOR
    # is the 21 x 51 output of the 3x3 80->192 (no halo); pool it here (max is exact)
    t = torch.from_numpy(got_stem).permute(0, 3, 1, 2)
    got_stem = R._maxpool(t).permute(0, 2, 3, 1).numpy()
  else:
    got_stem = _interior(got_stem, (got_stem.shape[1] - 10) // 2)
  want_stem = stem_out.permute(0, 2, 3, 1).numpy()
  assert got_stem.shape == want_stem.shape == (n, 10, 25, 192)
  rel = np.abs(got_stem - want_stem).max() / np.abs(want_stem).max()
  assert rel <= 1e-2, rel
  fmap = model.debug_tensor(-1, n).astype(np.float32)
  fh = (fmap.shape[1] - 1) // 2
  got_feats = _interior(fmap, fh).reshape(n, 5, 2048).mean(axis=1)
  want_feats = feats.numpy()
  rms = float(np.sqrt((want_feats ** 2).mean()))
  err = np.abs(got_feats - want_feats)
  assert err.max() <= 0.05 * rms and np.sqrt((err ** 2).mean()) <= 0.005 * rms, (err.max(), rms)
  # logits from the HIP features through the fp32 head vs the oracle's logits
  wd = ref.classification.weight.detach().numpy()
  bd = ref.classification.bias.detach().numpy()
  got_logits = got_feats @ wd.T + bd
  scale = np.abs(logits.numpy()).max()
  assert np.abs(got_logits - logits.numpy()).max() <= 5e-3 * max(scale, 1.0)
  # and the device head agrees with that host-side head
  e = np.exp(got_logits - got_logits.max(1, keepdims=True))
  # (round 6: the device head pools the float32 feature maps; the debug hook hands them over rounded to fp16, so the
  # host-side head sees ~2^-12 relative noise on the features)
  np.testing.assert_allclose(probs.numpy(), e / e.sum(1, keepdims=True), atol=2e-4)
