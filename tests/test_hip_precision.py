"""The CNN's 1e-3 bar where it is used: thousands of pileups per weight seed (GPU).

BASELINE.json `north_star`: "softmax genotype probabilities match within 1e-3" -- against the
reference's fp32 arithmetic (deepvariant/call_variants.py:913-918 runs the Keras model in
float32; deepvariant/dv_utils.py:343-366).  The other CNN tests stop at a few hundred images;
a genome is millions of candidates and the maximum error grows with the sample, so this file
checks 2048 encoder-drawn ILLUMINA30 pileups on each of three weight seeds (round-3 verdict,
profiles/r03_error_budget.txt: 1.20e-3 / 1.15e-3 at 1024 before the split weights of
model.hip `choose_split`), and the split-weight mechanism itself.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_PILEUPS = 2048


def _pileups(n, seed):
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts, multi_allelic=False)
  out, _ = _Encoder(opts, opts.width).encode(batch, 7)
  return np.ascontiguousarray(out.reshape(-1, 100, 221, 7)[:n])


def _model(weights, max_batch, split_from=None, split_default=False):
  """split_from: DV_SPLIT_FROM (94 = no layer, 0 = every conv_mfma layer); split_default: the round-4 set
  (DV_SPLIT_DEFAULT=1); neither: the product default -- plain fp16 weights since round 5."""
  from deepvariant_amd.inception_v3 import InceptionV3
  old = {k: os.environ.pop(k, None) for k in ('DV_SPLIT_FROM', 'DV_SPLIT_DEFAULT')}
  if split_from is not None:
    os.environ['DV_SPLIT_FROM'] = str(split_from)
  if split_default:
    os.environ['DV_SPLIT_DEFAULT'] = '1'
  try:
    m = InceptionV3((100, 221, 7), max_batch=max_batch)
  finally:
    for k, v in old.items():
      os.environ.pop(k, None)
      if v is not None:
        os.environ[k] = v
  m.load_flat_weights(weights)
  return m


def _oracle_probs(ref, x, batch=128):
  torch.set_num_threads(min(128, os.cpu_count() or 1))
  with torch.no_grad():
    return torch.cat([ref(torch.from_numpy(x[i:i + batch]), channels_last=True)
                      for i in range(0, len(x), batch)]).numpy()


@pytest.mark.parametrize('seed', [17, 29, 43])
def test_softmax_within_1e3_on_2048_pileups(seed):
  """max |dp| <= 1e-3 over 2048 pileups x 3 classes, product defaults: plain fp16 weights, shifts calibrated
  (dv_model_calibrate) on 256 OTHER pileups; one forward."""
  from oracle import inception_ref as R
  ref = R.make_random_model(7, seed=seed)
  x = _pileups(N_PILEUPS, seed=1000 + seed)
  model = _model(ref.export_flat(), N_PILEUPS)
  model.calibrate(torch.from_numpy(_pileups(256, seed=555000 + seed)).cuda())
  got = model(torch.from_numpy(x).cuda()).cpu().numpy()
  want = _oracle_probs(ref, x)
  err = np.abs(got - want).max(axis=1)
  spread = float((want.max(0) - want.min(0)).max())
  print('seed %d: max |dp| %.3g, mean %.3g, 99.9th percentile %.3g, probability spread %.3g' % (
      seed, err.max(), err.mean(), np.quantile(err, 0.999), spread))
  assert err.max() <= 1e-3, (seed, float(err.max()))
  if seed != 43:   # seed 43's random network is almost constant; the other two must move
    assert spread > 5e-2, spread


def _fp16_exact_weights(seed):
  """Weights whose BN-folded values are exactly representable in fp16: fp16-valued kernels and a
  moving variance with var + 1e-3 == 0.25 in float32, i.e. a fold factor of exactly 2."""
  from oracle import inception_ref as R
  ref = R.make_random_model(7, seed=seed)
  var = np.float32(0.25) - np.float32(1e-3)
  while np.float32(var + np.float32(1e-3)) < np.float32(0.25):
    var = np.nextafter(var, np.float32(1.0))
  assert np.float32(var + np.float32(1e-3)) == np.float32(0.25)
  with torch.no_grad():
    for cb in ref.convs:
      cb.conv.weight.copy_((cb.conv.weight * 0.5).half().float())
      cb.bn.running_var.fill_(float(var))
  return ref


def test_split_weights_with_zero_low_halves_are_bit_identical():
  """W_lo = 0 when the folded weights are fp16 numbers: every split layer then adds exact zeros to
  its fp32 accumulators, so a model with EVERY conv_mfma layer split and one with none agree bit
  for bit -- the (hi, lo) chunk pairing, the shared pixel slot and the doubled K walk change
  nothing else.  (Blocks 256 pixels x <NB,2> and the <NB,1> tail shape: n = 70 and n = 700.)"""
  ref = _fp16_exact_weights(seed=5)
  w = ref.export_flat()
  for n in (70, 700):
    x = torch.from_numpy(_pileups(n, seed=77 + n))
    plain = _model(w, n, split_from=94)
    split = _model(w, n, split_from=0)
    a = plain(x.cuda()).cpu().numpy()
    b = split(x.cuda()).cpu().numpy()
    fa = plain.debug_tensor(-1, n)
    fb = split.debug_tensor(-1, n)
    assert np.array_equal(fa, fb)
    assert np.array_equal(a, b)
    assert np.abs(a - _oracle_probs(ref, x.numpy())).max() <= 1e-3


def test_split_weights_move_the_features_towards_the_fp32_oracle():
  """With ordinary weights the split layers compute with W_hi + W_lo: the 2048 pooled features of
  an all-split model are measurably closer to the fp32 oracle than those of an unsplit one
  (the weight rounding is ~3/4 of the error variance, tools/r4_layer_sensitivity.py; conv_mfma
  launches are about half of the layers), and the round-4 set (DV_SPLIT_DEFAULT=1: two heads per 17x17 block + mixed8..10) lies in between."""
  from oracle import inception_ref as R
  n = 256
  ref = R.make_random_model(7, seed=17)
  w = ref.export_flat()
  x = _pileups(n, seed=303)
  pre = ((torch.from_numpy(x).float() - 128.0) / 128.0).permute(0, 3, 1, 2)
  torch.set_num_threads(min(128, os.cpu_count() or 1))
  with torch.no_grad():
    want = ref.features(pre.contiguous(memory_format=torch.channels_last)).numpy()
  rms = {}
  for name, first in (('none', 94), ('default', None), ('all', 0)):
    m = _model(w, n, split_from=first, split_default=first is None)
    m(torch.from_numpy(x).cuda())
    fmap = m.debug_tensor(-1, n).astype(np.float32)
    halo = (fmap.shape[1] - 1) // 2
    if halo:
      fmap = fmap[:, halo:-halo, halo:-halo]
    got = fmap.reshape(n, 5, 2048).mean(axis=1)
    rms[name] = float(np.sqrt(((got - want) ** 2).mean()))
  print('rms feature error vs fp32: no split %.3g, default split %.3g, all conv_mfma layers split %.3g' % (
      rms['none'], rms['default'], rms['all']))
  assert rms['all'] < 0.85 * rms['none'], rms
  assert rms['default'] < 0.97 * rms['none'], rms
