"""Does a calibrated bias correction for E[dW . x] buy the CNN's precision back without a second MFMA?  (CPU only.)

The HIP CNN rounds every BN-folded weight to fp16 once: W16 = W + dW.  dW is the same for every image, and the
activations it multiplies are far from zero-mean (post-ReLU, and the constant -1 rows below a pile-up), so a
layer's output error  sum_k dW[c,k] x[k]  has a per-channel mean  b[c] = E_x[ conv(x, dW)[c] ]  that can be
subtracted in the fp32 bias at no run-time cost.  b is estimated on a small calibration batch (another synthetic
seed) and the effect is measured on DIFFERENT pileups, emulating the two roundings inside the fp32 oracle as
tools/r4_precision_plan.py does.  Test infrastructure: runs the oracle only, never the product.

  python tools/r5_bias_correction.py --n 2048 --seeds 17,29,101 > profiles/r05_bias_correction_emulation.txt
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import inception_ref as R   # noqa: E402
from oracle import oracle as O          # noqa: E402

M9_10 = list(range(76, 94))
M8 = [70, 71, 72, 75]
DEFAULT_SPLIT = sorted(M8 + M9_10 + [30, 39, 40, 49, 50, 59, 60, 69])   # b1 + pooled projection of mixed4..7


def pileups(n, seed):
  from deepvariant_amd import synth
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts, multi_allelic=False)
  out, _ = O.encode_packed(opts, batch, n_threads=os.cpu_count() or 1)
  return np.ascontiguousarray(np.asarray(out).reshape(-1, 100, 221, 7)[:n])


def folded(cb):
  inv = 1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS)
  shift = cb.bn.bias - cb.bn.running_mean * inv
  return cb.conv.weight * inv[:, None, None, None], shift


def calibrate(ref, x_cal, batch=64):
  """b[i][c] = mean over calibration images and output positions of conv(x_in, W16 - W)[c]."""
  sums = [None] * len(ref.convs)
  count = [0] * len(ref.convs)
  hooks = []
  for i, cb in enumerate(ref.convs):
    with torch.no_grad():
      w, _ = folded(cb)
      dw = w.half().float() - w

    def hook(mod, inp, out, i=i, dw=dw):
      e = F.conv2d(inp[0], dw, None, mod.conv.stride, mod.conv.padding)
      s = e.sum(dim=(0, 2, 3)).double()
      sums[i] = s if sums[i] is None else sums[i] + s
      count[i] += e.shape[0] * e.shape[2] * e.shape[3]
    hooks.append(cb.register_forward_hook(hook))
  forward(ref, x_cal, batch)
  for h in hooks:
    h.remove()
  return [(s / c).float() for s, c in zip(sums, count)]


def emulated(ref, exact_w=(), corr=None, exact_a=()):
  m = copy.deepcopy(ref)
  for i, cb in enumerate(m.convs):
    with torch.no_grad():
      w, shift = folded(cb)
      if i in exact_w:
        hi = w.half().float()
        w = hi + (w - hi).half().float()
      else:
        w = w.half().float()
        if corr is not None:
          shift = shift - corr[i]
      cb.conv.weight.copy_(w)
      cb.bn.running_mean.zero_()
      cb.bn.running_var.fill_(1.0 - R.BN_EPS)
      cb.bn.bias.copy_(shift)
    if i not in exact_a:
      cb.register_forward_hook(lambda mod, inp, out: out.half().float())
  return m


def forward(model, x, batch=64):
  outs = []
  with torch.no_grad():
    for i in range(0, x.shape[0], batch):
      outs.append(model(torch.from_numpy(x[i:i + batch]), channels_last=True))
  return torch.cat(outs).numpy()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=2048)
  ap.add_argument('--ncal', type=int, default=256)
  ap.add_argument('--seeds', default='17,29,101')
  args = ap.parse_args()
  torch.set_num_threads(os.cpu_count() or 1)
  seeds = [int(s) for s in args.seeds.split(',')]
  print('# emulated fp16 CNN vs fp32 oracle, %d ILLUMINA30 pileups per weight seed, bias correction calibrated on %d '
        'OTHER pileups; max / p99.9 / mean |dp|' % (args.n, args.ncal))
  allw = set(range(94))
  for seed in seeds:
    t0 = time.time()
    ref = R.make_random_model(7, seed=seed)
    x = pileups(args.n, seed=1000 + seed)
    x_cal = pileups(args.ncal, seed=777000 + seed)
    p32 = forward(ref, x)
    corr = calibrate(ref, x_cal)
    configs = {
        'fp16 W + A (round 3)': dict(),
        'default split set (round 4)': dict(exact_w=set(DEFAULT_SPLIT)),
        'bias-corrected fp16 W + A': dict(corr=corr),
        'bias-corrected + split mixed8-10': dict(exact_w=set(M8 + M9_10), corr=corr),
        'bias-corrected + default split set': dict(exact_w=set(DEFAULT_SPLIT), corr=corr),
        'exact W everywhere (A only)': dict(exact_w=allw),
        'bias-corrected W, exact A (W residual only)': dict(corr=corr, exact_a=allw),
        'fp16 W, exact A (W only)': dict(exact_a=allw),
    }
    for name, kw in configs.items():
      e = np.abs(forward(emulated(ref, **kw), x) - p32).max(axis=1)
      print('seed %-4d %-48s %.2e / %.2e / %.2e   over 1e-3: %d  (%.0f s)' % (
          seed, name, e.max(), np.quantile(e, 0.999), e.mean(), int((e > 1e-3).sum()), time.time() - t0), flush=True)


if __name__ == '__main__':
  main()
