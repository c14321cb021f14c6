"""Which layers have to carry exact (split fp16) weights for max |dp| <= 1e-3 at N >= 2048?  (CPU only.)

Emulates the HIP CNN's two roundings inside the fp32 oracle (as tools/r3_error_budget.py does: BN-folded
weights rounded to fp16 = W, every stored activation rounded to fp16 = A) with the W rounding switched OFF
for a chosen set of layers -- what a W_hi + W_lo weight image (two MFMAs per product, include/dvhip.h
`dv_model_desc.split_weights`) gives -- and reports max / mean |dp| against the unrounded oracle on
encoder-drawn ILLUMINA30 pileups.  Test infrastructure: runs the oracle only, never the product.

  python tools/r4_precision_plan.py --n 2048 --seeds 17,29,43 > profiles/r04_precision_plan.txt
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import inception_ref as R   # noqa: E402
from oracle import oracle as O          # noqa: E402

M9_10 = list(range(76, 94))
HEADS17 = [30, 31, 34, 39, 40, 41, 44, 49, 50, 51, 54, 59, 60, 61, 64, 69]
M8 = [70, 71, 72, 75]
HEADS35 = [5, 6, 8, 11, 12, 13, 15, 18, 19, 20, 22, 25]
M3 = [26, 27, 28, 29]
GAP_IN = [85, 87, 88, 91, 92, 93]

CONFIGS = {
    'baseline: fp16 W + A everywhere': ([], []),
    'split W in mixed9+10': (M9_10, []),
    'split W in mixed9+10, fp32 GAP input': (M9_10, GAP_IN),
    'split W in mixed8 convs + mixed9+10, fp32 GAP input': (M8 + M9_10, GAP_IN),
    'split W in 17x17 heads + mixed8 + mixed9+10, fp32 GAP input': (HEADS17 + M8 + M9_10, GAP_IN),
    'split W in conv1 + all 1x1 heads + mixed3 + mixed8 + mixed9+10, fp32 GAP input':
        ([0] + HEADS35 + M3 + HEADS17 + M8 + M9_10, GAP_IN),
    'split W everywhere (A only)': (list(range(94)), []),
    'split W + exact A in mixed8 convs + mixed9+10': (M8 + M9_10, list(range(70, 94))),
    'exact A everywhere (W only)': ([], list(range(94))),
}


def pileups(n, seed):
  from deepvariant_amd import synth
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts, multi_allelic=False)
  out, _ = O.encode_packed(opts, batch, n_threads=os.cpu_count() or 1)
  return np.ascontiguousarray(np.asarray(out).reshape(-1, 100, 221, 7)[:n])


def emulated(ref, exact_w, exact_a):
  m = copy.deepcopy(ref)
  for i, cb in enumerate(m.convs):
    with torch.no_grad():
      inv = 1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS)
      shift = cb.bn.bias - cb.bn.running_mean * inv
      w = cb.conv.weight * inv[:, None, None, None]
      if i in exact_w:      # W_hi + W_lo: 22 significant bits
        hi = w.half().float()
        w = hi + (w - hi).half().float()
      else:
        w = w.half().float()
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
  ap.add_argument('--seeds', default='17,29,43')
  ap.add_argument('--only', default='')
  args = ap.parse_args()
  torch.set_num_threads(os.cpu_count() or 1)
  seeds = [int(s) for s in args.seeds.split(',')]
  print('# emulated fp16 CNN vs fp32 oracle, %d ILLUMINA30 pileups per weight seed; max / mean |dp|' % args.n)
  rows = {}
  for seed in seeds:
    t0 = time.time()
    ref = R.make_random_model(7, seed=seed)
    x = pileups(args.n, seed=1000 + seed)
    p32 = forward(ref, x)
    for name, (ew, ea) in CONFIGS.items():
      if args.only and args.only not in name:
        continue
      e = np.abs(forward(emulated(ref, set(ew), set(ea)), x) - p32).max(axis=1)
      rows.setdefault(name, []).append((float(e.max()), float(e.mean())))
      print('# seed %d %-70s %.2e / %.2e  (%.0f s)' % (seed, name, e.max(), e.mean(), time.time() - t0), flush=True)
  print('%-82s %s' % ('configuration', '   '.join('seed %-3d max / mean' % s for s in seeds)))
  for name, vals in rows.items():
    print('%-82s %s' % (name, '   '.join('%.2e / %.2e' % v for v in vals)))


if __name__ == '__main__':
  main()
