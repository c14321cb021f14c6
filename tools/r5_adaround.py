"""Prototype (GPU box, test infrastructure): does choosing the ROUNDING DIRECTION of every fp16 weight against the
layer's input statistics -- instead of rounding to nearest -- buy the CNN's 1e-3 bar on the long-read shapes?

For a convolution with BN-folded fp32 weights W [cout, K] and inputs x (im2col patches), the fp16 image W_q changes a
pre-activation by dW . x with dW = W_q - W.  Rounding to nearest makes the dW_k independent (variance sum_k dW_k^2
E[x_k^2]); the x_k of a real layer are strongly correlated, so picking floor / ceil per weight to minimise
dW C dW^T, C = E[x x^T] measured on a few calibration images, cancels most of it (the idea of AdaRound, applied at 11
bits).  The fp32 weights handed to dv_model_load_weights are W_q / inv, so the product's own rounding reproduces the
choice; the oracle keeps the ORIGINAL weights.

  python tools/r5_adaround.py --workload ont --seeds 202 --n 2048
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import cnn_tail as T           # noqa: E402
from oracle import inception_ref as R    # noqa: E402

SHAPES = {'illumina': (100, 221, 7), 'hifi': (100, 147, 10), 'ont': (100, 199, 9)}


def fp16_neighbours(w):
  """(nearest fp16 of w, the fp16 neighbour on the other side of w), both as float32."""
  near = w.half().float()
  exp = torch.floor(torch.log2(near.abs().clamp_min(6.2e-5)))
  ulp = torch.pow(2.0, exp - 10)
  other = near + torch.where(w > near, ulp, -ulp)
  other = other.half().float()
  same = other == near          # (exactly representable weights, or clamped subnormals): no choice
  return near, torch.where(same, near, other)


def optimise_layer(wf, cmat, sweeps, gen):
  """wf [cout, K] folded fp32 weights, cmat [K, K] second moments of the inputs -> (W_q, cost before, after).
  Coordinate descent: the output channels are independent problems, so one step decides column k for every
  cout at once (exact per row), in a random column order, `sweeps` times."""
  near, other = fp16_neighbours(wf)
  cur = near - wf
  alt = other - wf
  diag = torch.diagonal(cmat)
  g = cur @ cmat
  before = float((cur * g).sum())
  k_total = wf.shape[1]
  for _ in range(sweeps):
    order = torch.randperm(k_total, generator=gen, device=wf.device).tolist()
    flips = 0
    for k in order:
      d = alt[:, k] - cur[:, k]
      gain = 2.0 * d * g[:, k] + d * d * diag[k]
      flip = gain < 0
      if not bool(flip.any()):
        continue
      d = torch.where(flip, d, torch.zeros_like(d))
      g += d[:, None] * cmat[k][None, :]
      new_cur = cur[:, k] + d
      alt[:, k] = torch.where(flip, cur[:, k], alt[:, k])
      cur[:, k] = new_cur
      flips += int(flip.sum())
    if flips == 0:
      break
  after = float((cur * (cur @ cmat)).sum())
  return wf + cur, before, after


def adaround(ref_gpu, images, n_stat=48, max_cols=150000, sweeps=3, seed=0):
  """-> a copy of `ref_gpu` (fp32) whose conv weights round to the optimised fp16 images after BN folding."""
  gen = torch.Generator(device=images.device).manual_seed(seed)
  out = copy.deepcopy(ref_gpu)
  new_w = {}
  report = []

  def hook(cb, inp):
    x = inp[0][:n_stat]
    co, ci, kh, kw = cb.conv.weight.shape
    cols = F.unfold(x, (kh, kw), padding=cb.conv.padding, stride=cb.conv.stride)    # [n, K, L]
    p = cols.permute(1, 0, 2).reshape(cols.shape[1], -1)
    if p.shape[1] > max_cols:
      idx = torch.randint(0, p.shape[1], (max_cols,), generator=gen, device=p.device)
      p = p[:, idx]
    cmat = (p @ p.t()) / p.shape[1]
    inv = 1.0 / torch.sqrt(cb.bn.running_var + cb.bn.eps)
    wf = cb.conv.weight.reshape(co, -1) * inv[:, None]
    wq, before, after = optimise_layer(wf, cmat, sweeps, gen)
    new_w[id(cb)] = (wq / inv[:, None]).reshape(co, ci, kh, kw)
    report.append((before, after))

  hooks = [cb.register_forward_pre_hook(hook) for cb in ref_gpu.convs]
  R.ConvBN.as_gemm = True
  try:
    with torch.no_grad():
      ref_gpu(images[:n_stat])
  finally:
    R.ConvBN.as_gemm = False
    for h in hooks:
      h.remove()
  with torch.no_grad():
    for cb_src, cb_dst in zip(ref_gpu.convs, out.convs):
      cb_dst.conv.weight.copy_(new_w[id(cb_src)])
  b = sum(r[0] for r in report)
  a = sum(r[1] for r in report)
  return out, b, a


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='ont', choices=sorted(SHAPES))
  ap.add_argument('--n', type=int, default=2048)
  ap.add_argument('--seeds', default='202')
  ap.add_argument('--ncal', type=int, default=256)
  ap.add_argument('--sweeps', type=int, default=3)
  args = ap.parse_args()
  from deepvariant_amd.inception_v3 import InceptionV3
  shape = SHAPES[args.workload]
  x = T.illumina_pileups_gpu(args.n, seed=424242) if args.workload == 'illumina' else T.longread_images_gpu(args.workload, args.n)
  chunk = min(args.n, 8192)
  for seed in [int(s) for s in args.seeds.split(',')]:
    ref = R.make_random_model(shape[2], seed=seed)
    ref_gpu = R.make_random_model(shape[2], seed=seed).cuda()
    want = T.oracle_probs_gpu(ref_gpu, x)
    cal = (T.illumina_pileups_gpu(args.ncal, seed=990000 + seed) if args.workload == 'illumina'
           else T.longread_images_gpu(args.workload, args.ncal, seed=4711 + seed))
    t0 = time.time()
    tuned, before, after = adaround(ref_gpu, cal, sweeps=args.sweeps, seed=seed)
    print('# %s seed %d: rounding optimised on %d calibration images in %.0f s; sum over layers of E[(dW.x)^2]: nearest %.4g -> '
          'optimised %.4g (x %.3f)' % (args.workload, seed, min(48, args.ncal), time.time() - t0, before, after, after / before),
          flush=True)
    flats = {'nearest': ref.export_flat(), 'optimised': tuned.cpu().export_flat()}
    for name, flat in flats.items():
      for calibrated in (False, True):
        m = InceptionV3(shape, max_batch=chunk)
        m.load_flat_weights(flat)
        if calibrated:
          m.calibrate(cal)
        got = T.hip_probs(m, x, chunk)
        del m
        print('seed %d %-9s %-12s %s' % (seed, name, 'calibrated' if calibrated else 'uncalibrated',
                                        T.fmt(T.tail_stats(got, want))), flush=True)


if __name__ == '__main__':
  main()
