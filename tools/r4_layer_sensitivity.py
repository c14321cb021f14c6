"""Per-layer, first-order budget of the fp16 CNN's error (CPU only; test infrastructure).

tools/r3_error_budget.py measures the error stage by stage by re-running the fp32 oracle with the
HIP kernels' two roundings emulated in one stage at a time.  This tool gets the same budget PER
LAYER from one backward pass per probability: a rounding error e on a stored quantity q moves a
probability by (dp/dq) e, the roundings are independent and uniform over one fp16 ulp, so

    Var(dp) = sum over stored activations a   (dp/da)^2  ulp(a)^2 / 12          (A, per layer)
            + sum over folded weights w'      (dp/dw')^2 ulp(w')^2 / 12         (W, per layer)

with w' = w / sqrt(var + eps) (what the kernels store in fp16).  Output: the share of every layer
in that variance, W and A apart, averaged over the sampled pileups -- which layers a higher
precision has to cover to move the tail of max |dp|.

  python tools/r4_layer_sensitivity.py --n 24 --seeds 17,29 > profiles/r04_layer_sensitivity.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import inception_ref as R   # noqa: E402
from oracle import oracle as O          # noqa: E402

STAGES = ['stem', 'mixed0', 'mixed1', 'mixed2', 'mixed3', 'mixed4', 'mixed5', 'mixed6', 'mixed7', 'mixed8',
          'mixed9', 'mixed10']
COUNTS = [5, 7, 7, 7, 4, 10, 10, 10, 10, 6, 9, 9]


def pileups(n, seed):
  from deepvariant_amd import synth
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts, multi_allelic=False)
  out, _ = O.encode_packed(opts, batch, n_threads=8)
  return np.ascontiguousarray(np.asarray(out).reshape(-1, 100, 221, 7)[:n])


def ulp16(t):
  """fp16 unit in the last place at |t| (normal range; subnormals share 2^-24)."""
  e = torch.floor(torch.log2(t.abs().clamp_min(2.0 ** -14)))
  return torch.pow(2.0, e - 10)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=24)
  ap.add_argument('--seeds', default='17,29')
  args = ap.parse_args()
  torch.set_num_threads(os.cpu_count() or 1)
  stage = []
  for name, c in zip(STAGES, COUNTS):
    stage += [name] * c
  for seed in [int(s) for s in args.seeds.split(',')]:
    ref = R.make_random_model(7, seed=seed)
    for p in ref.parameters():
      p.requires_grad_(True)
    x = pileups(args.n, seed=1000 + seed)
    nl = len(ref.convs)
    var_w = np.zeros(nl)
    var_a = np.zeros(nl)
    var_tot = []
    acts = {}
    hooks = []
    for i, cb in enumerate(ref.convs):
      def hook(mod, inp, out, i=i):
        out.retain_grad()
        acts[i] = out
      hooks.append(cb.register_forward_hook(hook))
    inv = [1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS) for cb in ref.convs]
    for k in range(args.n):
      img = torch.from_numpy(x[k:k + 1])
      worst = None
      for j in range(3):
        ref.zero_grad(set_to_none=True)
        p = ref(img)
        p[0, j].backward()
        vw = np.zeros(nl)
        va = np.zeros(nl)
        for i, cb in enumerate(ref.convs):
          a = acts[i].detach()
          g = acts[i].grad
          va[i] = float((g * g * ulp16(a) ** 2 * (a > 0)).sum() / 12.0)
          wf = cb.conv.weight.detach() * inv[i][:, None, None, None]
          gw = cb.conv.weight.grad / inv[i][:, None, None, None]
          vw[i] = float((gw * gw * ulp16(wf) ** 2).sum() / 12.0)
        if worst is None or vw.sum() + va.sum() > worst[0].sum() + worst[1].sum():
          worst = (vw, va)
      var_w += worst[0]
      var_a += worst[1]
      var_tot.append(worst[0].sum() + worst[1].sum())
    for h in hooks:
      h.remove()
    var_w /= args.n
    var_a /= args.n
    tot = var_w.sum() + var_a.sum()
    sig = np.sqrt(np.array(var_tot))
    print('# seed %d, %d pileups: predicted sigma(dp) of the worst class: mean %.2e, max %.2e  (mean |dp| = 0.8 sigma)'
          % (seed, args.n, sig.mean(), sig.max()))
    print('# share of Var(dp): W %.1f %%, A %.1f %%' % (100 * var_w.sum() / tot, 100 * var_a.sum() / tot))
    print('%-4s %-8s %-22s %8s %8s' % ('idx', 'stage', 'layer', 'W %', 'A %'))
    for i, cb in enumerate(ref.convs):
      co, ci, kh, kw = cb.conv.weight.shape
      print('%-4d %-8s %-22s %8.2f %8.2f' % (i, stage[i], '%dx%d %d->%d' % (kh, kw, ci, co),
                                             100 * var_w[i] / tot, 100 * var_a[i] / tot))
    print('%-8s %8s %8s' % ('stage', 'W %', 'A %'))
    for s in STAGES:
      idx = [i for i in range(nl) if stage[i] == s]
      print('%-8s %8.2f %8.2f' % (s, 100 * var_w[idx].sum() / tot, 100 * var_a[idx].sum() / tot))
    sys.stdout.flush()


if __name__ == '__main__':
  main()
