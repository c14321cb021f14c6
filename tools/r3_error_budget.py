"""Where the fp16 CNN's error against the fp32 oracle comes from (run on a GPU box).

The HIP Inception-v3 (libdvhip.so) rounds twice per layer: the BatchNorm-folded weights to fp16 and
every stored activation to fp16; products and sums are fp32.  This tool measures, on ILLUMINA30
pileups drawn by the product encoder, (1) max / mean |dp| of the HIP forward against the fp32
oracle (oracle/inception_ref.py) and (2) the same for an EMULATION of those two roundings inside the
oracle, switched on for one stage at a time -- the per-stage error budget VERDICT r2 asked for.

  python tools/r3_error_budget.py --n 1024 --seeds 17,29,43 > profiles/r03_error_budget.txt
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import inception_ref as R   # noqa: E402  (test infrastructure: this tool is a checker)

STAGES = ['stem', 'mixed0', 'mixed1', 'mixed2', 'mixed3', 'mixed4', 'mixed5', 'mixed6', 'mixed7', 'mixed8',
          'mixed9', 'mixed10']


def stage_of_conv():
  """construction index of a conv -> stage name (oracle/inception_ref.py builds them in order)."""
  counts = [5, 7, 7, 7, 4, 10, 10, 10, 10, 6, 9, 9]
  out = []
  for name, n in zip(STAGES, counts):
    out += [name] * n
  assert len(out) == 94
  return out


def pileups(n, seed):
  from deepvariant_amd import synth
  from deepvariant_amd.pileup_image_native import _Encoder
  opts = synth.illumina_options(7)
  batch = synth.make_illumina_batch(n, seed=seed, options=opts, multi_allelic=False)
  out, _ = _Encoder(opts, opts.width).encode(batch, 7)
  return np.ascontiguousarray(out.reshape(-1, 100, 221, 7)[:n])


def emulated(ref, round_w, round_a, head_fp16_features=False):
  """A copy of the oracle whose stages in `round_w` use BN-folded fp16 weights and whose stages in
  `round_a` round every conv + BN + ReLU output to fp16 (what the HIP kernels store)."""
  m = copy.deepcopy(ref)
  stage = stage_of_conv()
  for i, cb in enumerate(m.convs):
    if stage[i] in round_w:
      with torch.no_grad():
        inv = 1.0 / torch.sqrt(cb.bn.running_var + R.BN_EPS)
        shift = cb.bn.bias - cb.bn.running_mean * inv
        w = (cb.conv.weight * inv[:, None, None, None]).half().float()
        cb.conv.weight.copy_(w)
        cb.bn.running_mean.zero_()
        cb.bn.running_var.fill_(1.0 - R.BN_EPS)
        cb.bn.bias.copy_(shift)
    if stage[i] in round_a:
      cb.register_forward_hook(lambda mod, inp, out: out.half().float())
  return m


def forward(model, x, batch, threads):
  torch.set_num_threads(threads)
  outs = []
  with torch.no_grad():
    for i in range(0, x.shape[0], batch):
      outs.append(model(torch.from_numpy(x[i:i + batch]), channels_last=True))
  return torch.cat(outs).numpy()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=1024)
  ap.add_argument('--seeds', default='17,29,43')
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--threads', type=int, default=min(128, os.cpu_count() or 1))
  ap.add_argument('--stages', default='all')
  args = ap.parse_args()
  from deepvariant_amd.inception_v3 import InceptionV3
  print('# fp16 CNN vs fp32 oracle on %d ILLUMINA30 pileups per weight seed (100x221x7); |dp| = max over the 3 classes'
        % args.n)
  print('# emulation = the oracle with BN-folded fp16 weights (W) and / or fp16-rounded activations (A) in the named stage only')
  rows = {}
  for seed in [int(s) for s in args.seeds.split(',')]:
    t0 = time.time()
    ref = R.make_random_model(7, seed=seed)
    x = pileups(args.n, seed=1000 + seed)
    p32 = forward(ref, x, args.batch, args.threads)
    hip = InceptionV3((100, 221, 7), max_batch=min(args.n, 2048))
    hip.load_flat_weights(ref.export_flat())
    p16 = np.concatenate([hip(torch.from_numpy(x[i:i + 2048]).cuda()).cpu().numpy() for i in range(0, args.n, 2048)])
    del hip

    def err(p):
      e = np.abs(p - p32).max(axis=1)
      return float(e.max()), float(e.mean())
    rows.setdefault('HIP forward (libdvhip.so)', []).append(err(p16))
    every = set(STAGES)
    configs = [('emulated: W + A everywhere', every, every), ('emulated: W everywhere', every, set()),
               ('emulated: A everywhere', set(), every)]
    if args.stages == 'all':
      groups = [['stem'], ['mixed0', 'mixed1', 'mixed2'], ['mixed3'], ['mixed4', 'mixed5', 'mixed6', 'mixed7'],
                ['mixed8'], ['mixed9'], ['mixed10']]
      for g in groups:
        configs.append(('emulated: W + A in %s only' % '+'.join(g), set(g), set(g)))
      configs.append(('emulated: W + A everywhere EXCEPT mixed9+mixed10', every - {'mixed9', 'mixed10'},
                      every - {'mixed9', 'mixed10'}))
      configs.append(('emulated: W + A everywhere EXCEPT mixed8..10', every - {'mixed8', 'mixed9', 'mixed10'},
                      every - {'mixed8', 'mixed9', 'mixed10'}))
      configs.append(('emulated: W everywhere, A everywhere EXCEPT mixed8..10', every,
                      every - {'mixed8', 'mixed9', 'mixed10'}))
    for name, rw, ra in configs:
      rows.setdefault(name, []).append(err(forward(emulated(ref, rw, ra), x, args.batch, args.threads)))
    spread = float((p32.max(0) - p32.min(0)).max())
    print('# seed %d: %.0f s, probability spread over the batch %.3f, emulation vs HIP max |dp| %.2e' % (
        seed, time.time() - t0, spread,
        np.abs(forward(emulated(ref, every, every), x[:256], args.batch, args.threads) - p16[:256]).max()),
          flush=True)
  print('%-58s %s' % ('configuration', '   '.join('seed %-3s max / mean' % s for s in args.seeds.split(','))))
  for name, vals in rows.items():
    print('%-58s %s' % (name, '   '.join('%.2e / %.2e' % v for v in vals)))


if __name__ == '__main__':
  main()
