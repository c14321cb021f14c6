"""max / mean |dp| of the HIP CNN against the fp32 oracle for several split-weight settings (GPU box).

  python tools/r4_precision_gpu.py --n 2048 --seeds 17,29 --split_from 94,70,30,12,0
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))

from oracle import inception_ref as R   # noqa: E402  (checker)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=2048)
  ap.add_argument('--seeds', default='17,29')
  ap.add_argument('--split_from', default='94,70,0')
  ap.add_argument('--layer_sets', default='', help="';'-separated DV_SPLIT_LAYERS lists to sweep instead")
  args = ap.parse_args()
  import test_hip_precision as T
  print('# HIP forward vs fp32 oracle, %d ILLUMINA30 pileups per seed; max / mean / 99.9th percentile |dp|' % args.n)
  for seed in [int(s) for s in args.seeds.split(',')]:
    ref = R.make_random_model(7, seed=seed)
    x = T._pileups(args.n, seed=1000 + seed)
    want = T._oracle_probs(ref, x)
    settings = ([('DV_SPLIT_LAYERS', v) for v in args.layer_sets.split(';')] if args.layer_sets else
                [('DV_SPLIT_FROM', v) for v in args.split_from.split(',')])
    for key, val in settings:
      os.environ[key] = val
      try:
        m = T._model(ref.export_flat(), args.n, split_from=None if key == 'DV_SPLIT_LAYERS' else int(val))
      finally:
        os.environ.pop(key, None)
      got = m(torch.from_numpy(x).cuda()).cpu().numpy()
      del m
      e = np.abs(got - want).max(axis=1)
      print('seed %d %s=%-3s  %.3e / %.3e / %.3e' % (seed, key, val, e.max(), e.mean(), np.quantile(e, 0.999)),
            flush=True)


if __name__ == '__main__':
  main()
