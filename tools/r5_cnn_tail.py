"""Tail of |dp| (HIP CNN vs fp32 oracle) at N = 65,536 on held-out weight seeds, per precision setting (GPU box).

  python tools/r5_cnn_tail.py --n 65536 --seeds 101,202,303 --settings split,none,cal,cal+m8 > profiles/r05_cnn_tail.txt

settings: 'split' = round-4 default split weights, 'none' = plain fp16 weights, 'cal' = plain fp16 weights + the
calibrated shift correction (dv_model_calibrate), 'cal+split' = both.  Test infrastructure (tests/cnn_tail.py).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from tests import cnn_tail as T   # noqa: E402
from oracle import inception_ref as R   # noqa: E402


def build(setting, weights, cal):
  from deepvariant_amd.inception_v3 import InceptionV3
  env = {'split': {'DV_SPLIT_DEFAULT': '1'}, 'none': {'DV_SPLIT_FROM': '94'}, 'cal': {'DV_SPLIT_FROM': '94'},
         'cal+split': {'DV_SPLIT_DEFAULT': '1'}, 'all': {'DV_SPLIT_FROM': '0'}}[setting]
  os.environ.update(env)
  try:
    m = InceptionV3((100, 221, 7), max_batch=8192)
  finally:
    for k in env:
      os.environ.pop(k, None)
  m.load_flat_weights(weights)
  if setting.startswith('cal'):
    m.calibrate(cal)
  return m


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=65536)
  ap.add_argument('--seeds', default='101,202,303')
  ap.add_argument('--settings', default='split,none')
  ap.add_argument('--ncal', type=int, default=256)
  args = ap.parse_args()
  t0 = time.time()
  x = T.illumina_pileups_gpu(args.n, seed=424242)
  print('# %d ILLUMINA30 pileups drawn by the HIP encoder (%.0f s); HIP CNN vs fp32 oracle (torch-ROCm fp32 on the GPU)'
        % (args.n, time.time() - t0), flush=True)
  for seed in [int(s) for s in args.seeds.split(',')]:
    ref = R.make_random_model(7, seed=seed)
    ref_gpu = R.make_random_model(7, seed=seed).cuda()
    d = T.check_gpu_oracle(ref, ref_gpu, x, n=256, tol=1e-5)
    t1 = time.time()
    want = T.oracle_probs_gpu(ref_gpu, x)
    print('# seed %d: GPU oracle vs CPU oracle on 256 images max |dp| %.2e; oracle over the sample %.0f s' % (
        seed, d, time.time() - t1), flush=True)
    cal = T.illumina_pileups_gpu(args.ncal, seed=990000 + seed)
    w = ref.export_flat()
    for setting in args.settings.split(','):
      m = build(setting, w, cal)
      got = T.hip_probs(m, x, 8192)
      del m
      print('seed %d %-10s %s' % (seed, setting, T.fmt(T.tail_stats(got, want))), flush=True)


if __name__ == '__main__':
  main()
