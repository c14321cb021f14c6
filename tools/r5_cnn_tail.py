"""Tail of |dp| (HIP CNN vs fp32 oracle) at N = 65,536 on held-out weight seeds, per precision setting (GPU box).

  python tools/r5_cnn_tail.py --n 65536 --seeds 101,202,303 --settings split,none,cal,cal+m8 > profiles/r05_cnn_tail.txt

settings (round 6): 'product' = the product's default for the shape (calibrate_for_checkpoint; precise mode for more than 8
channels), 'fast' = DV_PRECISE=0 + calibration, 'precise' = DV_PRECISE=1 + calibration, 'none' = plain fp16, uncalibrated.
Older: 'split' = round-4 default split weights, 'none' = plain fp16 weights, 'cal' = plain fp16 weights + the
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


SHAPES = {'illumina': (100, 221, 7), 'hifi': (100, 147, 10), 'ont': (100, 199, 9)}


def build(setting, weights, cal, shape=(100, 221, 7), max_batch=8192):
  from deepvariant_amd.inception_v3 import InceptionV3
  env = {'split': {'DV_SPLIT_DEFAULT': '1'}, 'none': {'DV_SPLIT_FROM': '94', 'DV_PRECISE': '0'}, 'cal': {'DV_SPLIT_FROM': '94'},
         'fast': {'DV_PRECISE': '0'},      # round 6: fp16 activations everywhere + the checkpoint calibration (the long-read shapes' opt-out)
         'cal+split': {'DV_SPLIT_DEFAULT': '1'}, 'all': {'DV_SPLIT_FROM': '0'},
         'precise': {'DV_PRECISE': '1'}}.get(setting, {})     # 'precise' (round 6): hi + lo activations through the 17x17 and 8x8 stages
  os.environ.update(env)
  try:
    m = InceptionV3(shape, max_batch=max_batch)
  finally:
    for k in env:
      os.environ.pop(k, None)
  m.load_flat_weights(weights)
  if setting.startswith('product'):          # round 6: the checkpoint's fixed calibration set ('product1024': 1,024 images of it)
    m.calibrate_for_checkpoint(int(setting[7:] or 256))
  elif setting in ('precise', 'fast'):
    m.calibrate_for_checkpoint(256)
  elif setting.startswith('cal'):
    m.calibrate(cal)
  return m


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=65536)
  ap.add_argument('--seeds', default='101,202,303')
  ap.add_argument('--settings', default='split,none')
  ap.add_argument('--ncal', default='256', help='calibration images; a comma list repeats the cal* settings per size')
  ap.add_argument('--workload', default='illumina', choices=sorted(SHAPES))
  args = ap.parse_args()
  shape = SHAPES[args.workload]
  t0 = time.time()
  if args.workload == 'illumina':
    x = T.illumina_pileups_gpu(args.n, seed=424242)
  else:
    x = T.longread_images_gpu(args.workload, args.n)
  print('# %d %s pileups %s drawn by the HIP encoder (%.0f s); HIP CNN vs fp32 oracle (torch-ROCm fp32 on the GPU)'
        % (args.n, args.workload, shape, time.time() - t0), flush=True)
  chunk = min(args.n, 8192)
  for seed in [int(s) for s in args.seeds.split(',')]:
    ref = R.make_random_model(shape[2], seed=seed)
    ref_gpu = R.make_random_model(shape[2], seed=seed).cuda()
    d = T.check_gpu_oracle(ref, ref_gpu, x, n=min(256, args.n), tol=1e-5)
    t1 = time.time()
    want = T.oracle_probs_gpu(ref_gpu, x)
    print('# seed %d: GPU oracle vs CPU oracle on 256 images max |dp| %.2e; oracle over the sample %.0f s' % (
        seed, d, time.time() - t1), flush=True)
    w = ref.export_flat()
    for setting in args.settings.split(','):
      for ncal in ([int(v) for v in args.ncal.split(',')] if setting.startswith('cal') else [0]):
        cal = None
        if ncal:
          cal = (T.illumina_pileups_gpu(ncal, seed=990000 + seed) if args.workload == 'illumina'
                 else T.longread_images_gpu(args.workload, ncal, seed=4711 + seed))
        m = build(setting, w, cal, shape, chunk)
        got = T.hip_probs(m, x, chunk)
        del m
        label = setting + ('@%d' % ncal if ncal else '')
        print('seed %d %-14s %s' % (seed, label, T.fmt(T.tail_stats(got, want))), flush=True)


if __name__ == '__main__':
  main()
