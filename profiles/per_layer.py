#!/usr/bin/env python3
"""Per-conv-layer time / TFLOP/s from a rocprofv3 rocpd results.db of bench.py.
usage: per_layer.py results.db <candidates per infer chunk> [channels]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import inception_ref as R  # noqa: E402


def main(path, n, channels=7):
  cur = sqlite3.connect(path).cursor()
  suffix = [r[0] for r in cur.execute(
      "select name from sqlite_master where type='table' and "
      "name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
  rows = list(cur.execute(
      f'select d.start, d.end, k.kernel_name, d.grid_size_x, d.grid_size_y from '
      f'rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} k '
      f'on d.kernel_id = k.id order by d.start'))
  conv = [r for r in rows if 'conv_' in r[2]]
  tab = R.conv_layer_table(channels, 100, 221)
  last = conv[-len(tab):]
  tot_us = tot_fl = 0
  print('layer kh kw  cin cout  oh  ow       us      TF/s  kernel grid')
  for i, ((s, e, name, gx, gy), (kh, kw, ci, co, oh, ow)) in enumerate(zip(last, tab)):
    fl = 2.0 * kh * kw * ci * co * oh * ow * n
    us = (e - s) / 1e3
    tot_us += us
    tot_fl += fl
    print('%3d %2d %2d %5d %4d %3d %3d %9.1f %8.1f  %s %dx%d' %
          (i, kh, kw, ci, co, oh, ow, us, fl / (e - s) / 1e3, name.split('kernelI')[1].split('EEv')[0] if 'kernelI' in name else '', gx // 256, gy))
  print('total conv: %.1f us, %.1f TFLOP/s' % (tot_us, tot_fl / tot_us / 1e6))


if __name__ == '__main__':
  main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 7)
