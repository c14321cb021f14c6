#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, %) from a rocprofv3 rocpd
results.db -- ROCm 7.2's rocprofv3 writes a sqlite database by default; this is
the same table `--stats` prints.  usage: summarize_rocpd.py results.db"""
import collections
import sqlite3
import sys


def main(path):
  cur = sqlite3.connect(path).cursor()
  suffix = [r[0] for r in cur.execute(
      "select name from sqlite_master where type='table' and "
      "name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
  rows = cur.execute(
      f'select d.start, d.end, k.kernel_name from rocpd_kernel_dispatch{suffix} d '
      f'join rocpd_info_kernel_symbol{suffix} k on d.kernel_id = k.id')
  agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
  for s, e, n in rows:
    a = agg[n]
    a[0] += e - s
    a[1] += 1
    a[2] = min(a[2], e - s)
    a[3] = max(a[3], e - s)
  tot = sum(v[0] for v in agg.values()) or 1
  print('%-70s %8s %12s %10s %10s %10s %7s' %
        ('Name', 'Calls', 'TotalNs', 'AvgNs', 'MinNs', 'MaxNs', 'Pct'))
  for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-70s %8d %12d %10d %10d %10d %6.2f%%' %
          (n[:70], v[1], v[0], v[0] // v[1], v[2], v[3], 100.0 * v[0] / tot))


if __name__ == '__main__':
  main(sys.argv[1])
