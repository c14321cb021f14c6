#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd results.db
(`rocprofv3 --kernel-trace --pmc <COUNTER> -- cmd`).  usage: summarize_pmc.py results.db
Prints: kernel, counter, launches, total, per-launch.  FETCH_SIZE/WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE counts a 128-B request as 64 B (MI355X_MICROARCH.md, HBM section),
so wide coalesced reads are up to 2x the printed figure."""
import collections
import sqlite3
import sys


def main(path):
  cur = sqlite3.connect(path).cursor()
  suf = [r[0] for r in cur.execute(
      "select name from sqlite_master where type='table' and "
      "name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
  rows = cur.execute(
      f'select k.kernel_name, i.name, p.value, d.id from rocpd_pmc_event{suf} p '
      f'join rocpd_info_pmc{suf} i on p.pmc_id=i.id '
      f'join rocpd_kernel_dispatch{suf} d on p.event_id=d.event_id '
      f'join rocpd_info_kernel_symbol{suf} k on d.kernel_id=k.id')
  agg = collections.defaultdict(lambda: [0.0, set()])
  for kn, cn, v, did in rows:
    a = agg[(kn, cn)]
    a[0] += v
    a[1].add(did)
  print('%-110s %-14s %8s %14s %14s' % ('Kernel', 'Counter', 'Launches', 'Total', 'PerLaunch'))
  for (kn, cn), (v, ids) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-110s %-14s %8d %14.4g %14.4g' % (kn[:110], cn, len(ids), v, v / max(len(ids), 1)))


if __name__ == '__main__':
  main(sys.argv[1])
