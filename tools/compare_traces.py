#!/usr/bin/env python3
"""Side-by-side per-launch times of DV_OP_TRACE dumps: compare_traces.py base.txt other.txt...
Only rows that differ by more than 3 us are printed (pass --all for every row)."""
import re
import sys


def load(f):
  lines = [l.rstrip() for l in open(f) if l.startswith('[dv-op]')]
  idx = [i for i, l in enumerate(lines) if 'total' in l]
  blk = lines[idx[-2] + 1:idx[-1]]
  out = []
  for l in blk:
    m = re.search(r'([\d.]+) us', l)
    out.append((l[8:l.find(m.group(0))].strip(), float(m.group(1))))
  return out


files = [a for a in sys.argv[1:] if not a.startswith('--')]
cols = [load(f) for f in files]
print('%-84s' % 'op' + ''.join('%9s' % f.split('trace_')[-1][:8].replace('.txt', '') for f in files))
for i, (name, _) in enumerate(cols[0]):
  row = [c[i][1] for c in cols]
  if '--all' in sys.argv or max(row) - min(row) > 3:
    print('%-84s' % name[:84] + ''.join('%9.1f' % x for x in row))
print('%-84s' % 'total' + ''.join('%9.1f' % sum(t for _, t in c) for c in cols))
