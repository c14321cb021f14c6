#!/usr/bin/env python3
"""Side-by-side per-launch times of DV_OP_TRACE dumps: compare_traces.py base.txt other.txt..."""
import sys


def load(f):
  lines = [l for l in open(f) if l.startswith('[dv-op]')]
  idx = [i for i, l in enumerate(lines) if 'total' in l]
  blk = lines[idx[-2] + 1:idx[-1]]
  return [(l[8:66].strip(), float(l[66:].split()[0])) for l in blk]


cols = [load(f) for f in sys.argv[1:]]
print('%-58s' % 'op' + ''.join('%9s' % f.split('trace_')[-1][:8].replace('.txt', '') for f in sys.argv[1:]))
for i, (name, _) in enumerate(cols[0]):
  print('%-58s' % name + ''.join('%9.1f' % c[i][1] for c in cols))
print('%-58s' % 'total' + ''.join('%9.1f' % sum(t for _, t in c) for c in cols))
