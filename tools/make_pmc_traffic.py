#!/usr/bin/env python3
"""pmc_fetch.txt + pmc_write.txt (profiles/summarize_pmc.py tables of the separate
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over `bench.py --steps 2 --warmup 1`)
-> the JSON bench.py reads for `roofline.traffic`.

usage: make_pmc_traffic.py pmc_fetch.txt pmc_write.txt <candidates per step> <source label> > out.json
Units: the counters are in KB (x1024 B); gfx950 counts a 128-byte fetch as 64 bytes, so FETCH is
doubled (upper bound for wide reads, MI355X_MICROARCH.md HBM section)."""
import json
import re
import sys

CONV = ('conv_mfma_kernel', 'conv_resident_kernel', 'conv_pool1x1_kernel', 'conv_first_u8_kernel', 'stem_a_kernel',
        'stem_b_kernel', 'imgconv_kernel')


def table(path, counter):
  out = {}
  for line in open(path):
    parts = line.split()
    if len(parts) >= 5 and parts[1] == counter:     # (names are cut at 64 characters: accumulate)
      n, v = out.get(parts[0], (0, 0.0))
      out[parts[0]] = (n + int(parts[2]), v + float(parts[3]) * 1024.0)
  return out


fetch, write = table(sys.argv[1], 'FETCH_SIZE'), table(sys.argv[2], 'WRITE_SIZE')
n_items, source = int(sys.argv[3]), sys.argv[4]
enc = [k for k in fetch if 'encode_items_kernel' in k][0]
passes = fetch[enc][0]                       # one encoder launch per forward pass
conv = [k for k in fetch if any(c in k for c in CONV)]
json.dump({
    'source': source,
    'candidates_per_step': n_items,
    'forward_passes': passes,
    'conv': {
        'launches_per_pass': sum(fetch[k][0] for k in conv) / passes,
        'fetch_bytes_per_pass_x2': 2.0 * sum(fetch[k][1] for k in conv) / passes,
        'write_bytes_per_pass': sum(write[k][1] for k in conv if k in write) / passes,
    },
    'all_kernels': {
        'fetch_bytes_per_pass_x2': 2.0 * sum(v[1] for k, v in fetch.items() if 'fillBuffer' not in k) / passes,
        'write_bytes_per_pass': sum(v[1] for k, v in write.items() if 'fillBuffer' not in k) / passes,
    },
    'encoder': {
        'fetch_bytes_per_launch_x2': 2.0 * fetch[enc][1] / passes,
        'write_bytes_per_launch': write[enc][1] / passes,
    },
}, sys.stdout, indent=1)
