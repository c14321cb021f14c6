#!/usr/bin/env python3
"""Per-kernel table from the concatenated summarize_pmc.py outputs of tools/r5_pmc_sq.sh.

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CYCLES-per-SE-normalised ...): the counters are sums over
all SQs; the ratios printed here only divide counters of the same pass and the same kernel:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * simds_per_sq_sum)   (see the header of the output)
  wait_inst   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        (share of resident-wave time parked on s_waitcnt)
  wait_any    = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  issue       = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_active  = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES,  lds_wait = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  bank_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
launches = {}
for line in open(sys.argv[1]):
  m = re.match(r'(\S+)\s+(\S+)\s+(\d+)\s+(\S+)\s+(\S+)\s*$', line)
  if not m or m.group(1) == 'Kernel':
    continue
  k, c, n, tot = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
  rows[k][c] = tot
  launches[k] = n


def short(k):
  k = re.sub(r'^_ZN?\d*', '', k)
  for name in ('conv_mfma_kernel', 'conv_pool_resident_kernel', 'conv_resident_kernel', 'stem_a_kernel', 'stem_b_kernel',
               'imgconv_kernel', 'chain_kernel', 'encode_items_kernel', 'avgpool3s1_kernel', 'maxpool3s2_kernel',
               'head_kernel', 'conv_first_u8_kernel', 'merge_alt_channels_kernel', 'conv_pool1x1_kernel'):
    if name in k:
      t = re.search(name + r'(I[^E]*E)?', k)
      return name + (t.group(1) or '' if t else '')
  return k[:40]


def ratio(d, a, b, scale=1.0):
  return '%6.3f' % (scale * d[a] / d[b]) if a in d and b in d and d[b] else '     -'


print('# counters: sums over every SQ / TA of the chip and every launch of the kernel in one eager forward')
print('# mfma/busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES: MFMA-pipe busy cycles summed over SIMDs per SQ-busy cycle')
print('#   (SQ_BUSY_CYCLES counts per SE/XCD instance, see the raw file; compare kernels, and against mfma_peak below)')
print('# mfma_cyc/wave_cyc = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES): SQ_WAVE_CYCLES is in quad-cycles')
print('%-44s %5s %9s %9s %9s %9s %9s %9s %9s %9s' % ('kernel', 'n', 'mfma/busy', 'mfma/wave', 'wait_inst', 'wait_any',
                                                    'issue', 'lds_act', 'lds_wait', 'bank_cf'))
order = sorted(rows, key=lambda k: -rows[k].get('SQ_BUSY_CYCLES', 0))
for k in order:
  d = rows[k]
  if 'SQ_BUSY_CYCLES' not in d:
    continue
  print('%-44s %5d %9s %9s %9s %9s %9s %9s %9s %9s' % (
      short(k)[:44], launches[k], ratio(d, 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES'),
      ratio(d, 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 0.25), ratio(d, 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'), ratio(d, 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_ACTIVE_INST_LDS', 'SQ_WAVE_CYCLES'), ratio(d, 'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')))
print()
print('# raw per-kernel totals')
for k in order:
  print(short(k), ' '.join('%s=%.4g' % (c, v) for c, v in sorted(rows[k].items())))
