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
  """kernel name + its integer / bool template arguments (Li4ELi2E... -> <4,2,...>)."""
  for name in ('conv_mfma_kernel', 'conv_pool_resident_kernel', 'conv_resident_kernel', 'stem_a_kernel', 'stem_b_kernel',
               'imgconv_kernel', 'chain_kernel', 'encode_items_kernel', 'avgpool3s1_kernel', 'maxpool3s2_kernel',
               'head_kernel', 'conv_first_u8_kernel', 'merge_alt_channels_kernel', 'conv_pool1x1_kernel'):
    if name in k:
      t = re.search(name + r'I((?:L[ib]\d+E)+)', k)      # (summarize_pmc.py cuts names at 64 characters)
      args = ','.join(re.findall(r'L[ib](\d+)E', t.group(1))) if t else ''
      return name + ('<%s>' % args if args else '')
  return re.sub(r'^_ZN?\d*', '', k)[:40]


def ratio(d, a, b, scale=1.0):
  return '%6.3f' % (scale * d[a] / d[b]) if a in d and b in d and d[b] else '     -'


print('# counters: sums over every SQ / TA of the chip and every launch of the kernel in one eager run of bench.py --steps 1 --warmup 1')
print('# (three forwards); three separate --pmc passes (SQ set 1, SQ set 2 / LDS, TA / TCP / GRBM), --kernel-trace only')
print('# mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of the chip\'s matrix-pipe cycles that')
print('#              execute an MFMA while the kernel runs (32 cycles per v_mfma_f32_32x32x16_f16, checked against SQ_INSTS_MFMA)')
print('# wait_inst  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked on s_waitcnt /')
print('#              s_barrier), issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; lds_act / lds_wait likewise; bank_cf = SQ_LDS_BANK_CONFLICT /')
print('#              SQ_LDS_IDX_ACTIVE (extra LDS cycles per LDS-array cycle); ta_busy = TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 x 256 TAs)')
print('%-52s %5s %9s %9s %9s %9s %9s %9s %9s %9s' % ('kernel', 'n', 'mfma_util', 'wait_inst', 'wait_any', 'issue', 'lds_act',
                                                    'lds_wait', 'bank_cf', 'ta_busy'))
order = sorted(rows, key=lambda k: -rows[k].get('GRBM_GUI_ACTIVE', 0))
for k in order:
  d = rows[k]
  if 'SQ_BUSY_CYCLES' not in d or 'GRBM_GUI_ACTIVE' not in d:
    continue
  cyc = d['GRBM_GUI_ACTIVE'] / 8.0
  util = '%6.3f' % (d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024)) if cyc else '     -'
  ta = '%6.3f' % (d['TA_TA_BUSY_sum'] / (cyc * 256)) if cyc and 'TA_TA_BUSY_sum' in d else '     -'
  print('%-52s %5d %9s %9s %9s %9s %9s %9s %9s %9s' % (
      short(k)[:52], launches[k], util, ratio(d, 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'), ratio(d, 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_ACTIVE_INST_LDS', 'SQ_WAVE_CYCLES'), ratio(d, 'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES'),
      ratio(d, 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'), ta))
print()
print('# raw per-kernel totals')
for k in order:
  print(short(k), ' '.join('%s=%.4g' % (c, v) for c, v in sorted(rows[k].items())))
