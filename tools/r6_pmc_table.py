#!/usr/bin/env python3
"""Per-kernel table from the concatenated summarize_pmc.py outputs of tools/r6_pmc_sq.sh  (round 6: + the MFMA FLOPs each
kernel family EXECUTED (SQ_INSTS_MFMA x 32,768) against its NOMINAL FLOPs from an op trace of the same command, argv[2]).

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
# ---- executed / nominal MFMA FLOPs per kernel family
def is_avg(k):
  """conv_mfma_kernel<NB,PT,MINB,SLAB,WAVES,SPLIT,SIDE_POOL,AVG[,WIDE]>: the pooled-heads variant."""
  m = re.search(r'conv_mfma_kernel<([\d,]+)>', short(k))
  a = m.group(1).split(',') if m else []
  return len(a) >= 8 and a[7] == '1'


FAMILIES = [   # (kernel-name test, op-trace label test, title)
    (lambda k: 'stem_a_kernel' in k, lambda l: 'stem_a ' in l, 'stem_a'),
    (lambda k: 'stem_b_kernel' in k, lambda l: 'stem_b ' in l, 'stem_b'),
    (lambda k: 'conv_pool_resident' in k, lambda l: '-> maxpool3s2' in l and 'resident' in l, 'conv_pool_resident (3x3 80->192 + pool)'),
    (lambda k: 'chain_kernel' in k, lambda l: l.startswith('chain '), 'chain_kernel (all)'),
    (lambda k: 'imgconv_kernel' in k, lambda l: '[imgconv' in l, 'imgconv_kernel (all)'),
    (lambda k: 'conv_mfma_kernel' in k and is_avg(k), lambda l: 'avgpool3s1 in the epilogue' in l,
     'conv_mfma<4,2,...,AVG> (pooled heads)'),
    (lambda k: 'conv_mfma_kernel' in k and not is_avg(k) or 'conv_first_u8' in k or 'conv_resident_kernel' in k
     or 'conv_pool1x1' in k,
     lambda l: l.startswith('conv') and 'avgpool3s1 in the epilogue' not in l and '[imgconv' not in l and
     not ('-> maxpool3s2' in l and 'resident' in l), 'conv_mfma (every other variant) + conv_first_u8'),
]
if len(sys.argv) > 2:
  nominal = collections.defaultdict(float)
  blocks = 0
  for line in open(sys.argv[2]):
    m = re.match(r'\[dv-op\] (.*?)\s+([\d.]+) us\s+([\d.]+) TF/s', line)
    if line.startswith('[dv-op] total'):
      blocks += 1
    if not m:
      continue
    label = re.sub(r'^\[blank[^\]]*\] ', '', m.group(1))
    for i, (_, lt, _) in enumerate(FAMILIES):
      if lt(label):
        nominal[i] += float(m.group(2)) * 1e-6 * float(m.group(3)) * 1e12
        break
  print()
  print('# MFMA FLOPs executed (SQ_INSTS_MFMA x 32,768 per v_mfma_f32_32x32x16_f16) / nominal FLOPs (2 x MACs of the layers, from')
  print('# %s: %d forwards there, the counter run\'s launches scaled to it).  > 1: halo recomputation, cout padding,' % (sys.argv[2], blocks))
  print('# idle fragment slots; < 1: blank-row skipping (stem kernels) and the row-band / blank-tap work the nominal count includes')
  print('%-52s %14s %14s %8s' % ('family', 'executed TFLOP', 'nominal TFLOP', 'ratio'))
  tot_e = tot_n = 0.0
  for i, (kt, _, title) in enumerate(FAMILIES):
    ks = [k for k in rows if kt(k) and 'SQ_INSTS_MFMA' in rows[k]]
    if not ks or not nominal[i] or not blocks:
      continue
    # full-size forwards in the counter run = launches of the encoder (once per step; the model's own one-example
    # forward on the all-blank image at load time launches every classifier kernel once more, with next to no work)
    fw = max([launches[k] for k in rows if 'encode_items_kernel' in k] or [1])
    ex = sum(rows[k]['SQ_INSTS_MFMA'] for k in ks) * 32768.0 / fw
    nm = nominal[i] / blocks
    tot_e += ex
    tot_n += nm
    print('%-52s %14.3f %14.3f %8.3f' % (title, ex / 1e12, nm / 1e12, ex / nm))
  if tot_n:
    print('%-52s %14.3f %14.3f %8.3f' % ('all conv kernels, per forward', tot_e / 1e12, tot_n / 1e12, tot_e / tot_n))
print()
print('# raw per-kernel totals')
for k in order:
  print(short(k), ' '.join('%s=%.4g' % (c, v) for c, v in sorted(rows[k].items())))
