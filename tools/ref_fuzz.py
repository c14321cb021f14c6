"""Long differential fuzz: oracle/encoder_oracle.cpp (the restatement) against oracle/_ref/libdvref.so (the reference's
own encoder sources, compiled unmodified: oracle/ref_build/) over every channel set of tests/fuzz_inputs.py.
TEST INFRASTRUCTURE, CPU only, needs /root/reference (or a prebuilt oracle/_ref).  `python tools/ref_fuzz.py [seeds] [config,config...]`;
the committed test (tests/test_reference_encoder_cpu.py) runs 12 seeds per set, this runs thousands.
Round 4: 6000 seeds x 8 channel sets = 48,000 pile-ups, 0 differing (profiles/r04_reference_fuzz.txt)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests import fuzz_inputs as FZ
from deepvariant_amd import dv_types as T
t0 = time.time()
n = bad = 0
def both(fn):
  with O.reference_backend():
    r = fn()
  return r, fn()
ALL = FZ.CONFIGS + FZ.ULTIMA_CONFIGS + FZ.SAMPLE_PROBABILITY_CONFIGS      # (the last two lists: round 4)
ONLY = sys.argv[2].split(',') if len(sys.argv) > 2 else None
for (name, channels, width, height, okw, ckw) in ALL:
  if ONLY and name not in ONLY:
    continue
  opts = FZ.options(channels, width, height, **dict(okw))
  enums = [O.channel_str_to_enum(c) for c in channels]
  for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6000):
    rng = np.random.default_rng(777000 + 1000 * len(name) + seed)
    depth = int(rng.choice([0, 1, 2, 5, 12, 30, height - 5, height, height + 25, 3 * height]))
    call, ref_window, reads, image_start, combo = FZ.make_case(rng, width, depth, **dict(ckw))
    if 'avg_base_quality' in channels:
      for r in reads:
        r.aligned_quality = bytes(min(q, 93) for q in r.aligned_quality)
    blank = [enums[int(rng.integers(0, len(enums)))]] if seed % 4 == 3 else None
    positions = [int(r.alignment.position.position) - int(rng.integers(0, 30)) for r in reads] if seed % 5 == 4 else None
    kw = dict(pileup_height=(height if seed % 3 else 0), mean_coverage=float(rng.integers(0, 60)), alignment_positions=positions, channels_to_blank=blank)
    if seed % 7 == 6:      # SampleOptions.use_non_uniform_downsampling, thresholds that fit and thresholds that fall back
      kw['non_uniform_downsampling_threshold'] = int(rng.choice([0, 1, 3, 12, 60]))
    try:
      a, b = both(lambda: O.build_pileup(opts, call, ref_window, reads, image_start, combo, **kw))
    except Exception as e:
      print('EXC', name, seed, repr(e)[:200]); bad += 1; continue
    n += 1
    if not np.array_equal(a, b):
      bad += 1
      print('DIFF', name, seed, np.argwhere(a != b)[:3].tolist())
  print(name, 'done', n, 'bad', bad, '%.0fs' % (time.time() - t0), flush=True)
print('TOTAL', n, 'bad', bad)
