import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepvariant_amd import dv_types as T, make_examples_core as mec, protowire as pw
from deepvariant_amd.realigner import utils as U
from tests import realigner_fixture as RF, test_oracle_golden as G
mode = 'rows'
ref, sets = RF.load()
meta, golden = G.load_alt_goldens(mode)
options = T.MakeExamplesOptions(pic_options=G.alt_pic_options(mode, True),
                                sample_options=[T.SampleOptions(role='main', name='NA12878', pileup_height=100)])
proc = mec.RegionProcessor(options, ref)
reads = sets['wgs']; spans = [U.read_range(r) for r in reads]
images = {}
for region in mec.partition(T.Range('chr20', 9_999_999, 10_010_000), 1000):
  _, encoded = proc.examples_in_region(region, [r for r, s in zip(reads, spans) if U.ranges_overlap(s, region)])
  for blob in encoded:
    ex = pw.decode_example(blob)
    v = pw.decode_variant(ex['variant/encoded'][0])
    idx = tuple(pw.decode_alt_allele_indices(ex['alt_allele_indices/encoded'][0]))
    images[(v.start, tuple(v.alternate_bases), idx)] = np.frombuffer(ex['image/encoded'][0], np.uint8).reshape(300, 221, 6)
for k, (start, end, refb, alts, idx) in enumerate(meta):
  a, b = images[(start, alts, idx)], golden[k]
  if not np.array_equal(a, b):
    rows = np.where((a != b).any(axis=(1, 2)))[0]
    print(k, start, refb, alts, idx, 'rows differ', rows[:12], len(rows), 'ours nonzero rows', [int(a[i*100:(i+1)*100].any(axis=(1,2)).sum()) for i in range(3)], 'gold', [int(b[i*100:(i+1)*100].any(axis=(1,2)).sum()) for i in range(3)])
