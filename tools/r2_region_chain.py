"""Per-stage wall clock of the region chain (make_examples_core.RegionProcessor) on the two golden
fixtures, one process, one GPU: where the time of a real calling region goes once every stage
is this repo's own.  Usage: python tools/r2_region_chain.py  (on a GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deepvariant_amd import dv_types as T
from deepvariant_amd import make_examples_core as mec
from deepvariant_amd.realigner import utils as U
from tests import pacbio_chain as PC
from tests import realigner_fixture as RF
from tests.golden.make_golden import wgs_options


class Clock:
  def __init__(self):
    self.t = {}

  def wrap(self, obj, name, label):
    inner = getattr(obj, name)

    def timed(*a, **k):
      t0 = time.perf_counter()
      try:
        return inner(*a, **k)
      finally:
        self.t[label] = self.t.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)


def run(title, proc, region, size, reads, repeat=2):
  spans = [U.read_range(r) for r in reads]
  for it in range(repeat):
    clock = Clock()
    clock.wrap(proc, 'realign_reads', 'realign')
    clock.wrap(proc, 'candidates_in_region', 'count+call(+phase)')
    clock.wrap(proc.generator, 'encode_region', 'encode (pack + kernel + tf.Example)')
    if proc.direct_phasing is not None:
      clock.wrap(proc.direct_phasing, 'phase', '  of which phasing')
    t0 = time.perf_counter()
    n_regions = n_reads = n_cands = n_examples = 0
    for r in mec.partition(region, size):
      in_reads = [x for x, s in zip(reads, spans) if U.ranges_overlap(s, r)]
      cands, examples = proc.examples_in_region(r, in_reads)
      n_regions += 1
      n_reads += len(in_reads)
      n_cands += len(cands)
      n_examples += len(examples)
    total = time.perf_counter() - t0
    for name in ('realign_reads', 'candidates_in_region'):
      setattr(proc, name, getattr(type(proc), name).__get__(proc))
    proc.generator.encode_region = type(proc.generator).encode_region.__get__(proc.generator)
    if proc.direct_phasing is not None:
      proc.direct_phasing.phase = type(proc.direct_phasing).phase.__get__(proc.direct_phasing)
    if it == repeat - 1:
      print('%s: %d regions, %d reads, %d candidates, %d examples in %.3f s (%.1f examples/s, %.0f reads/s)' % (
          title, n_regions, n_reads, n_cands, n_examples, total, n_examples / total, n_reads / total))
      for k, v in clock.t.items():
        print('    %-40s %7.1f ms  %5.1f %%' % (k, 1e3 * v, 100 * v / total))


def main():
  ref, sets = RF.load()
  options = T.MakeExamplesOptions(pic_options=wgs_options(),
                                  sample_options=[T.SampleOptions(role='main', name='NA12878', pileup_height=100)])
  run('Illumina WGS golden (realigner on, 1 kb regions)', mec.RegionProcessor(options, ref),
      T.Range('chr20', 9_999_999, 10_010_000), 1000, sets['wgs'])
  pref, preads, _, _ = PC.load()
  poptions = T.MakeExamplesOptions(pic_options=PC.pic_options(True), trim_reads_for_pileup=True,
                                   sample_options=[T.SampleOptions(role='main', name='s', pileup_height=100)])
  po = mec.RegionProcessorOptions(realigner_enabled=False, vsc_min_fraction_indels=0.12, track_ref_reads=True,
                                  phase_reads=True, partition_size=PC.PARTITION)
  run('PacBio golden (phasing, alt-aligned, 25 kb regions)', mec.RegionProcessor(poptions, pref, po), PC.REGION,
      PC.PARTITION, preads)


if __name__ == '__main__':
  main()
