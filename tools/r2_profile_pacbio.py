import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepvariant_amd import dv_types as T, make_examples_core as mec
from deepvariant_amd.realigner import utils as U
from tests import pacbio_chain as PC
pref, preads, _, _ = PC.load()
poptions = T.MakeExamplesOptions(pic_options=PC.pic_options(True), trim_reads_for_pileup=True,
                                 sample_options=[T.SampleOptions(role='main', name='s', pileup_height=100)])
po = mec.RegionProcessorOptions(realigner_enabled=False, vsc_min_fraction_indels=0.12, track_ref_reads=True,
                                phase_reads=True, partition_size=PC.PARTITION)
proc = mec.RegionProcessor(poptions, pref, po)
spans = [U.read_range(r) for r in preads]
def go():
  for r in mec.partition(PC.REGION, PC.PARTITION):
    proc.examples_in_region(r, [x for x, s in zip(preads, spans) if U.ranges_overlap(s, r)])
go()
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
