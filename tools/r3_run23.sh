# round 3, GPU run 23: where realign_table's time goes (wrapping its callees with timers from outside)
python - <<'PY'
import os, sys, time, json
sys.argv = ['bench.py', '--mode', 'bam']
sys.path.insert(0, os.getcwd())
from deepvariant_amd.realigner import realigner as R, window_selector as W, debruijn_graph as D
from deepvariant_amd import fast_pass_aligner as F, packing
acc = {}
def timed(label, fn):
  def w(*a, **k):
    t = time.perf_counter()
    try: return fn(*a, **k)
    finally: acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
  return w
W.select_windows = timed('select_windows (device counts + candidates)', W.select_windows)
W._candidates_to_windows = timed('  candidates -> windows', W._candidates_to_windows)
W.variant_reads_candidates_from_allele_counter = timed('  variant_reads counts from counter', W.variant_reads_candidates_from_allele_counter)
D.build_from_table = timed('debruijn build_from_table (native, in threads)', D.build_from_table)
F.FastPassAligner.align_reads_arrays = timed('aligner align_reads_arrays (native, in threads)', F.FastPassAligner.align_reads_arrays)
F.FastPassAligner.set_haplotypes = timed('aligner set_haplotypes', F.FastPassAligner.set_haplotypes)
F.FastPassAligner.set_reference = timed('aligner set_reference', F.FastPassAligner.set_reference)
packing.ReadTable.with_alignments = timed('with_alignments', packing.ReadTable.with_alignments)
packing.ReadTable.take = timed('take', packing.ReadTable.take)
R._map_in_order = timed('_map_in_order (both stages, wall)', R._map_in_order)
import runpy
try:
  runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
  pass
print(json.dumps({k: round(v * 1e3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}, indent=1))
PY
