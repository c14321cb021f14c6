# round 3, GPU run 10: where the real-data region loop spends its host time (cProfile of bench.py --mode bam), and the
# --procs sweep again with lazily built Read objects and setup / loop times split
set -x
O=gpurun_out/r3j
mkdir -p $O
python - > $O/bam_profile.txt 2>&1 <<'PY'
import cProfile, pstats, sys, io, contextlib
sys.argv = ['bench.py', '--mode', 'bam']
import runpy
pr = cProfile.Profile()
buf = io.StringIO()
pr.enable()
try:
  runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
  pass
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats('cumulative').print_stats(70)
st.sort_stats('tottime').print_stats(40)
PY
head -5 $O/bam_profile.txt | cut -c1-400
for R in 1 4 8; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_$R.json 2> $O/bam_$R.err; python -c "
import json;d=json.load(open('$O/bam_$R.json'));print($R, d['value'], d['wall_s'], d.get('examples_per_s_region_loop_only'), d.get('setup_s_max_over_ranks', d.get('setup_s')))"
done
