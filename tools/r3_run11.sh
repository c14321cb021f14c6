# round 3, GPU run 11: real-data line, packed-record cache + lazily built Reads; --procs sweep (JSON = last stdout line)
set -x
O=gpurun_out/r3k
mkdir -p $O
for R in 1 8 16 32; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_$R.out 2> $O/bam_$R.err; tail -1 $O/bam_$R.out > $O/bam_$R.json; python -c "
import json;d=json.load(open('$O/bam_$R.json'));print($R, round(d['value'],1), round(d['wall_s'],3), round(d.get('examples_per_s_region_loop_only'),1), d.get('setup_s_max_over_ranks', d.get('setup_s')), d.get('stage_ms'))"
done
