# round 3, GPU run 18: lazily built Read objects (packing.LazyRead): GPU suite parts that run the region chain, real-data line
set -x
O=gpurun_out/r3s
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_realigner.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for R in 1 8 16; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_$R.out 2> $O/bam_$R.err; tail -1 $O/bam_$R.out > $O/bam_$R.json; python -c "
import json;d=json.load(open('$O/bam_$R.json'));print($R, round(d['value'],1), round(d['wall_s'],3), round(d.get('examples_per_s_region_loop_only'),1), d.get('setup_s_max_over_ranks', d.get('setup_s')), d.get('stage_ms'))"
done
