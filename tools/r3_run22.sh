# round 3, GPU run 22: the region chain's table path (no Read objects): goldens, object path == table path, real-data line
set -x
O=gpurun_out/r3za
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_realigner.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
for K in 1 0 1 0; do
if [ $K = 1 ]; then export DV_REGION_OBJECTS=1; else unset DV_REGION_OBJECTS; fi
  timeout 400 python bench.py --mode bam --procs 1 > $O/bam_o$K.out 2> $O/bam_o$K.err; tail -1 $O/bam_o$K.out > $O/bam_o$K.json; python -c "
import json;d=json.load(open('$O/bam_o$K.json'));print('objects=$K', round(d['value'],1), round(d['examples_per_s_region_loop_only'],1), {k[:12]: round(v) for k, v in d['stage_ms'].items()})"
done
unset DV_REGION_OBJECTS
for R in 8 16; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_$R.out 2> $O/bam_$R.err; tail -1 $O/bam_$R.out > $O/bam_$R.json; python -c "
import json;d=json.load(open('$O/bam_$R.json'));print($R, round(d['value'],1), round(d['wall_s'],3), round(d.get('examples_per_s_region_loop_only'),1))"
done
