# round 3, GPU run 24: window counts from the counter's event arrays + vectorised with_alignments: tests, real-data line
set -x
O=gpurun_out/r3zd
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_allelecounter.py tests/test_hip_pipeline.py tests/test_hip_realigner.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
for i in 1 2; do
  timeout 400 python bench.py --mode bam --procs 1 > $O/bam_$i.out 2> $O/bam_$i.err; tail -1 $O/bam_$i.out > $O/bam_$i.json; python -c "
import json;d=json.load(open('$O/bam_$i.json'));print(round(d['value'],1), round(d['examples_per_s_region_loop_only'],1), {k[:12]: round(v) for k, v in d['stage_ms'].items()})"
done
for R in 8 12; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_p$R.out 2> $O/bam_p$R.err; tail -1 $O/bam_p$R.out > $O/bam_p$R.json; python -c "
import json;d=json.load(open('$O/bam_p$R.json'));print($R, round(d['value'],1), round(d['wall_s'],3), round(d.get('examples_per_s_region_loop_only'),1))"
done
