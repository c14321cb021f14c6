mkdir -p gpurun_out/r2v
timeout 900 python -m pytest tests/test_hip_allelecounter.py -q -x > gpurun_out/r2v/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2v/pytest.log
tail -5 gpurun_out/r2v/pytest.log
timeout 600 python bench.py --mode alleles --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r2v/bench_alleles.json | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/r2v/bench_alleles.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved'], d['roofline']['bases_per_s'])"
