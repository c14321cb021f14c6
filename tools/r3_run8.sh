# round 3, GPU run 8b: encoder with the two-stage row pipeline: encoder parity tests first (bounded), then the suite, bench
set -x
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_hip_known_answers.py tests/test_hip_synthetic.py tests/test_hip_fuzz.py tests/test_hip_golden.py -x -q > gpurun_out/r3h/pytest_enc.log 2>&1; echo "rc=$?" >> gpurun_out/r3h/pytest_enc.log
tail -5 gpurun_out/r3h/pytest_enc.log
if grep -q "rc=0" gpurun_out/r3h/pytest_enc.log; then
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3h/pytest_gpu.log
tail -5 gpurun_out/r3h/pytest_gpu.log
for i in 1 2; do
DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3h/bench_$i.json 2> gpurun_out/r3h/bench_$i.err; python -c "import json;d=json.load(open('gpurun_out/r3h/bench_$i.json'));print(d['value'],d['roofline']['frac'],d['roofline_encoder']['avg_launch_ms'],d['roofline_encoder']['frac'])"
done
fi
