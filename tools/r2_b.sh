set -x
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -s > gpurun_out/r2b/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest.log
grep -v "^$" gpurun_out/r2b/pytest.log | tail -40
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2b/trace.json 2> gpurun_out/r2b/trace.err
grep "dv-op" gpurun_out/r2b/trace.err | tail -80
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_v2.json 2> gpurun_out/r2b/bench_v2.err; cat gpurun_out/r2b/bench_v2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V2', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"
DV_NO_IMGCONV=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_v1.json 2> gpurun_out/r2b/bench_v1.err; cat gpurun_out/r2b/bench_v1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V1', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"
