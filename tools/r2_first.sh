set -x
mkdir -p gpurun_out/r2a
timeout 600 python -m pytest tests/test_hip_stem_fused.py -x -q > gpurun_out/r2a/pytest_stem.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/pytest_stem.log
tail -40 gpurun_out/r2a/pytest_stem.log
timeout 600 python -m pytest tests/test_hip_inception.py -q > gpurun_out/r2a/pytest_incep.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/pytest_incep.log
tail -15 gpurun_out/r2a/pytest_incep.log
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/trace.json 2> gpurun_out/r2a/trace.err
grep "dv-op" gpurun_out/r2a/trace.err | tail -82 | head -12
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_fused.json 2> gpurun_out/r2a/bench_fused.err; cat gpurun_out/r2a/bench_fused.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSED', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"
DV_NO_STEM_FUSE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_plain.json 2> gpurun_out/r2a/bench_plain.err; cat gpurun_out/r2a/bench_plain.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PLAIN', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"
