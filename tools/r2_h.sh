mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_hip_stem_fused.py -q -s -x > gpurun_out/r2h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2h/pytest.log
grep -v "^$" gpurun_out/r2h/pytest.log | tail -4
DV_STEM_PROF=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2h/trace.json 2> gpurun_out/r2h/trace.err
grep "dv-stem-b\|stem_\|total" gpurun_out/r2h/trace.err | tail -5
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"; done
