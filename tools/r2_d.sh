set -x
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -s -x > gpurun_out/r2d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest.log
grep -v "^$" gpurun_out/r2d/pytest.log | tail -12
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/trace.json 2> gpurun_out/r2d/trace.err
grep "dv-op" gpurun_out/r2d/trace.err | tail -80 | grep -v avgpool
for v in A B A B; do
if [ $v = A ]; then export DV_NO_IMGCONV=; unset DV_NO_IMGCONV; else export DV_NO_IMGCONV=1; fi
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v(A=imgconv,B=v1)', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'])"
done
