mkdir -p gpurun_out/r2o
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -x > gpurun_out/r2o/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2o/pytest.log
tail -12 gpurun_out/r2o/pytest.log
DV_NO_POOL2_FUSE=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2o/trace_base.txt
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2o/trace_fuse.txt
grep -A3 "conv 3x3 s1 80->192" gpurun_out/r2o/trace_base.txt | tail -4;  grep -A3 "conv 3x3 s1 80->192" gpurun_out/r2o/trace_fuse.txt | tail -4
for v in 1 0 1 0; do if [ $v = 1 ]; then export DV_NO_POOL2_FUSE=1; else unset DV_NO_POOL2_FUSE; fi; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH nofuse=$v', d['value'], d['ms_per_step'])"; done
