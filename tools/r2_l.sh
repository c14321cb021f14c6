mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -x > gpurun_out/r2l/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2l/pytest.log
tail -12 gpurun_out/r2l/pytest.log
DV_IMGCONV_RING=2 DV_NO_BAND=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2l/trace_noband.txt
DV_IMGCONV_RING=2 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2l/trace_band.txt
for i in 1 2; do DV_IMGCONV_RING=2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'])"; done
