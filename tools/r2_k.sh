mkdir -p gpurun_out/r2k
timeout 600 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py -q -x > gpurun_out/r2k/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2k/pytest.log
tail -4 gpurun_out/r2k/pytest.log
DV_IMGCONV_RING=2 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2k/trace_ring2.txt
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2k/trace_ring3.txt
DV_IMGCONV_MINP=40 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2k/trace_r3p40.txt
DV_IMGCONV_MINP=40 DV_IMGCONV_1X1=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2k/trace_r3all.txt
python tools/compare_traces.py gpurun_out/r2k/trace_ring2.txt gpurun_out/r2k/trace_ring3.txt gpurun_out/r2k/trace_r3p40.txt gpurun_out/r2k/trace_r3all.txt > gpurun_out/r2k/cmp.txt 2>&1; tail -3 gpurun_out/r2k/cmp.txt
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'])"; done
