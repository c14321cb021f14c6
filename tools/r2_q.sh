mkdir -p gpurun_out/r2q
P=$PWD/deepvariant_amd/libdvhip_prev.so
DV_LIB_PATH=$P DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2q/trace_prev.txt
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2q/trace_new.txt
python tools/compare_traces.py gpurun_out/r2q/trace_prev.txt gpurun_out/r2q/trace_new.txt > gpurun_out/r2q/cmp.txt; tail -1 gpurun_out/r2q/cmp.txt
for v in prev new prev new; do if [ $v = prev ]; then export DV_LIB_PATH=$P; else unset DV_LIB_PATH; fi; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH $v', d['value'], d['ms_per_step'])"; done
unset DV_LIB_PATH
timeout 600 python -m pytest tests/test_hip_inception.py -q -x 2>&1 | tail -2
