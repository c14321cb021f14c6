mkdir -p gpurun_out/r2u
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2u/trace_base.txt
DV_CU_PAIR=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2u/trace_pair.txt
python tools/compare_traces.py gpurun_out/r2u/trace_base.txt gpurun_out/r2u/trace_pair.txt > gpurun_out/r2u/cmp.txt; grep "1x1\|total" gpurun_out/r2u/cmp.txt
for v in 0 1 0 1; do DV_CU_PAIR=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH pair=$v', d['value'], d['ms_per_step'])"; done
