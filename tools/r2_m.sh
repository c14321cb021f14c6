mkdir -p gpurun_out/r2m
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2m/trace_base.txt
DV_CONV42_BLOCKS=2 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2m/trace_lb2.txt
python tools/compare_traces.py gpurun_out/r2m/trace_base.txt gpurun_out/r2m/trace_lb2.txt | grep "nb4\|total"
for v in 1 2 1 2; do DV_CONV42_BLOCKS=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH $v', d['value'], d['ms_per_step'])"; done
