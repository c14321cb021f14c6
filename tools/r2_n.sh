mkdir -p gpurun_out/r2n
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2n/trace_base.txt
DV_CONV_PT=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r2n/trace_pt1.txt
python tools/compare_traces.py gpurun_out/r2n/trace_base.txt gpurun_out/r2n/trace_pt1.txt > gpurun_out/r2n/cmp.txt; cat gpurun_out/r2n/cmp.txt
