mkdir -p gpurun_out/nb6
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/nb6/base.txt
DV_NB6=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/nb6/nb6.txt
python tools/compare_traces.py gpurun_out/nb6/base.txt gpurun_out/nb6/nb6.txt > gpurun_out/nb6/cmp.txt
grep -v "  0.0 %\|+0\.[0-9] %\|-0\.[0-9] %" gpurun_out/nb6/cmp.txt | head -70
