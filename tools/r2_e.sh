mkdir -p gpurun_out/r2e
DV_STEM_PROF=1 DV_OP_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/trace.json 2> gpurun_out/r2e/trace.err
grep "dv-stem-b\|stem_b" gpurun_out/r2e/trace.err | tail -6
