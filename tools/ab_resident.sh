mkdir -p gpurun_out/r2rs
python -m pytest tests/test_hip_resident.py -x -q 2>&1 | tail -15
run() { label=$1; shift
  env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2> gpurun_out/r2rs/$label.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_step'],3), round(d['other_kernels_ms_per_step'],3))"
}
for r in 1 2; do
  run generic DV_RESIDENT=0
  run conv4 DV_RESIDENT=1
  run all96 DV_RESIDENT=2
done
for mode in 0 2; do
DV_RESIDENT=$mode DV_OP_TRACE=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "dv-op" > gpurun_out/r2rs/trace_$mode.txt
done
python tools/compare_traces.py gpurun_out/r2rs/trace_0.txt gpurun_out/r2rs/trace_2.txt 2>/dev/null | head -70 || paste gpurun_out/r2rs/trace_0.txt gpurun_out/r2rs/trace_2.txt | head -70
