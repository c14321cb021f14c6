mkdir -p gpurun_out/r2bl
python -m pytest tests/test_hip_blank_skip.py -x -q 2>&1 | tail -15
run() { label=$1; shift
  env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2> gpurun_out/r2bl/$label.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_step'],3), round(d['other_kernels_ms_per_step'],3))"
}
for r in 1 2; do
  run base DV_X=1
  run blank DV_BLANK_SKIP=1
done
DV_BLANK_SKIP=1 DV_OP_TRACE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "dv-op" | head -8
