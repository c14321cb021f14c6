mkdir -p gpurun_out/r2ab
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2> gpurun_out/r2ab/$label.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_step'],3), round(d['other_kernels_ms_per_step'],3))"
}
for r in 1 2; do
  run base DV_X=1
  run slab4 DV_CONV_SLAB4=1
  run w8 DV_CONV_W8=1
done
# bit-identity of the variants against the default kernels (the tests compare every path with the plain one)
DV_CONV_SLAB4=1 python -m pytest tests/test_hip_stem_fused.py tests/test_hip_inception.py -q -x 2>&1 | tail -2
DV_CONV_W8=1 python -m pytest tests/test_hip_stem_fused.py tests/test_hip_inception.py -q -x 2>&1 | tail -2
