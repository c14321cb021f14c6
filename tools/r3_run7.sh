# round 3, GPU run 7: chain.hip generalised to the 35x35 stage (256-pixel tiles, 3x3 / 5x5): parity, A/B, per-launch table
set -x
mkdir -p gpurun_out/r3g
timeout 1200 python -m pytest tests/test_hip_chain.py -x -q > gpurun_out/r3g/pytest_chain.log 2>&1; echo "rc=$?" >> gpurun_out/r3g/pytest_chain.log
tail -12 gpurun_out/r3g/pytest_chain.log
for mode in all pairs none; do
  if [ $mode = pairs ]; then export DV_CHAIN2D_MIN_LEN=2; elif [ $mode = none ]; then unset DV_CHAIN2D_MIN_LEN; export DV_NO_CHAIN2D=1; fi
  DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_$mode.json 2> gpurun_out/r3g/bench_$mode.err; python -c "import json;d=json.load(open('gpurun_out/r3g/bench_$mode.json'));print('$mode',d['value'],d['roofline']['frac'])"
done
unset DV_NO_CHAIN2D
DV_OP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3g/op_trace.err
grep "dv-op" gpurun_out/r3g/op_trace.err | tail -60 > gpurun_out/r3g/op_trace.txt
grep -E "chain|total|imgconv" gpurun_out/r3g/op_trace.txt
