# round 3: first GPU run of the fused branch chains (chain.hip): parity, bench, per-launch table
set -x
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_hip_chain.py -x -q > gpurun_out/r3a/pytest_chain.log 2>&1; echo "rc=$?" >> gpurun_out/r3a/pytest_chain.log
tail -15 gpurun_out/r3a/pytest_chain.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; cat gpurun_out/r3a/bench.json
DV_NO_CHAIN=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_nochain.json 2> gpurun_out/r3a/bench_nochain.err; cat gpurun_out/r3a/bench_nochain.json
DV_OP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3a/op_trace.err
grep "dv-op" gpurun_out/r3a/op_trace.err | tail -80 > gpurun_out/r3a/op_trace.txt
grep -E "chain|total" gpurun_out/r3a/op_trace.txt
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_stem_fused.py tests/test_hip_resident.py -x -q > gpurun_out/r3a/pytest_cnn.log 2>&1; echo "rc=$?" >> gpurun_out/r3a/pytest_cnn.log
tail -5 gpurun_out/r3a/pytest_cnn.log
