# round 3, GPU run 6: chain.hip with the pre-barrier fragment prefetch; default bench line with same-run PMC traffic; bench --mode bam
set -x
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_hip_chain.py -x -q > gpurun_out/r3f/pytest_chain.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/pytest_chain.log
tail -3 gpurun_out/r3f/pytest_chain.log
DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3f/bench_quick.json 2> gpurun_out/r3f/bench_quick.err; python -c "import json;d=json.load(open('gpurun_out/r3f/bench_quick.json'));print('quick',d['value'],d['roofline']['frac'])"
timeout 900 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; python -c "import json;d=json.load(open('gpurun_out/r3f/bench.json'));print(d['value'],d['roofline'],d['roofline_encoder']['traffic'])"; tail -3 gpurun_out/r3f/bench.err
timeout 900 python bench.py --mode bam > gpurun_out/r3f/bench_bam.json 2> gpurun_out/r3f/bench_bam.err; cat gpurun_out/r3f/bench_bam.json; tail -5 gpurun_out/r3f/bench_bam.err
DV_OP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3f/op_trace.err
grep "dv-op" gpurun_out/r3f/op_trace.err | tail -66 > gpurun_out/r3f/op_trace.txt
grep -E "chain|total" gpurun_out/r3f/op_trace.txt
