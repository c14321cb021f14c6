# round 3: chain.hip after the epilogue / set-up / 8+7 changes: parity, bench A/B, phase profile
set -x
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_hip_chain.py -x -q > gpurun_out/r3d/pytest_chain.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/pytest_chain.log
tail -5 gpurun_out/r3d/pytest_chain.log
DV_CHAIN_REGS=1 timeout 900 python -m pytest tests/test_hip_chain.py -x -q > gpurun_out/r3d/pytest_chain_regs.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/pytest_chain_regs.log
tail -3 gpurun_out/r3d/pytest_chain_regs.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_$i.json 2> gpurun_out/r3d/bench.err; python -c "import json;d=json.load(open('gpurun_out/r3d/bench_$i.json'));print('dma',d['value'],d['roofline']['frac'])"
DV_CHAIN_REGS=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_regs_$i.json 2> gpurun_out/r3d/bench_regs.err; python -c "import json;d=json.load(open('gpurun_out/r3d/bench_regs_$i.json'));print('regs',d['value'],d['roofline']['frac'])"
done
DV_NO_CHAIN=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_nochain.json 2> gpurun_out/r3d/bench_nochain.err; python -c "import json;d=json.load(open('gpurun_out/r3d/bench_nochain.json'));print('nochain',d['value'],d['roofline']['frac'])"
DV_OP_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3d/op_trace.err
grep "dv-op" gpurun_out/r3d/op_trace.err | tail -72 > gpurun_out/r3d/op_trace.txt
grep -E "chain|total" gpurun_out/r3d/op_trace.txt
DV_OP_TRACE=1 DV_CHAIN_PROF=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3d/prof.err
grep -E "dv-chain|chain " gpurun_out/r3d/prof.err | tail -45 > gpurun_out/r3d/prof.txt
tail -20 gpurun_out/r3d/prof.txt
