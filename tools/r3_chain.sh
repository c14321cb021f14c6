# round 3: chain.hip phase profile (s_memtime per phase) for both loaders + parity of this round's other GPU-side changes
set -x
mkdir -p gpurun_out/r3c
DV_OP_TRACE=1 DV_CHAIN_PROF=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3c/prof_regs.err
grep -E "dv-chain|chain " gpurun_out/r3c/prof_regs.err | tail -50 > gpurun_out/r3c/prof_regs.txt
DV_CHAIN_DMA=1 DV_OP_TRACE=1 DV_CHAIN_PROF=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r3c/prof_dma.err
grep -E "dv-chain|chain " gpurun_out/r3c/prof_dma.err | tail -50 > gpurun_out/r3c/prof_dma.txt
cat gpurun_out/r3c/prof_regs.txt
timeout 900 python -m pytest tests/test_hip_chain.py tests/test_hip_inception.py tests/test_hip_allelecounter.py tests/test_hip_blank_skip.py -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/pytest.log
tail -8 gpurun_out/r3c/pytest.log
