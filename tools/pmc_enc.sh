R=$PWD
cd /tmp && export TMPDIR=/tmp
export DV_NO_GRAPH=1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pe$i -- python $R/bench.py --batch 1900 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pe$i.log 2>&1
  python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/pe$i -name '*.db' | head -1) 2>&1 | grep -E "encode_items|Counter" > $R/gpurun_out/pe$i.txt
  rm -rf $R/gpurun_out/pe$i
done
