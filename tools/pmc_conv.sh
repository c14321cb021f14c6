R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TD_[A-Z_0-9]+)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/counters.txt
export DV_NO_GRAPH=1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc$i -- python $R/bench.py --batch 1900 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc$i.log 2>&1
  python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/pmc$i -name '*.db' | head -1) > $R/gpurun_out/pmc$i.txt 2>&1
  rm -rf $R/gpurun_out/pmc$i
done
