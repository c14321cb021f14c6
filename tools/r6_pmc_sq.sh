#!/bin/bash
# Round 6: SQ / TA / LDS counters per kernel over one full-size forward of the final build (three separate
# --pmc passes, --kernel-trace only; DV_NO_GRAPH=1 so that every launch is a dispatch rocprofv3 sees).
#   gpurun -- bash tools/r6_pmc_sq.sh [workload]        -> gpurun_out/r6/pmc_sq[_<workload>].txt
R=$PWD
O=$R/gpurun_out/r6
mkdir -p $O
W=${1:-illumina30}
SUF=""; [ "$W" != illumina30 ] && SUF=_$W
cd /tmp && export TMPDIR=/tmp
export DV_NO_GRAPH=1 DV_BENCH_NO_PMC=1
i=0
: > $O/pmc_sq$SUF.raw
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VALU" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-workloads --no-dense --calibration-images 0 > $O/pmc$i.log 2>&1
  python $R/profiles/summarize_pmc.py $(find $O/pmc$i -name '*.db' | head -1) >> $O/pmc_sq$SUF.raw 2>&1
  rm -rf $O/pmc$i
done
OT=$O/op_trace_raw.txt; [ "$W" != illumina30 ] && OT=$O/op_trace_$W.txt
python $R/tools/r6_pmc_table.py $O/pmc_sq$SUF.raw $OT > $O/pmc_sq$SUF.txt
cat $O/pmc_sq$SUF.txt
