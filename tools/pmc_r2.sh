# usage: tools/pmc_r2.sh <outdir-name>   (SQ / LDS counters per kernel, two passes)
R=$PWD
OUT=$R/gpurun_out/${1:-pmc_r2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DV_NO_GRAPH=1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -- python $R/bench.py --batch 1900 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  python $R/profiles/summarize_pmc.py $(find $OUT/p$i -name '*.db' | head -1) > $OUT/pmc$i.txt 2>&1
  rm -rf $OUT/p$i
done
