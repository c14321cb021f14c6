#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment tools/r3_run*.sh):
#   gpurun -- bash tools/r4_run.sh <step> [<step> ...]      outputs under gpurun_out/r4/
# steps: precision | tests | smoke | bench | trace | stats | bam | ab:<ENV=VAL> | workloads
set -x
O=gpurun_out/r4
mkdir -p $O
R=$PWD
bench_line() {  # label, env...
  local label=$1; shift
  env "$@" DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>> $O/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$label', round(d['value']), round(d['ms_per_step'],3), 'conv', round(d['roofline']['ms_per_step'],3), 'other', round(d['other_kernels_ms_per_step'],3), 'parity', d.get('parity'))" | tee -a $O/ab.txt
}
for step in "$@"; do
  case $step in
    precision) timeout 900 python -m pytest tests/test_hip_precision.py -q -s > $O/precision.log 2>&1; echo "rc=$?" >> $O/precision.log; grep -E "seed|rms|passed|failed|rc=|Error|assert" $O/precision.log | tail -20 ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log ;;
    bench) DV_BENCH_PMC_SAVE=$O/pmc_hbm_traffic.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; cat $O/bench_default.json ;;
    trace) DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace_raw.txt; grep -c dv-op $O/op_trace_raw.txt ;;
    stats) cd /tmp && export TMPDIR=/tmp
       DV_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/stats.log 2>&1
       python $R/profiles/summarize_rocpd.py $(find $R/$O/stats -name '*.db' | head -1) > $R/$O/kernel_stats.txt
       rm -rf $R/$O/stats; cd $R; head -20 $O/kernel_stats.txt ;;
    bam) timeout 300 python bench.py --mode bam > $O/bench_bam.json 2> $O/bench_bam.err; cat $O/bench_bam.json ;;
    workloads) for w in hifi35 ont50; do
         timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; cat $O/bench_$w.json
         DV_OP_TRACE=1 timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace_$w.txt
         ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_$w -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/stats_$w.log 2>&1
           python $R/profiles/summarize_rocpd.py $(find $R/$O/stats_$w -name '*.db' | head -1) > $R/$O/kernel_stats_$w.txt; rm -rf $R/$O/stats_$w )
       done ;;
    ab:*) kv=${step#ab:}; for r in 1 2; do bench_line default; bench_line "$kv" "$kv"; done ;;
    *) echo "unknown step $step" ;;
  esac
done
