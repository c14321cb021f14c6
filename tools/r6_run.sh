#!/bin/bash
# One parameterised GPU-box script for round 6:  gpurun -- bash tools/r6_run.sh <step> [<step> ...]   (outputs: gpurun_out/r6/)
# steps: tail:<settings> | tests | tests:<file-or-expr> | smoke | bench | trace | stats | pmc | workloads | ab:<ENV=VAL> | line:<label>:<ENV=VAL,...>
set -x
O=gpurun_out/r6
mkdir -p $O
R=$PWD
bench_line() {  # label, env...
  local label=$1; shift
  env "$@" DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-workloads 2>> $O/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$label', round(d['value']), round(d['ms_per_step'],3), 'conv', round(d['roofline']['ms_per_step'],3), 'other', round(d['other_kernels_ms_per_step'],3), 'enc', round(d['roofline_encoder']['avg_launch_ms'],3))" | tee -a $O/ab.txt
}
for step in "$@"; do
  case $step in
    tail:*) timeout 1500 python tools/r5_cnn_tail.py --settings ${step#tail:} > $O/cnn_tail_${step#tail:}.txt 2> $O/cnn_tail.err; echo "rc=$?" >> $O/cnn_tail.err; cat $O/cnn_tail_${step#tail:}.txt; tail -3 $O/cnn_tail.err ;;
    tests) timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log ;;
    tests:*) timeout 1500 python -m pytest ${step#tests:} -m gpu -q -s > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log; grep -E "seed|hifi|ont|max|passed|failed|rc=|Error|assert" $O/pytest_sel.log | tail -30 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log ;;
    bench) DV_BENCH_PMC_SAVE=$O/pmc_hbm_traffic.txt timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench.err; cat $O/bench_default.json ;;
    trace:*) rest=${step#trace:}; label=${rest%%:*}; kvs=${rest#*:}; [ "$kvs" = "$rest" ] && kvs=DV_X=0
       env ${kvs//,/ } DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads > /dev/null 2> $O/op_trace_$label.txt; grep -c dv-op $O/op_trace_$label.txt ;;
    wl:*) rest=${step#wl:}; w=${rest%%:*}; rest=${rest#*:}; label=${rest%%:*}; kvs=${rest#*:}; [ "$kvs" = "$rest" ] && kvs=DV_X=0
       env ${kvs//,/ } DV_BENCH_NO_PMC=1 timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>> $O/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w $label', round(d['value']), round(d['ms_per_step'],3), 'conv', round(d['roofline']['ms_per_step'],3), 'other', round(d['other_kernels_ms_per_step'],3), 'enc', round(d['roofline_encoder']['avg_launch_ms'],3), 'enc_frac', round(d['roofline_encoder']['frac'],3))" | tee -a $O/ab.txt ;;
    stemprof:*) rest=${step#stemprof:}; label=${rest%%:*}; kvs=${rest#*:}; [ "$kvs" = "$rest" ] && kvs=DV_X=0
       env ${kvs//,/ } DV_STEM_PROF=1 DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-workloads --calibration-images 0 2>&1 > /dev/null | grep -E "dv-stem-b|stem_b conv" | tail -6 | sed "s/^/$label /" | tee -a $O/stemprof.txt ;;
    trace) DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --no-dense > /dev/null 2> $O/op_trace_raw.txt; grep -c dv-op $O/op_trace_raw.txt
           DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads --dense-only > /dev/null 2> $O/op_trace_dense_raw.txt ;;
    densestats) cd /tmp && export TMPDIR=/tmp
       DV_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_d -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-workloads --dense-only --calibration-images 0 > $R/$O/stats_d.log 2>&1
       python $R/profiles/summarize_rocpd.py $(find $R/$O/stats_d -name '*.db' | head -1) > $R/$O/kernel_stats_dense.txt
       rm -rf $R/$O/stats_d; cd $R; head -12 $O/kernel_stats_dense.txt ;;
    stats) cd /tmp && export TMPDIR=/tmp
       DV_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-workloads --no-dense --calibration-images 0 > $R/$O/stats.log 2>&1
       python $R/profiles/summarize_rocpd.py $(find $R/$O/stats -name '*.db' | head -1) > $R/$O/kernel_stats.txt
       rm -rf $R/$O/stats; cd $R; head -20 $O/kernel_stats.txt ;;
    pmc) bash tools/r6_pmc_sq.sh > $O/pmc_sq.log 2>&1; tail -30 $O/pmc_sq.txt ;;
    wstats) for w in hifi35 ont50; do
         ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_$w -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-dense --calibration-images 0 > $R/$O/stats_$w.log 2>&1
           python $R/profiles/summarize_rocpd.py $(find $R/$O/stats_$w -name '*.db' | head -1) > $R/$O/kernel_stats_$w.txt; rm -rf $R/$O/stats_$w )
       done ;;
    workloads) for w in hifi35 ont50; do
         timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; cat $O/bench_$w.json
         DV_OP_TRACE=1 timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-dense > /dev/null 2> $O/op_trace_$w.txt
         ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_$w -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-dense --calibration-images 0 > $R/$O/stats_$w.log 2>&1
           python $R/profiles/summarize_rocpd.py $(find $R/$O/stats_$w -name '*.db' | head -1) > $R/$O/kernel_stats_$w.txt; rm -rf $R/$O/stats_$w )
       done ;;
    ab:*) kv=${step#ab:}; for r in 1 2; do bench_line default; bench_line "$kv" "$kv"; done ;;
    line:*) rest=${step#line:}; label=${rest%%:*}; kvs=${rest#*:}; if [ "$kvs" = "$rest" ]; then bench_line "$label"; else bench_line "$label" ${kvs//,/ }; fi ;;
    *) echo "unknown step $step" ;;
  esac
done
