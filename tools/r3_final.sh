# round-3 evidence in ONE GPU call: GPU tests, smoke, default bench line (with its own two counter passes),
# per-launch table, rocprofv3 kernel stats, the real-data line
set -x
O=gpurun_out/r3final
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
DV_BENCH_PMC_SAVE=$O/pmc_hbm_traffic.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; cat $O/bench_default.json
DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace_raw.txt
timeout 300 python bench.py --mode bam > $O/bench_bam.json 2> $O/bench_bam.err; cat $O/bench_bam.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
DV_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/stats.log 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/$O/stats -name '*.db' | head -1) > $R/$O/kernel_stats.txt
rm -rf $R/$O/stats
head -16 $R/$O/kernel_stats.txt
