# round-2 final evidence in ONE GPU call: GPU tests, smoke, default bench line, per-launch table,
# rocprofv3 kernel stats and the two HBM-traffic counter passes (separate --pmc runs)
set -x
O=gpurun_out/r2final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; cat $O/bench_default.json
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace_raw.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/stats.log 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/$O/stats -name '*.db' | head -1) > $R/$O/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/fetch.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/$O/fetch -name '*.db' | head -1) > $R/$O/pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/write.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/$O/write -name '*.db' | head -1) > $R/$O/pmc_write.txt
rm -rf $R/$O/stats $R/$O/fetch $R/$O/write
head -16 $R/$O/kernel_stats.txt
