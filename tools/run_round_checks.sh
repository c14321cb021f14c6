set -x
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/q/pytest.log
tail -5 gpurun_out/q/pytest.log
timeout 600 python bench.py > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; cat gpurun_out/q/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/q/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/q/stats.log 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/q/stats -name '*.db' | head -1) > $R/gpurun_out/q/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/q/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/q/fetch.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/q/fetch -name '*.db' | head -1) > $R/gpurun_out/q/pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/q/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/q/write.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/q/write -name '*.db' | head -1) > $R/gpurun_out/q/pmc_write.txt
rm -rf $R/gpurun_out/q/stats $R/gpurun_out/q/fetch $R/gpurun_out/q/write
head -12 $R/gpurun_out/q/kernel_stats.txt
