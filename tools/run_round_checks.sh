set -x
mkdir -p gpurun_out/n
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/n/pytest.log
tail -5 gpurun_out/n/pytest.log
timeout 600 python bench.py > gpurun_out/n/bench.json 2> gpurun_out/n/bench.err; cat gpurun_out/n/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/n/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/n/stats.log 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/n/stats -name '*.db' | head -1) > $R/gpurun_out/n/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/n/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/n/fetch.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/n/fetch -name '*.db' | head -1) > $R/gpurun_out/n/pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/n/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/n/write.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/n/write -name '*.db' | head -1) > $R/gpurun_out/n/pmc_write.txt
rm -rf $R/gpurun_out/n/stats $R/gpurun_out/n/fetch $R/gpurun_out/n/write
head -12 $R/gpurun_out/n/kernel_stats.txt
