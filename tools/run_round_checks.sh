set -x
mkdir -p gpurun_out/p
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/p/pytest.log
tail -5 gpurun_out/p/pytest.log
timeout 600 python bench.py > gpurun_out/p/bench.json 2> gpurun_out/p/bench.err; cat gpurun_out/p/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p/stats.log 2>&1
python $R/profiles/summarize_rocpd.py $(find $R/gpurun_out/p/stats -name '*.db' | head -1) > $R/gpurun_out/p/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/p/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p/fetch.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/p/fetch -name '*.db' | head -1) > $R/gpurun_out/p/pmc_fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/p/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p/write.log 2>&1
python $R/profiles/summarize_pmc.py $(find $R/gpurun_out/p/write -name '*.db' | head -1) > $R/gpurun_out/p/pmc_write.txt
rm -rf $R/gpurun_out/p/stats $R/gpurun_out/p/fetch $R/gpurun_out/p/write
head -12 $R/gpurun_out/p/kernel_stats.txt
