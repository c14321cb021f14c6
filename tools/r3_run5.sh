# round 3, GPU run 5: full GPU suite, default bench line with the new parity sample, per-stage error budget
set -x
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3e/pytest_gpu.log
tail -6 gpurun_out/r3e/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; python -c "import json;d=json.load(open('gpurun_out/r3e/bench.json'));print(d['value'],d['roofline']['frac'],d['parity'],d['cpu_baseline'])"
timeout 1500 python tools/r3_error_budget.py --n 1024 --seeds 17,29,43 > gpurun_out/r3e/error_budget.txt 2> gpurun_out/r3e/error_budget.err
cat gpurun_out/r3e/error_budget.txt; tail -3 gpurun_out/r3e/error_budget.err
