mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_hip_host_pipeline.py -q -x > gpurun_out/r2j/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2j/pytest.log
tail -25 gpurun_out/r2j/pytest.log
timeout 600 python bench.py --mode host --steps 10 --warmup 2 > gpurun_out/r2j/bench_host.json 2> gpurun_out/r2j/bench_host.err; cat gpurun_out/r2j/bench_host.json; tail -3 gpurun_out/r2j/bench_host.err
