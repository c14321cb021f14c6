mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests/test_hip_allelecounter.py -q -x > gpurun_out/r2s/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2s/pytest.log
tail -30 gpurun_out/r2s/pytest.log
