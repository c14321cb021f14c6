mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_hip_pipeline.py -q -x > gpurun_out/r2t/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2t/pytest.log
tail -30 gpurun_out/r2t/pytest.log
