mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_hip_region_multisample.py -q -x > gpurun_out/r2p/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2p/pytest.log
tail -30 gpurun_out/r2p/pytest.log
