mkdir -p gpurun_out/r2x
timeout 900 python -m pytest tests/test_hip_inception.py -q -x > gpurun_out/r2x/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2x/pytest.log
tail -3 gpurun_out/r2x/pytest.log
for v in 1 0 1 0; do if [ $v = 1 ]; then export DV_NO_BAND_NB6=1; else unset DV_NO_BAND_NB6; fi; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH no_band_nb6=$v', d['value'], d['ms_per_step'])"; done
