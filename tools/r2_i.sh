mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_hip_region_multisample.py tests/test_hip_stem_fused.py tests/test_hip_pipeline.py -q -x > gpurun_out/r2i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2i/pytest.log
tail -25 gpurun_out/r2i/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2i/bench.json')); print(d['value'], d['parity'], d['cpu_baseline'])"
