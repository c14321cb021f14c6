# NOTE: avgpool3s1_planes_kernel and the DV_AVGPOOL_V1 knob were removed after this run (slower; profiles/r03_chain2d_ab.txt)
# round 3, GPU run 15: average pools through LDS (avgpool3s1_planes_kernel) against the per-output kernel (DV_AVGPOOL_V1=1)
set -x
O=gpurun_out/r3p
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_inception.py tests/test_hip_chain.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace.txt
grep "dv-op" $O/op_trace.txt | awk '/total/{n++} n==3' | grep -E "total|avgpool" | cut -c1-120
for i in 1 2; do
for K in 1 0; do
if [ $K = 1 ]; then export DV_AVGPOOL_V1=1; else unset DV_AVGPOOL_V1; fi
DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_${K}_$i.json 2> $O/bench_${K}_$i.err; python -c "import json;d=json.load(open('$O/bench_${K}_$i.json'));print('v1=$K', round(d['value']), round(d['roofline']['frac'],4), round(d['other_kernels_ms_per_step'],3))"
done
done
