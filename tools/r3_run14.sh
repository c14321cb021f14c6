# NOTE: gemm1x1_kernel and its DV_NO_GEMM1X1 knob were removed after this run (slower; profiles/r03_gemm1x1_experiment.txt)
# round 3, GPU run 14: gemm1x1_kernel (grouped 1x1 heads, both operands through LDS): bit-identity, per-launch table, A/B
set -x
O=gpurun_out/r3o
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_chain.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace.txt
grep "dv-op" $O/op_trace.txt | awk '/total/{n++} n==3' | grep -E "total|1x1" | cut -c1-150
for i in 1 2; do
for K in 1 0; do
if [ $K = 1 ]; then export DV_NO_GEMM1X1=1; else unset DV_NO_GEMM1X1; fi
DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_${K}_$i.json 2> $O/bench_${K}_$i.err; python -c "import json;d=json.load(open('$O/bench_${K}_$i.json'));print('no_gemm=$K', round(d['value']), round(d['roofline']['frac'],4), d.get('parity',{}).get('ok'))"
done
done
