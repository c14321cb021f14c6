# round 3, GPU run 13: chain-input tensors without a halo (DV_CHAIN_KEEP_HALO=1 = the previous layout): parity tests, A/B
set -x
O=gpurun_out/r3n
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_chain.py tests/test_hip_inception.py tests/test_hip_stem_fused.py tests/test_hip_resident.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
for K in 1 0; do
if [ $K = 1 ]; then export DV_CHAIN_KEEP_HALO=1; else unset DV_CHAIN_KEEP_HALO; fi
DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_${K}_$i.json 2> $O/bench_${K}_$i.err; python -c "import json;d=json.load(open('$O/bench_${K}_$i.json'));print('keep_halo=$K', round(d['value']), round(d['roofline']['frac'],4))"
done
done
unset DV_CHAIN_KEEP_HALO
DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/op_trace.txt
grep "dv-op" $O/op_trace.txt | awk '/total/{n++} n==3' | grep -E "total|768->|chain 1x7|chain 7x1|chain 3x3|256->|288->64\+" | cut -c1-120
