# NOTE: the DV_CHAIN_STAGGER knob this script sweeps was removed after the run (no effect; profiles/r03_chain2d_ab.txt)
# round 3, GPU run 12: do the chain kernels' output stores leave as one burst?  DV_CHAIN_STAGGER = cycles between the
# starts of 8 phase groups of workgroups; bench A/B + the chain phase profile with and without
set -x
O=gpurun_out/r3l
mkdir -p $O
for S in 0 1500 3000 6000 0 3000; do
DV_CHAIN_STAGGER=$S DV_BENCH_NO_PMC=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_$S.json 2> $O/bench_$S.err; python -c "import json;d=json.load(open('$O/bench_$S.json'));print($S, d['value'],d['roofline']['frac'],d['parity']['ok'])"
done
for S in 0 3000; do
DV_CHAIN_STAGGER=$S DV_CHAIN_PROF=1 DV_NO_GRAPH=1 DV_OP_TRACE=1 DV_BENCH_NO_PMC=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/prof_$S.txt
grep "wave 0" $O/prof_$S.txt | tail -9
done
