#!/bin/bash
# Round 6 evidence set from the final build (GPU box):  gpurun -- bash tools/r6_evidence.sh   -> gpurun_out/r6/, copied to profiles/r06_*
set -x
O=gpurun_out/r6
mkdir -p $O
bash tools/r6_run.sh tests smoke bench trace stats densestats workloads
timeout 900 python tools/r5_cnn_tail.py --n 65536 --seeds 101,202,303 --settings none,product > $O/cnn_tail.txt 2> $O/cnn_tail.err
( timeout 900 python tools/r5_cnn_tail.py --workload ont --n 2048 --seeds 101,202,303 --settings none,fast,product ; timeout 900 python tools/r5_cnn_tail.py --workload hifi --n 2048 --seeds 101,202,303 --settings none,fast,product ; timeout 1200 python tools/r5_cnn_tail.py --workload ont --n 65536 --seeds 202 --settings fast,product ) > $O/cnn_tail_longread.txt 2>> $O/cnn_tail.err
bash tools/r6_pmc_sq.sh > $O/pmc_sq.log 2>&1
for p in 1 2 4 8; do timeout 600 python bench.py --mode bam --procs $p --repeat 10 2> $O/bench_bam_procs$p.err | grep '^{' > $O/bench_bam_procs$p.json; tail -c 300 $O/bench_bam_procs$p.json; done
tail -5 $O/cnn_tail_longread.txt
