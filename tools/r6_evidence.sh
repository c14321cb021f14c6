set -x
bash tools/r6_run.sh tests smoke bench stats densestats workloads
O=gpurun_out/r6
timeout 900 python tools/r5_cnn_tail.py --n 65536 --seeds 101,202,303 --settings product > $O/cnn_tail.txt 2> $O/cnn_tail.err
( timeout 900 python tools/r5_cnn_tail.py --workload ont --n 2048 --seeds 101,202,303 --settings none,product ; timeout 900 python tools/r5_cnn_tail.py --workload hifi --n 2048 --seeds 101,202,303 --settings none,product ; timeout 900 python tools/r5_cnn_tail.py --workload ont --n 65536 --seeds 202 --settings product ) > $O/cnn_tail_longread.txt 2>> $O/cnn_tail.err
bash tools/r6_pmc_sq.sh > $O/pmc_sq.log 2>&1
tail -5 $O/cnn_tail_longread.txt
