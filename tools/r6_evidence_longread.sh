set -x
O=gpurun_out/r6
mkdir -p $O
bash tools/r6_run.sh tests bench workloads
( timeout 900 python tools/r5_cnn_tail.py --workload ont --n 2048 --seeds 101,202,303 --settings none,fast,product ; timeout 900 python tools/r5_cnn_tail.py --workload hifi --n 2048 --seeds 101,202,303 --settings none,fast,product ; timeout 1200 python tools/r5_cnn_tail.py --workload ont --n 65536 --seeds 202 --settings fast,product ) > $O/cnn_tail_longread.txt 2>> $O/cnn_tail.err
