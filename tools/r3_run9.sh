# round 3, GPU run 9: the real-data line with R host processes sharing the GPU (make_examples --ranks_per_gpu R)
set -x
O=gpurun_out/r3i
mkdir -p $O
for R in 1 4 8 16; do
  timeout 400 python bench.py --mode bam --procs $R > $O/bam_$R.json 2> $O/bam_$R.err; tail -c 600 $O/bam_$R.json; tail -3 $O/bam_$R.err
done
