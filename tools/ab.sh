#!/bin/bash
# usage: tools/ab.sh A B [rounds]  -- alternates build_variants/libdvhip_{A,B}.so, prints value / conv ms / other ms
cp deepvariant_amd/libdvhip.so /tmp/orig.so
for r in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp build_variants/libdvhip_$v.so deepvariant_amd/libdvhip.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['ms_per_step'],3), round(d['other_kernels_ms_per_step'],3))"
  done
done
cp /tmp/orig.so deepvariant_amd/libdvhip.so
