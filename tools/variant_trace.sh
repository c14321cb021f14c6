#!/bin/bash
# usage: tools/variant_trace.sh NAME...   (build_variants/libdvhip_NAME.so) -> gpurun_out/trace_NAME.txt
cp deepvariant_amd/libdvhip.so /tmp/orig.so
for v in "$@"; do
  cp build_variants/libdvhip_$v.so deepvariant_amd/libdvhip.so
  DV_OP_TRACE=1 DV_NO_GRAPH=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/trace_$v.txt > /dev/null
  echo "$v $(python bench.py --steps 10 --warmup 2 --no-cpu-baseline | cut -c50-110)"
done
cp /tmp/orig.so deepvariant_amd/libdvhip.so
DV_OP_TRACE=1 DV_NO_GRAPH=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/trace_base.txt > /dev/null
echo "base $(python bench.py --steps 10 --warmup 2 --no-cpu-baseline | cut -c50-110)"
