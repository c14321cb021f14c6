#!/bin/bash
cp deepvariant_amd/libdvhip.so /tmp/orig.so
for v in "$@"; do
  cp build_variants/libdvhip_$v.so deepvariant_amd/libdvhip.so
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['roofline_encoder']; print('$v', round(e['avg_launch_ms'],4), round(e['achieved']))"
done
cp /tmp/orig.so deepvariant_amd/libdvhip.so
