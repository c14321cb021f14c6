#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [<file.hip> ...]: a second libdvhip in which the
# named translation units are compiled with the extra flags -> build_variants/libdvhip_<name>.so
# (A/B runs on one box: DV_LIB_PATH=build_variants/libdvhip_<name>.so python bench.py ...)
set -e
name=$1; flags=$2; shift 2
cd /root/repo/deepvariant_amd/csrc
make -s -j8
mkdir -p /root/repo/build_variants /tmp/variant_$name
objs=""
for o in *.o; do
  src=${o%.o}.hip
  if [[ " $* " == *" $src "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I. -Wno-unused-function -Wno-unused-result $flags -c $src -o /tmp/variant_$name/$o
    objs="$objs /tmp/variant_$name/$o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_variants/libdvhip_$name.so $objs -lz -ldl
ls -la /root/repo/build_variants/libdvhip_$name.so
