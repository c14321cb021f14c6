#!/bin/bash
# The WHOLE library with its host code under AddressSanitizer + UndefinedBehaviorSanitizer (hipcc instruments the host
# pass of every .hip / .cpp translation unit; device code is built as usual), in a scratch copy of the sources, and the
# whole CPU test suite on it (deepvariant_amd/_lib.py loads the library named by DV_LIB_PATH).  Reports go to
# $WORK/san.<pid>.  Development tool; the product library is untouched.
#   tools/host_asan/run_full.sh [pytest arguments...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WORK=${DV_ASAN_WORK:-/tmp/dv_full_asan}
rm -rf "$WORK" && mkdir -p "$WORK/deepvariant_amd"
cp -r "$ROOT/deepvariant_amd/csrc" "$WORK/deepvariant_amd/csrc" && cp -r "$ROOT/include" "$WORK/include"
rm -f "$WORK"/deepvariant_amd/csrc/*.o
make -s -C "$WORK/deepvariant_amd/csrc" -j8 EXTRA="-fsanitize=address,undefined -fno-omit-frame-pointer -g -Wno-option-ignored" > "$WORK/build.log" 2>&1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
DV_LIB_PATH="$WORK/deepvariant_amd/libdvhip.so" LD_PRELOAD="$RT" \
  ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0:log_path="$WORK/san" \
  UBSAN_OPTIONS=print_stacktrace=1:log_path="$WORK/san" \
  python -m pytest ${@:-tests/ -m "not gpu" -n 4} -q -p no:cacheprovider 2>&1 | tail -3
echo "sanitizer report files: $(ls "$WORK"/san.* 2>/dev/null | wc -l)"
cat "$WORK"/san.* 2>/dev/null | grep -E "runtime error|ERROR: AddressSanitizer" | sort | uniq -c | sort -rn | head -20
