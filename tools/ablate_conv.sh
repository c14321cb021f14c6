# Timing ablations of conv_mfma_kernel (results of the ablated builds are WRONG by construction; only
# their per-launch times mean anything).  Build each variant with
#   make -C deepvariant_amd/csrc EXTRA="-DDV_ABLATE_X" && cp deepvariant_amd/libdvhip.so build_variants/libdvhip_ablX.so
# (touch model.cc / rm model.o between variants), restore the normal build, then run this on the GPU box.
#   ablX      no pixel-operand buffer_loads in the K loop        ablW    no weight ds_reads per chunk
#   ablXW     neither (MFMAs + loop control + slab copy only)     ablLOOP no K loop at all (prologue + epilogue)
#   ablEPI    no output stores                                    ablLOOPEPI  prologue only
mkdir -p gpurun_out/abl
DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/abl/base.txt
V="ablX ablW ablXW ablLOOP ablEPI ablLOOPEPI"
for v in $V; do
  [ -f build_variants/libdvhip_$v.so ] || continue
  DV_LIB_PATH=$PWD/build_variants/libdvhip_$v.so DV_OP_TRACE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/abl/$v.txt
done
python tools/compare_traces.py gpurun_out/abl/base.txt $(for v in $V; do [ -f gpurun_out/abl/$v.txt ] && echo gpurun_out/abl/$v.txt; done) --all > gpurun_out/abl/cmp.txt
grep -E "^op|conv |total" gpurun_out/abl/cmp.txt | grep -v imgconv
