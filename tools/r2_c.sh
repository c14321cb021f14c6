set -x
DV_IMGCONV_TAPS=9,25,7 bash tools/pmc_r2.sh r2c
grep -E "imgconv_kernelILi3ELi3ELi1ELi3|imgconv_kernelILi7ELi1ELi1ELi3|stem_b|stem_a|conv_mfma_kernelILi3ELi2|conv_mfma_kernelILi4ELi2" gpurun_out/r2c/pmc1.txt | sort
grep -E "imgconv_kernelILi3ELi3ELi1ELi3|imgconv_kernelILi7ELi1ELi1ELi3|stem_b|stem_a|conv_mfma_kernelILi3ELi2|conv_mfma_kernelILi4ELi2" gpurun_out/r2c/pmc2.txt | sort
