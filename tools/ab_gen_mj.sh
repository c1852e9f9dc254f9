#!/bin/bash
# A/B of the taps staged together by gen_conv_mfma_kernel (TAE_GEN_MJ = 4 in-tree: 107 KB of LDS at k = 5, one workgroup per CU;
# 2: 73 KB, two; 1: 55 KB, two) on the GPU box.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_mj_ab.txt; : > $out
for rep in 1 2; do for lib in in-tree mj2 mj1; do
  if [ $lib = in-tree ]; then unset TAE_LIB; else export TAE_LIB=$PWD/tools/probes/libs/libturboae_$lib.so; fi
  for cfg in "2048 enc_num_unit=256 dec_num_unit=256" "2048 enc_num_unit=136 dec_num_unit=136" "2048 enc_kernel_size=11 dec_kernel_size=11"; do
    timeout 300 python tools/quick_bench_any.py $cfg 2>&1 | tail -1 | sed "s/^/$lib: /" | tee -a $out
  done
done; done
