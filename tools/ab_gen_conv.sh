#!/bin/bash
# A/B of the generic convolution kernel (fp32 MFMA vs the r03 vector-ALU kernel, TAE_GEN_CONV=valu) on the GPU box + the generic test tier.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_conv_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -5 | tee -a $out
for mode in valu mfma; do
  TAE_GEN_CONV=$mode timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm 2>&1 | tail -1 | tee -a $out
  TAE_GEN_CONV=$mode timeout 300 python tools/quick_bench_any.py 2048 enc_num_unit=136 dec_num_unit=136 2>&1 | tail -1 | tee -a $out
  TAE_GEN_CONV=$mode timeout 300 python tools/quick_bench_any.py 2048 enc_num_unit=256 dec_num_unit=256 2>&1 | tail -1 | tee -a $out
  TAE_GEN_CONV=$mode timeout 300 python tools/quick_bench_any.py 2048 dec_kernel_size=11 enc_kernel_size=11 2>&1 | tail -1 | tee -a $out
  TAE_GEN_CONV=$mode timeout 300 python tools/quick_bench_any.py 64 dec_kernel_size=11 enc_kernel_size=11 2>&1 | tail -1 | tee -a $out
done
