#!/bin/bash
# A/B of the XCD-aware tile order of the generic conv / projection kernels (TAE_GEN_XCD=0: launch order) + the generic test tier.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_xcd_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for rep in 1 2; do for f in 0 1; do
  for cfg in "16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm" "2048 enc_num_unit=256 dec_num_unit=256" "2048 enc_kernel_size=11 dec_kernel_size=11"; do
    TAE_GEN_XCD=$f timeout 300 python tools/quick_bench_any.py $cfg 2>&1 | tail -1 | sed "s/^/xcd=$f /" | tee -a $out
  done
done; done
TAE_GEN_XCD=1 bash tools/prof_rnn.sh lstm 16384 r04_lstm_generic8 | head -6 | tee -a $out
