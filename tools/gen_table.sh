#!/bin/bash
# The generic path before / after its move onto the fp32 matrix cores, one box: r03 kernels (TAE_GEN_CONV=valu TAE_GEN_RNN=valu
# TAE_GEN_RNN_NB=1) vs the defaults, one forward each (tools/quick_bench_any.py).  -> gpurun_out/r04_generic_table.txt
mkdir -p gpurun_out; out=gpurun_out/r04_generic_table.txt; : > $out
run() { timeout 600 python tools/quick_bench_any.py "$@" 2>&1 | tail -1; }
for mode in r03 r04; do
  if [ $mode = r03 ]; then export TAE_GEN_CONV=valu TAE_GEN_RNN=valu TAE_GEN_RNN_NB=1; else unset TAE_GEN_CONV TAE_GEN_RNN TAE_GEN_RNN_NB; fi
  { run 16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm
    run 16384 decoder=TurboAE_rate3_rnn dec_rnn=rnn
    TAE_FORCE_GENERIC=1 run 16384 decoder=TurboAE_rate3_rnn
    run 4096 encoder=TurboAE_rate3_rnn decoder=TurboAE_rate3_rnn enc_rnn=lstm dec_rnn=lstm enc_num_layer=3
    run 2048 enc_num_unit=136 dec_num_unit=136
    run 2048 enc_num_unit=256 dec_num_unit=256
    run 2048 enc_kernel_size=11 dec_kernel_size=11
    run 64 enc_kernel_size=11 dec_kernel_size=11
    run 2048 num_iter_ft=9
    run 2048 encoder=TurboAE_rate3_cnn_dense decoder=TurboAE_rate3_cnn_dense precision=f32; } | sed "s/^/$mode: /" | tee -a $out
done
