#!/bin/bash
# A/B of the generic recurrence kernel (fp32 MFMA vs vector ALU, TAE_GEN_RNN=valu) on the GPU box + the generic test tier.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_rnn_mfma_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -5 | tee -a $out
for mode in valu mfma; do
  for cell in lstm rnn; do for B in 512 16384; do
    TAE_GEN_RNN=$mode timeout 300 python tools/quick_bench_any.py $B decoder=TurboAE_rate3_rnn dec_rnn=$cell 2>&1 | tail -1 | sed "s/^/rnn=$mode /" | tee -a $out
  done; done
  TAE_GEN_RNN=$mode timeout 300 python tools/quick_bench_any.py 4096 encoder=TurboAE_rate3_rnn decoder=TurboAE_rate3_rnn enc_rnn=lstm dec_rnn=lstm enc_num_layer=3 2>&1 | tail -1 | sed "s/^/rnn=$mode /" | tee -a $out
done
