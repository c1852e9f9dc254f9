#!/bin/bash
# A/B of the MFMA recurrence's blocks per workgroup (TAE_GEN_RNN_NT = 1: 16, 2: 32) on the GPU box + the generic test tier + a kernel trace.
# HISTORICAL: ran on commit a67ac83; the 32-block instantiation (no gain, profiles/r04_gen_rnn_nt_ab.txt) and its knob were removed afterwards - the kernel keeps its NT template parameter.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_rnn_nt_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for nt in 1 2; do
  for cell in lstm rnn gru; do
    TAE_FORCE_GENERIC=1 TAE_GEN_RNN_NT=$nt timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=$cell 2>&1 | tail -1 | sed "s/^/nt=$nt /" | tee -a $out
  done
done
bash tools/prof_rnn.sh lstm 16384 r04_lstm_generic4 | head -8 | tee -a $out
