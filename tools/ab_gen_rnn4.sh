#!/bin/bash
# A/B of the first-layer input projection fused into the MFMA recurrence (TAE_GEN_RNN_FUSE=0: GI through HBM) + the generic test tier.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_rnn_fuse_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for f in 0 1; do
  for cell in lstm rnn; do
    TAE_GEN_RNN_FUSE=$f timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=$cell 2>&1 | tail -1 | sed "s/^/fuse=$f /" | tee -a $out
  done
  TAE_FORCE_GENERIC=1 TAE_GEN_RNN_FUSE=$f timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn 2>&1 | tail -1 | sed "s/^/fuse=$f forced-generic gru /" | tee -a $out
done
bash tools/prof_rnn.sh lstm 16384 r04_lstm_generic5 | head -8 | tee -a $out
