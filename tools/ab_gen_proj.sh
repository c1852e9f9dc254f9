#!/bin/bash
# A/B of the pipelined k = 1 kernel of the generic path (TAE_GEN_PROJ=0: the general conv kernel) + the generic test tier + kernel trace.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_proj_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for f in 0 1 0 1; do
  TAE_GEN_PROJ=$f timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm 2>&1 | tail -1 | sed "s/^/proj=$f /" | tee -a $out
done
bash tools/prof_rnn.sh lstm 16384 r04_lstm_generic7 | head -8 | tee -a $out
