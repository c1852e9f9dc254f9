#!/bin/bash
# A/B of the GI layout of the generic MFMA recurrence (TAE_GEN_GI_GROUP=0: (block, position) rows; default: 16 blocks interleaved) + tests.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_gi_group_ab.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
for f in 0 1 0 1; do
  TAE_GEN_GI_GROUP=$f timeout 300 python tools/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm 2>&1 | tail -1 | sed "s/^/group=$f /" | tee -a $out
done
bash tools/prof_rnn.sh lstm 16384 r04_lstm_generic6 | head -7 | tee -a $out
