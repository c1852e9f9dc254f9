#!/bin/bash
# A/B of the generic RNN kernel's blocks-per-workgroup (TAE_GEN_RNN_NB) on the GPU box; the sha of x_dec must not depend on it.
mkdir -p gpurun_out; out=gpurun_out/r04_gen_rnn_nb_ab.txt; : > $out
for cell in lstm rnn; do for B in 512 2048 16384; do for nb in 1 4 8; do
  TAE_GEN_RNN_NB=$nb timeout 600 python tools/quick_bench_rnn.py $cell $B 2>&1 | tail -1 | tee -a $out
done; done; done
timeout 1500 python -m pytest tests/test_gpu_generic.py -x -q -m gpu 2>&1 | tail -3 | tee -a $out
