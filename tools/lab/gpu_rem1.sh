#!/bin/bash
# one-MFMA remainder slabs in gru_l1f: recurrent test tier, then same-box A/B against the previous build, then kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "rnn or gru or recurrent or Rnn or GRU or lstm" 2>&1 | grep -v amdgpu.ids | tail -6
for rep in 1 2 3; do for lib in tools/lab/probes/libs/libturboae_prev.so in-tree; do
  if [ "$lib" = "in-tree" ]; then unset TAE_LIB; else export TAE_LIB=$R/$lib; fi
  echo "$(basename $lib .so): $(timeout 300 python tools/lab/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn 2>&1 | grep forward | sed 's/.*forward/forward/')" | tee -a $OUT/ab_rem1.txt
done; done
unset TAE_LIB
bash tools/lab/l0_trace.sh rem1 tools/lab/probes/libs/libturboae_prev.so in-tree
