#!/bin/bash
# same-box A/B of two library builds on the three recurrent decoders at 16 384 blocks (alternating rounds, x_dec hash per run):
#   bash tools/lab/ab_y0_libs.sh <tag> <libA.so> <libB.so>
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; tag=$1; shift
for cell in gru lstm rnn; do for rep in 1 2 3; do for lib in "$@"; do
  export TAE_LIB=$PWD/$lib
  echo "$cell $(basename $lib .so): $(timeout 300 python tools/lab/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=$cell 2>&1 | grep forward | sed 's/.*forward/forward/')" | tee -a gpurun_out/ab_$tag.txt
done; done; done
