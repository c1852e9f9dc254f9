#!/bin/bash
# A/B/C of GRU-decoder builds (BASELINE configs[4] shape, 16 384 blocks) on one box, alternating; x_dec sha printed per run:
#   bash tools/lab/ab_gru_libs.sh <tag> in-tree tools/lab/probes/libs/libturboae_<name>.so ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; tag=$1; shift
for rep in 1 2 3; do
for lib in "$@"; do
  if [ "$lib" = "in-tree" ]; then unset TAE_LIB; else export TAE_LIB=$PWD/$lib; fi
  echo "$(basename $lib .so): $(timeout 300 python tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-120)" | tee -a gpurun_out/ab_$tag.txt
done; done
