#!/bin/bash
# HISTORICAL (r04): the TAE_GRU_PK code path was removed after this measurement (no gain); the script documents how profiles/r04_gru_pk_ab.txt was made.
# A/B of the packed-fp32 gate arithmetic in gru_rec_h (-DTAE_GRU_PK=1 build in tools/probes/libs) against the in-tree library on one
# box: GRU-decoder forward at 16 384 blocks, alternating; then the GRU tests on the variant.  -> gpurun_out/r04_gru_pk_ab.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out; out=gpurun_out/r04_gru_pk_ab.txt; : > $out
V=$PWD/tools/probes/libs/libturboae_grupk.so
for rep in 1 2 3; do
  python tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | tail -1 | cut -c1-110 | sed "s/^/in-tree: /" | tee -a $out
  TAE_LIB=$V python tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | tail -1 | cut -c1-110 | sed "s/^/pk:      /" | tee -a $out
done
TAE_LIB=$V timeout 1200 python -m pytest tests -x -q -m gpu -k "gru or rnn or GRU" 2>&1 | tail -3 | tee -a $out
