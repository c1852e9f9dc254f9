#!/bin/bash
# A/B of the GRU decoder (cfg 5 shape, 16 384 blocks) between the in-tree library and a variant: bash tools/ab_gru.sh <variant.so>
cd ${GRAFT_REPO_ROOT:-.}
V=$PWD/turboae_amd/lib/variants/$1
for i in 1 2 3; do
  echo "tree   : $(python tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-90)"
  echo "variant: $(TAE_LIB=$V python tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-90)"
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -k "rnn or gru" 2>&1 | tail -2
