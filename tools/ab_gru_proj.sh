#!/bin/bash
# A/B of the GRU layer-1 projection kernel geometry / store policy on one box (bit-identical results):
#   TAE_GRU_PROJ_PG = 1 (4-wave workgroups, two per CU) | 2 (8-wave, one per CU);  TAE_GRU_PROJ_NT = 1 (non-temporal GI stores) | 0
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for pg in 1 2; do for nt in 1 0; do
  echo -n "PG=$pg NT=$nt: "; TAE_GRU_PROJ_PG=$pg TAE_GRU_PROJ_NT=$nt python tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep -v amdgpu | cut -c1-110
done; done; done
