#!/bin/bash
# r06: final-tree verification after the Y0 re-layout: whole suite both orders + smoke, recurrent fuzz (both layer-1 forms), determinism soak, bench + kernel traces
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06r}
bash tools/gpu_final.sh $TAG > /dev/null 2>&1
grep "passed\|failed\|smoke:" $OUT/pytest_final_$TAG.txt
{ echo "== fuzz_rnn_u 120 cases seed 43"; timeout 1200 python tools/lab/probes/fuzz_rnn_u.py 120 43 2>&1 | grep -v amdgpu.ids | tail -130;
  echo "== fuzz_rnn_u 80 cases seed 44, layer 1 FORCED onto the fused kernels (TAE_RNN_L1=fused)"; TAE_DEBUG_KNOBS=1 TAE_RNN_L1=fused timeout 1200 python tools/lab/probes/fuzz_rnn_u.py 80 44 2>&1 | grep -v amdgpu.ids | tail -90; } > $OUT/${TAG}_fuzz_rnn_u.txt
grep -c FAIL $OUT/${TAG}_fuzz_rnn_u.txt; grep "worst\|cases" $OUT/${TAG}_fuzz_rnn_u.txt | tail -4
timeout 1200 python tools/determinism_soak.py 10 2>&1 | grep -v amdgpu.ids | tail -20 > $OUT/${TAG}_determinism_soak.txt; grep -c "differing values 0" $OUT/${TAG}_determinism_soak.txt; grep -v "differing values 0" $OUT/${TAG}_determinism_soak.txt
PROF_CFG=1 bash tools/gpu_round.sh $TAG none 2>&1 | tail -40
