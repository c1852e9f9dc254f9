#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{ echo "== fuzz_rnn_u 160 cases seed 31 (every 4th: GRU encoder in front)"; timeout 1200 python tools/lab/probes/fuzz_rnn_u.py 160 31 2>&1 | grep -v amdgpu.ids | tail -170; } > $OUT/r06_fuzz_rnn_u.txt
tail -3 $OUT/r06_fuzz_rnn_u.txt; grep -c FAIL $OUT/r06_fuzz_rnn_u.txt
{ echo "== determinism soak, 15 repeats"; timeout 1200 python tools/determinism_soak.py 15 2>&1 | grep -v amdgpu.ids | tail -25; } > $OUT/r06_determinism_soak.txt
cat $OUT/r06_determinism_soak.txt
{ SEEDS="404 505" bash tools/gpu_soak.sh; } > $OUT/r06_soak.txt 2>&1
cat $OUT/r06_soak.txt
