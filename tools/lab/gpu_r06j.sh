#!/bin/bash
# r06: final-tree verification after the fused LSTM / RNN layer 1: whole suite both orders + smoke, fuzz on the forced fused form, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06j}
bash tools/gpu_final.sh $TAG > /dev/null 2>&1
grep "passed\|failed\|smoke:" $OUT/pytest_final_$TAG.txt
{ echo "== fuzz_rnn_u 150 cases seed 41, layer 1 FORCED onto rnn_l1f_u_kernel (TAE_RNN_L1=fused)"; TAE_DEBUG_KNOBS=1 TAE_RNN_L1=fused timeout 1200 python tools/lab/probes/fuzz_rnn_u.py 150 41 2>&1 | grep -v amdgpu.ids | tail -160; } > $OUT/${TAG}_fuzz_rnn_u_fused.txt
tail -2 $OUT/${TAG}_fuzz_rnn_u_fused.txt; grep -c FAIL $OUT/${TAG}_fuzz_rnn_u_fused.txt
timeout 1200 python tools/determinism_soak.py 10 2>&1 | grep -v amdgpu.ids | tail -20 > $OUT/${TAG}_determinism_soak.txt; grep -c "differing values 0" $OUT/${TAG}_determinism_soak.txt; grep -v "differing values 0" $OUT/${TAG}_determinism_soak.txt
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 1900 $OUT/bench_$TAG.json
