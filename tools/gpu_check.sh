#!/bin/bash
# Runs on the GPU box through gpurun: GPU test tier, bench, rocprofv3 kernel trace of the same bench command.
# Outputs land in gpurun_out/ (merged back); copy the summaries to profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
python bench.py --steps 10 --warmup 2 2>$OUT/bench.err | tee $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
find $OUT/prof_bench -name "*stats*" | head
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
