#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06c}
timeout 1500 python -m pytest tests -m gpu -q -x -k "rnn or taps or tap or sharded or lstm or generic or host" 2>&1 | tail -15 > $OUT/pytest_$TAG.log
tail -5 $OUT/pytest_$TAG.log
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -c 2000 $OUT/bench_$TAG.json
