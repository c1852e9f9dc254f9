#!/bin/bash
# r06: fused LSTM / RNN layer 1 (one N tile, four steps per chunk) - recurrent tests, A/B timing, kernel trace + counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06h}
timeout 1500 python -m pytest tests -m gpu -q -x -k "rnn or lstm or recurrent or generic or taps or gru or fullsize" 2>&1 | tail -15 > $OUT/pytest_$TAG.log
tail -5 $OUT/pytest_$TAG.log
bash tools/lab/prof_rnn.sh lstm 16384 ${TAG}_lstm > /dev/null 2>&1
head -7 $OUT/${TAG}_lstm_by_grid.txt
bash tools/lab/pmc_rnn.sh lstm ${TAG}_lstm > /dev/null 2>&1
grep "l1f" $OUT/pmc_${TAG}_lstm.txt
