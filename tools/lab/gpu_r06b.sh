#!/bin/bash
# r06: full suite with the recurrent deviation log, LSTM kernel trace + counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06b}
rm -f $OUT/rnn_deviations.jsonl
TAE_DEVIATION_LOG=$OUT/rnn_deviations.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest_$TAG.log
tail -6 $OUT/pytest_$TAG.log
bash tools/lab/prof_rnn.sh lstm 16384 ${TAG}_lstm
head -7 $OUT/${TAG}_lstm_by_grid.txt
bash tools/lab/pmc_rnn.sh lstm ${TAG}_lstm > /dev/null 2>&1
grep "rnn_proj_u" $OUT/pmc_${TAG}_lstm.txt
