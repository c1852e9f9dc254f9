#!/bin/bash
# r06 first GPU visit: full suite with the recurrent deviation log, LSTM kernel trace + counters, GRU trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
rm -f $OUT/rnn_deviations.jsonl
TAE_DEVIATION_LOG=$OUT/rnn_deviations.jsonl timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/pytest_r06a.log
tail -6 $OUT/pytest_r06a.log
bash tools/lab/prof_rnn.sh lstm 16384 r06a_lstm
cat $OUT/r06a_lstm_by_grid.txt | head -12
bash tools/lab/pmc_rnn.sh lstm r06a_lstm > /dev/null 2>&1
grep "rnn_proj_u" $OUT/pmc_r06a_lstm.txt
