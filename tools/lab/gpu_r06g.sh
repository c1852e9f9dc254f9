#!/bin/bash
# r06: fused LSTM / RNN layer 1 - recurrent tests, A/B timing against the split form, kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06g}
timeout 1200 python -m pytest tests -m gpu -q -x -k "rnn or lstm or recurrent or generic or taps" 2>&1 | tail -15 > $OUT/pytest_$TAG.log
tail -5 $OUT/pytest_$TAG.log
for i in 1 2; do
  python tools/lab/quick_bench_rnn.py lstm 16384 | tail -1
  TAE_DEBUG_KNOBS=1 TAE_RNN_L1=split python tools/lab/quick_bench_rnn.py lstm 16384 | tail -1
done
python tools/lab/quick_bench_rnn.py rnn 16384 | tail -1
TAE_DEBUG_KNOBS=1 TAE_RNN_L1=split python tools/lab/quick_bench_rnn.py rnn 16384 | tail -1
python tools/lab/quick_bench_rnn.py lstm 500 | tail -1
TAE_DEBUG_KNOBS=1 TAE_RNN_L1=split python tools/lab/quick_bench_rnn.py lstm 500 | tail -1
bash tools/lab/prof_rnn.sh lstm 16384 ${TAG}_lstm > /dev/null 2>&1
head -7 $OUT/${TAG}_lstm_by_grid.txt
