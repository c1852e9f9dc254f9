#!/bin/bash
# rocprofv3 kernel trace of one generic-RNN configuration: bash tools/lab/prof_rnn.sh <cell> <B> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
d=/tmp/prof_rnn_$3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/lab/quick_bench_rnn.py $1 $2 > $d.log 2>&1 < /dev/null
tail -1 $d.log
python $R/tools/trace_summary.py $d $OUT/$3_by_grid.txt < /dev/null
