#!/bin/bash
# GPU-box visit for the fused GRU layer-1 kernel: GRU test tier, A/B timing (fused vs TAE_GRU_L1=split), kernel trace.
# usage: bash tools/lab/gpu_l1f.sh <tag> [tests: 1|0]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-l1f}; TESTS=${2:-1}
mkdir -p $OUT; cd $R
if [ "$TESTS" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "rnn or gru" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/pytest_$TAG.log
fi
for rep in 1 2; do
  echo "fused: $(timeout 300 python tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-110)" | tee -a $OUT/ab_$TAG.txt
  echo "split: $(TAE_DEBUG_KNOBS=1 TAE_GRU_L1=split timeout 300 python tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-110)" | tee -a $OUT/ab_$TAG.txt
done
echo "fused B=500: $(timeout 300 python tools/lab/quick_bench_cfg.py 100 500 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-110)" | tee -a $OUT/ab_$TAG.txt
echo "split B=500: $(TAE_DEBUG_KNOBS=1 TAE_GRU_L1=split timeout 300 python tools/lab/quick_bench_cfg.py 100 500 2 TurboAE_rate3_rnn 2>&1 | grep forward | cut -c1-110)" | tee -a $OUT/ab_$TAG.txt
cd /tmp; export TMPDIR=/tmp
d=/tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $d.log 2>&1 < /dev/null
tail -1 $d.log
python $R/tools/trace_summary.py $d $OUT/${TAG}_by_grid.txt < /dev/null
