#!/bin/bash
# kernel-trace time of the GRU layer-0 recurrence for a set of library builds: bash tools/lab/l0_trace.sh <tag> in-tree <lib.so> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
tag=$1; shift
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = "in-tree" ]; then unset TAE_LIB; else export TAE_LIB=$R/$lib; fi
  d=/tmp/l0_${tag}_$(basename $lib .so)_$rep
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $d.log 2>&1 < /dev/null
  python $R/tools/trace_summary.py $d $d.txt > /dev/null < /dev/null
  echo "$(basename $lib .so) rep $rep: $(grep -E 'gru_rec_h_kernel<true>|gru_l1f' $d.txt | awk '{print $1, $5}' | tr '\n' ' ')" | tee -a $OUT/l0_trace_$tag.txt
done; done
