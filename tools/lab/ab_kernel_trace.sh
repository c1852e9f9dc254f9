#!/bin/bash
# Per-kernel average times (rocprofv3 --kernel-trace) of the GRU-decoder configuration for several library builds on one box:
#   bash tools/lab/ab_kernel_trace.sh <tag> in-tree tools/lab/probes/libs/libturboae_<name>.so ...      (timing-experiment builds: results wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; tag=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "in-tree" ]; then unset TAE_LIB; else export TAE_LIB=$R/$lib; fi
  d=/tmp/abk_$(basename $lib .so)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $d.log 2>&1 < /dev/null)
  python $R/tools/trace_summary.py $d 2>/dev/null | grep 'gru_l1f\|gru_rec_h' | awk -v l=$(basename $lib .so) '{printf "%-22s %-30s avg %s ms (min %s)\n", l, $1, $4, $5}' | tee -a gpurun_out/abk_$tag.txt
done
