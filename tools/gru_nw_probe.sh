#!/bin/bash
# How much do the two waves of a SIMD overlap in the GRU recurrence?  TAE_GRU_NW=4: 4 waves per workgroup = ONE per SIMD (LDS allows one
# workgroup per CU), 8: two per SIMD (default).  Kernel traces at 16 384 and 8 192 blocks.  -> gpurun_out/r04_gru_nw_probe.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; R=$PWD; mkdir -p gpurun_out; out=$R/gpurun_out/r04_gru_nw_probe.txt; : > $out
cd /tmp; export TMPDIR=/tmp
for B in 16384 8192; do for nw in 8 4; do
  d=/tmp/prof_nw_${B}_$nw
  TAE_GRU_NW=$nw timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/quick_bench_cfg.py 100 $B 2 TurboAE_rate3_rnn > $d.log 2>&1 < /dev/null
  echo "== B=$B TAE_GRU_NW=$nw: $(tail -1 $d.log | cut -c1-90)" | tee -a $out
  python $R/tools/trace_summary.py $d < /dev/null | grep "gru_rec_h\|gru_proj_h" | tee -a $out
done; done
