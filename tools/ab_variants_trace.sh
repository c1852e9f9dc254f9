#!/bin/bash
# Per-kernel average times (rocprofv3 --kernel-trace --stats) of the GRU-decoder configuration for several library builds on one box:
#   tools/ab_variants_trace.sh in-tree tools/probes/libs/libturboae_x.so ...        (filter with KFILTER, default "gru_")
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  d=/tmp/abtrace_$$_$(basename $lib .so)
  if [ "$lib" = "in-tree" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/quick_bench_cfg.py 100 ${AB_B:-16384} 2 TurboAE_rate3_rnn > $d.log 2>&1
  else TAE_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $R/tools/quick_bench_cfg.py 100 ${AB_B:-16384} 2 TurboAE_rate3_rnn > $d.log 2>&1; fi
  echo "== $lib: $(grep forward $d.log | cut -c1-90)"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "${KFILTER:-gru_}" <<'PY2'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]: print(f"   {r['Name'][:72]:72s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY2
done
