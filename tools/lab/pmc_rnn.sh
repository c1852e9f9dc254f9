#!/bin/bash
# PMC passes (--pmc only, one counter group per pass) over a recurrent-decoder configuration at 16 384 blocks:
#   bash tools/lab/pmc_rnn.sh <gru|lstm|rnn> <tag>   -> gpurun_out/pmc_<tag>.txt (mean counter value per dispatch of every recurrent kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; CELL=${1:-gru}; TAG=${2:-rnn}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/pmc_$TAG.txt
run() {
  name=$1; shift
  d=/tmp/pmc_${TAG}_$name
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $d -o pmc -- python $R/tools/lab/quick_bench_rnn.py $CELL 16384 > $d.log 2>&1 < /dev/null
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $OUT/pmc_$TAG.txt "$CELL" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
gmax = collections.defaultdict(int)
for r in rows: gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r.get("Grid_Size", 0) or 0))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if not any(s in k for s in ("gru_", "rnn_rec_u", "rnn_proj_u", "rnn_l1f_u")) or int(r.get("Grid_Size", 0) or 0) != gmax[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "a") as fh:
    for k in sorted(agg):
        for c in sorted(agg[k]):
            line = f"{k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')}\t{c}\t{agg[k][c]/n[(k,c)]:.6g}\t(dispatches={n[(k,c)]})"
            fh.write(line + "\n"); print(line)
PY
}
echo "# rocprofv3 --pmc passes over \`python tools/lab/quick_bench_rnn.py $CELL 16384\`: mean counter value per full-size dispatch" >> $OUT/pmc_$TAG.txt
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
run hbm_r FETCH_SIZE
run hbm_w WRITE_SIZE
