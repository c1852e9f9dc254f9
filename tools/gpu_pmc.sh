#!/bin/bash
# PMC passes (separate from tracing, as the guide prescribes) over a short bench run.
# usage: bash tools/gpu_pmc.sh [tag]   -> gpurun_out/pmc_<tag>_<pass>/..., summary printed
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-cur}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B=${PMC_BATCH:-12288}
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline > $OUT/pmc_${TAG}_$name.log 2>&1
  f=$(find $OUT/pmc_${TAG}_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k in agg:
    if "dec_kernel" in k:
        print(k, {c: v / n[(k, c)] for c, v in agg[k].items()})
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES
