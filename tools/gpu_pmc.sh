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
B=${PMC_BATCH:-50000}
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-f32-pass --no-parity --no-other-configs --no-probe --no-graph --no-sweep --no-pmc --no-f16x1 > $OUT/pmc_${TAG}_$name.log 2>&1
  f=$(find $OUT/pmc_${TAG}_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $OUT/pmc_${TAG}_$name.txt $B <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# full-size dispatches only: since r04 a handle's range calibration launches the same kernels on a small batch at creation
gmax = collections.defaultdict(int)
for r in rows: gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r.get("Grid_Size", 0) or 0))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if int(r.get("Grid_Size", 0) or 0) != gmax[k]: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as fh:
    fh.write(f"# rocprofv3 --pmc pass over `python bench.py --steps 2 --warmup 1 --batch {sys.argv[3]} --no-cpu-baseline --no-f32-pass --no-parity --no-other-configs --no-probe --no-graph --no-sweep --no-pmc --no-f16x1`; mean counter value per FULL-SIZE dispatch (largest grid of each kernel)\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            line = f"{k}\t{c}\t{agg[k][c]/n[(k,c)]:.6g}\t(dispatches={n[(k,c)]})"
            fh.write(line + "\n")
            if "dec_kernel" in k or "enc_kernel" in k: print(line)
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run hbm_r FETCH_SIZE
run hbm_w WRITE_SIZE
