# LDS conflict attribution (profiles/r02_pmc_f16x2/lds_attribution.txt).  Build the timing variant first (here, no GPU needed):
#   make -C turboae_amd/csrc OUT=../lib/variants/libtae_x16.so OBJD=../lib/obj_x16 EXTRA="-DTAE_EXPERIMENT -DTAE_X=16"
# then on the GPU box:  gpurun -- 'bash tools/lab/pmc_lds_attribution.sh'
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh r02e 2>&1 | grep -E "dec_kernel_h" 
echo "=== x16 (no LDS operand reads in the K loop; results wrong, counters only)"
export TAE_LIB=$GRAFT_REPO_ROOT/turboae_amd/lib/variants/libtae_x16.so
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_x16 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-pass --no-parity --random-weights > $GRAFT_REPO_ROOT/gpurun_out/pmc_x16.log 2>&1
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_x16/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "dec_kernel_h" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("x16 dec_kernel_h", k, "%.4g" % (sum(v) / len(v)), "(n=%d)" % len(v))
PY
