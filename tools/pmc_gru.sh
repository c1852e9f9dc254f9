cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_gru -o pmc -- python $R/tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $R/gpurun_out/pmc_gru.log 2>&1
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gru/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "gru" in k:
        print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
