# PMC passes over the GRU-decoder configuration (16 384 blocks): matrix-pipe / wait / LDS counters, then HBM-side bytes per kernel.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
pass() {  # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_gru_$n -o pmc -- python $R/tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $R/gpurun_out/pmc_gru_$n.log 2>&1
  python - $n <<'PY'
import csv, glob, os, collections, sys
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gru_" + sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "gru" in k:
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
}
pass sq SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass hbm_r FETCH_SIZE
pass hbm_w WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
