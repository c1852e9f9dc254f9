#!/bin/bash
# Issue-slot breakdown of the GRU recurrence kernels (16 384 blocks): where the wave cycles of gru_rec_h go.  -> gpurun_out/r04_pmc_gru_rec.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r04_pmc_gru_rec.txt; : > $out; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep "ACTIVE_INST\|INSTS_VALU\|INSTS_MFMA\|WAIT_INST\|INST_CYCLES\|VALU_MFMA\|BUSY_CY\|INSTS_LDS\|INSTS_SALU\|INSTS_VMEM\|INSTS_SMEM" | tr '\n' ' ' | tee -a $out; echo | tee -a $out
pass() {  # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_grurec_$n -o pmc -- python $R/tools/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > /tmp/pmc_grurec_$n.log 2>&1 < /dev/null
  python - $n <<'PY' | tee -a $out
import csv, glob, os, collections, sys
fs = glob.glob("/tmp/pmc_grurec_" + sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs:
    print("pass", sys.argv[1], "failed:", open("/tmp/pmc_grurec_" + sys.argv[1] + ".log").read()[-400:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "gru_rec" in k or "gru_proj" in k:
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pass b SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES
pass c SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
