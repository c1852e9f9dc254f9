#!/bin/bash
# Does the layer-1 input projection run faster when its GI output fits the 256 MB Infinity Cache?  Per-kernel times of the GRU decoder
# at batches whose GI (B x 100 positions x 2432 B) is 125 / 249 / 498 MB against the production chunk's 4 GB.
cd ${GRAFT_REPO_ROOT:-$(pwd)}; R=$PWD; export TMPDIR=/tmp; cd /tmp
for B in 512 1024 2048 4096 16384; do
  rm -rf /tmp/gmp_$B
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gmp_$B -o t -- python $R/tools/quick_bench_cfg.py 100 $B 2 TurboAE_rate3_rnn > /dev/null 2>&1
  python - $B <<'PY'
import csv, glob, sys, collections
B = int(sys.argv[1])
f = glob.glob(f"/tmp/gmp_{B}/**/*kernel_trace.csv", recursive=True)[0]
t = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gru_" in n: t[n.split("(")[0].replace("void tae::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for k, v in sorted(t.items()):
    v = sorted(v)[len(v) // 4: -len(v) // 4 or None]
    us = sum(v) / len(v)
    out.append(f"{k[:28]} {us:8.1f} us = {us * 1e3 / (B * 100):6.3f} ns/pos")
print(f"B={B:6d} GI={B * 100 * 2432 / 1e6:7.0f} MB | " + " | ".join(out))
PY
done
