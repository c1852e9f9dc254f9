#!/bin/bash
# A short GPU-box visit: selected tests + one bench line (summary printed).  usage: bash tools/gpu_quick.sh <tag> "<pytest -k expr|none>" [bench flags]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-cur}; SEL=${2:-none}; shift 2 || true
mkdir -p $OUT; cd $R
if [ "$SEL" != "none" ]; then python -m pytest tests -m gpu -x -q -k "$SEL" 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/pytest_$TAG.log; fi
python bench.py --steps 10 --warmup 2 "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
grep -v amdgpu.ids $OUT/bench_$TAG.err | tail -5
python - $OUT/bench_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"bench: {d['value']/1e6:.2f} Mbit/s  ms/step {d['ms_per_step']:.2f}  dec {r['kernel_ms']:.2f} ms frac {r['frac']:.4f}  probe {r.get('sustained_probe_tflops')} frac_of_sustained {r.get('frac_of_sustained')}  ber {d['ber']:.5f}")
if "roofline_f32" in d: print("f32:", d["roofline_f32"]["value_bits_per_s"] / 1e6, d["roofline_f32"]["frac"])
for o in [x for x in r.get("other_configs", []) if "error" not in x]:
    print(f"  {o['config']}: {o['ms_per_forward']:.2f} ms  {o['bits_per_s']/1e6:.2f} Mbit/s  dec {o['decoder_ms']:.2f} ms frac {o['decoder_frac']:.3f}  enc {o['encoder_plus_norm_ms']:.2f} ms frac {o['encoder_frac']:.3f}  ber {o['ber']:.5f} [{o['weights']}]")
print("ranks:", d["config"].get("rccl_ranks_seen"), d["config"].get("per_rank_ms_per_step"))
print("parity:", d.get("parity")); c = d.get("cpu_baseline", {}); print("cpu:", c.get("value"), c.get("cores"))
PY
