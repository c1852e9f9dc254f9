#!/bin/bash
# One GPU-box visit of the build loop: [tests] + other-config timings + bench + rocprofv3 kernel trace (+ PMC passes with PMC=1).
# usage (through gpurun): bash tools/gpu_round.sh <tag> [pytest -k expression | all | none]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-cur}
SEL=${2:-all}
mkdir -p $OUT
cd $R
if [ "$SEL" = "all" ]; then python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_$TAG.log
elif [ "$SEL" != "none" ]; then python -m pytest tests -m gpu -q -k "$SEL" 2>&1 | tail -40 > $OUT/pytest_$TAG.log; fi
[ -f $OUT/pytest_$TAG.log ] && tail -8 $OUT/pytest_$TAG.log
# (the other BASELINE configs are timed inside bench.py since r03: roofline.other_configs)
python bench.py --steps 10 --warmup 2 ${BENCH_FLAGS:-} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
python - $OUT/bench_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r, c, p = d["roofline"], d.get("cpu_baseline", {}), d.get("parity", {})
print(f"bench: {d['value']/1e6:.2f} Mbit/s  ms/step {d['ms_per_step']:.2f} (median {d['ms_per_step_median']:.2f})  dec {r['kernel_ms']:.2f} ms frac {r['frac']:.4f}"
      + (f"  f32 frac {d['roofline_f32']['frac']:.4f}" if 'roofline_f32' in d else "") + f"  ber {d['ber']:.5f}")
for o in [x for x in r.get("other_configs", []) if "error" not in x]:
    print(f"  {o['config']}: {o['ms_per_forward']:.2f} ms  {o['bits_per_s']/1e6:.2f} Mbit/s  dec {o['decoder_ms']:.2f} ms frac {o['decoder_frac']:.3f}  enc {o['encoder_plus_norm_ms']:.2f} ms frac {o['encoder_frac']:.3f}")
print("probe:", r.get("sustained_probe_tflops"), "frac_of_sustained", r.get("frac_of_sustained"))
print("cpu:", {k: c.get(k) for k in ("value", "cores", "run_to_run_spread", "min_max_spread", "value_B2000", "value_1_thread", "thread_sweep_bits_per_s")})
print("parity:", p)
PY
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs --no-probe --no-graph --no-sweep --no-pmc --no-f16x1 > $OUT/prof_$TAG.log 2>&1
python $R/tools/trace_summary.py $OUT/prof_$TAG $OUT/prof_${TAG}_by_grid.txt
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f"
if [ "${PROF_CFG:-0}" = "1" ]; then     # kernel traces of the long-block (cfg 4 shape) and GRU (cfg 5) configurations
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg4 -o cfg4 -- python $R/tools/lab/quick_bench_cfg.py 1000 25000 2 > $OUT/prof_${TAG}_cfg4.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg5 -o cfg5 -- python $R/tools/lab/quick_bench_cfg.py 100 16384 2 TurboAE_rate3_rnn > $OUT/prof_${TAG}_cfg5.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg2 -o cfg2 -- python $R/tools/lab/quick_bench_cfg.py 100 100000 5 > $OUT/prof_${TAG}_cfg2.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg0 -o cfg0 -- python $R/tools/lab/quick_bench_cfg.py 100 500 2 > $OUT/prof_${TAG}_cfg0.log 2>&1
  for c in cfg4 cfg5 cfg2 cfg0; do f=$(find $OUT/prof_${TAG}_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "== $c"; head -6 "$f"; }; python $R/tools/trace_summary.py $OUT/prof_${TAG}_$c $OUT/prof_${TAG}_${c}_by_grid.txt > /dev/null; done
fi
if [ "${PMC:-0}" = "1" ]; then cd $R; bash tools/gpu_pmc.sh $TAG; fi
