#!/bin/bash
# A/B of library builds on ONE box with the contract bench's own workload (trained weights, 50 000 blocks): decoder kernel time,
# step time and the sustained-MFMA probe of each, alternating.  usage: bash tools/lab/ab_bench.sh <lib1.so|in-tree> <lib2.so> ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
for lib in "$@"; do
  if [ "$lib" = "in-tree" ]; then unset TAE_LIB; else export TAE_LIB=$PWD/$lib; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-sweep --no-f32-pass --no-parity --no-graph --no-pmc ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-45s dec %.2f ms  step %.2f ms  frac %.4f  probe %.0f  of_sustained %.3f' % ('$lib', r['kernel_ms'], d['ms_per_step'], r['frac'], r.get('sustained_probe_tflops',0), r.get('frac_of_sustained',0)))"
done; done
