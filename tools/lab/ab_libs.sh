#!/bin/bash
# A/B/C of several library builds on one box, alternating: tools/lab/ab_libs.sh <lib1.so|in-tree> <lib2.so> ...   (decoder-heavy configs only)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
for lib in "$@"; do
  echo "== lib: $lib"
  for c in "100 50000 2" "100 100000 5"; do
    if [ "$lib" = "in-tree" ]; then python tools/lab/quick_bench_cfg.py $c 2>&1 | grep -v amdgpu | cut -c1-100; else TAE_LIB=$lib python tools/lab/quick_bench_cfg.py $c 2>&1 | grep -v amdgpu | cut -c1-100; fi
  done
done; done
