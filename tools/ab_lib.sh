#!/bin/bash
# A/B of two builds of the library on one box: the in-tree one vs $1 (a variant .so), alternating, for the CNN configurations.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
V=$1
for rep in 1 2; do
for lib in "" "$V"; do
  echo "== lib: ${lib:-in-tree}"
  for c in "100 50000 2" "1000 25000 2" "100 500 2" "100 100000 5"; do ${lib:+env TAE_LIB=$lib} python tools/quick_bench_cfg.py $c 2>&1 | grep -v amdgpu | cut -c1-100; done
done; done
