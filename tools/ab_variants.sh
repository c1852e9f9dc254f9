#!/bin/bash
# A/B of library builds on ONE GPU box: the in-tree library vs turboae_amd/lib/variants/<name>.so (TAE_LIB), alternating.
# usage: bash tools/ab_variants.sh <variant.so> [B] [rounds]
cd ${GRAFT_REPO_ROOT:-.}
V=$PWD/turboae_amd/lib/variants/$1; N=${2:-50000}; R=${3:-3}
for i in $(seq $R); do
  echo "tree   : $(python tools/quick_bench.py $N 2>&1 | grep -v amdgpu | sed 's/TFLOP.*only)//' | tr '\n' ' ')"
  echo "variant: $(TAE_LIB=$V python tools/quick_bench.py $N 2>&1 | grep -v amdgpu | sed 's/TFLOP.*only)//' | tr '\n' ' ')"
done
