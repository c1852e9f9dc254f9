#!/bin/bash
# A/B on ONE GPU box (box-to-box clocks differ by a few percent): alternate two environment settings of the same library,
# decoder-only + forward timing at B blocks.   usage: bash tools/lab/ab_env.sh "<env A>" "<env B>" [B] [rounds]
cd ${GRAFT_REPO_ROOT:-.}
A=$1; Bv=$2; N=${3:-50000}; R=${4:-3}
for i in $(seq $R); do
  echo "A[$A]: $(env $A python tools/lab/quick_bench.py $N 2>&1 | grep -v amdgpu | tr '\n' ' ')"
  echo "B[$Bv]: $(env $Bv python tools/lab/quick_bench.py $N 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
