#!/bin/bash
# Final-tree verification: the WHOLE -m gpu suite, exactly as the driver runs it (-x -q), once in file order
# and once with the test files reversed (order independence), on the tree gpurun shipped.  Tails -> gpurun_out/pytest_final_<tag>.txt
# usage (through gpurun): bash tools/gpu_final.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-cur}
mkdir -p $OUT
cd $R
F=$OUT/pytest_final_$TAG.txt
{ echo "== tree: $(cat .tree_id 2>/dev/null || echo unknown)   date: $(date -u +%FT%TZ)";
  echo "== python -m pytest tests/ -x -q -m gpu   (file order)";
  python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15;
  echo "== python -m pytest -x -q -m gpu <test files reversed>";
  python -m pytest -x -q -m gpu $(ls -r tests/test_gpu_*.py) 2>&1 | grep -v amdgpu.ids | tail -15;
  echo "== smoke"; python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3; } > $F 2>&1
cat $F
