#!/bin/bash
# Soak: the seeded fuzz tests with other seeds / more cases (TAE_FUZZ_SEED, TAE_FUZZ_CASES) - hunts for shape-dependent bugs.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for seed in ${SEEDS:-101 202 303}; do
  echo "== seed $seed"; TAE_FUZZ_SEED=$seed TAE_FUZZ_CASES=${CASES:-60} python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4
done
