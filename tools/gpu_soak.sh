#!/bin/bash
# Soak: the seeded fuzz tests with other seeds / more cases (TAE_FUZZ_SEED, TAE_FUZZ_CASES) - hunts for shape-dependent bugs.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for seed in ${SEEDS:-101 202 303}; do
  echo "== seed $seed"; TAE_FUZZ_SEED=$seed TAE_FUZZ_CASES=${CASES:-60} python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4
  TAE_FUZZ_SEED_GENERIC=$seed TAE_FUZZ_CASES_GENERIC=${GCASES:-42} python -m pytest tests/test_gpu_generic.py -m gpu -q -k random 2>&1 | tail -3
done
