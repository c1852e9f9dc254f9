#!/bin/bash
# same-box A/B of library builds of any ABI version (tools/lab/ab_abi.py), alternating:  bash tools/lab/ab_abi.sh <lib1.so> <lib2.so> ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for lib in "$@"; do python tools/lab/ab_abi.py $lib 2>&1 | grep -v amdgpu; done; done
