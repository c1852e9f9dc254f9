#!/bin/bash
# Build a variant of the library with extra compile flags for ONE source file (timing / A-B experiments; results may be wrong):
#   tools/build_variant.sh <name> <source.hip> <flags...>   ->  tools/lab/probes/libs/libturboae_<name>.so   (git-ignored, travels with gpurun)
# Run with TAE_LIB=tools/lab/probes/libs/libturboae_<name>.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
mkdir -p $R/tools/lab/probes/libs /tmp/tae_var_$name
make -C $R/turboae_amd/csrc -s
obj=/tmp/tae_var_$name/$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-function -I$R/turboae_amd/csrc "$@" -c $R/turboae_amd/csrc/$src -o $obj
objs=""
for o in $R/turboae_amd/lib/obj/*.o; do [ "$(basename $o)" = "$(basename $obj)" ] && objs="$objs $obj" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/tools/lab/probes/libs/libturboae_$name.so
echo "built tools/lab/probes/libs/libturboae_$name.so"
