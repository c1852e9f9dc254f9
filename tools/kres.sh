#!/bin/bash
# kernel resource usage of one .hip source (VGPRs / SGPRs / scratch / occupancy per kernel), compiled for gfx950:
#   tools/kres.sh turboae_amd/csrc/turboae_h2.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-function \
  -c "$src" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /SGPRs Spill:/ {ss=$(NF-1)} /VGPRs Spill:/ {vs=$(NF-1)}
       /LDS Size/ {printf "%s vgpr %s sgpr %s scratch %s occ %s sgpr_spill %s vgpr_spill %s\n", n, v, s, sc, o, ss, vs}' | c++filt
rm -f /tmp/kres_$$.o
