#!/bin/bash
# Register / spill / scratch / LDS use of every kernel of npb_kernels.hip as the Makefile compiles it (fp64 build; add
# -DNPB_BUILD_F32 as $1 for the fp32-storage build).  No GPU needed.
cd "$(dirname "$0")/../nuclear_sim_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -freciprocal-math -fapprox-func \
  -fvisibility=hidden $1 -Rpass-analysis=kernel-resource-usage -c -o /dev/null npb_kernels.hip 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name); gsub(/^_Z[0-9]+/,"",name); sub(/12npb_params_t.*|mPd$|iimPKdPd$|PKdm.*|Pdm.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} / AGPRs:/ {a=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {oc=$(NF-1)} /LDS Size/ {printf "%-28s VGPR %3s AGPR %3s  SGPR-spill %4s VGPR-spill %3s scratch %5s B/lane  occupancy %s  LDS %s B\n", name, v, a, ss, vs, sc, oc, $(NF-1)}'
