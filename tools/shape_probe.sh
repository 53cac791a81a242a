#!/bin/bash
# Resources (VGPR / AGPR / scratch / LDS) of ONE shape of the sweep kernel without compiling every dispatch:
#   tools/shape_probe.sh "4, 0, 4, 8, 1, true, true, 2, 4" [extra hipcc flags, e.g. -DMG3D_EDGE_UNCOND=0]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
args="$1"; shift
cat > $T/p.hip <<EOT
#include "mg3d_sweep_kernel.h"
template __global__ void sweep_kernel<$args>(SweepArgs);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$ROOT/multigrid_parallel_amd/csrc "$@" \
   -c $T/p.hip -o $T/p.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs|ScratchSize|LDS Size|Occupancy|SGPRs:|error" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '
echo
if [ -n "$KEEP_ASM" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$ROOT/multigrid_parallel_amd/csrc "$@" -S --cuda-device-only -o $KEEP_ASM $T/p.hip; fi
rm -rf $T
