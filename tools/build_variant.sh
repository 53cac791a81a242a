#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  ->  multigrid_parallel_amd/lib/libmg3d_<name>.so
# an alternative build of the product library for A/B timing on one box (MG3D_LIB_PATH selects it)
set -e
name=$1; shift
cd "$(dirname "$0")/../multigrid_parallel_amd/csrc"
mkdir -p build_$name
for f in mg3d_kernels mg3d_sweep mg3d_ctx mg3d_dist mg3d_f32 mg3d_f32_dist mg3d_es mg3d_tiny; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include -I. "$@" -c $f.hip -o build_$name/$f.o &
done
gcc -O2 -fPIC -ffp-contract=off -std=gnu99 -Wall -I../../include -I. -c mg3d_host.c -o build_$name/mg3d_host.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmg3d_$name.so build_$name/*.o -L/opt/rocm/lib -lrccl -lm
echo built ../lib/libmg3d_$name.so
