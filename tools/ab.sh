#!/bin/bash
# usage: tools/ab.sh <variantA|-> <variantB> [reps]   (on the GPU box): alternates bench.py --breakdown between two library builds
A=$1; B=$2; R=${3:-2}
libdir=$GRAFT_REPO_ROOT/multigrid_parallel_amd/lib
for r in $(seq $R); do
  for v in $A $B; do
    if [ "$v" = "-" ]; then unset MG3D_LIB_PATH; else export MG3D_LIB_PATH=$libdir/libmg3d_$v.so; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown 2> gpurun_out/ab_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'V-cycles/s %.1f  ms %.3f' % (d['value'], d['ms_per_step']))"
    grep "kernel level 6" gpurun_out/ab_$v.err | awk '{printf "    %-16s %s\n", $4, $7}'
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'untimed-kernels run: V-cycles/s %.1f  ms %.3f' % (d['value'], d['ms_per_step']))"
  done
done
