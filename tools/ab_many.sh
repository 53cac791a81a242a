#!/bin/bash
# usage (on the GPU box): tools/ab_many.sh <variant|-> ...   -- per-level launch times and the bench line for several library builds, twice round-robin
libdir=$GRAFT_REPO_ROOT/multigrid_parallel_amd/lib
for r in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset MG3D_LIB_PATH; else export MG3D_LIB_PATH=$libdir/libmg3d_$v.so; fi
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'V-cycles/s %.1f  ms %.3f ' % (d['value'], d['ms_per_step']), ' '.join('%s %.3f' % (x['kernel'], x['ms']) for x in r['finest_level_launches']))"
    if [ $r = 1 ]; then REPS=100 python tools/level_bench.py 2>/dev/null | sed "s/^/   $v /"; fi
  done
done
