#!/bin/bash
# usage (on the GPU box): tools/revalidate.sh  -- the launch-policy decisions of DESIGN.md section 4, one at a time against the default, inside the cycle
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-28s %.1f V-cycles/s  %.3f ms ' % ('$1', d['value'], d['ms_per_step']), ' '.join('%s %.3f' % (x['kernel'], x['ms']) for x in r['finest_level_launches']))"; }
run default
# (round 4: the switches whose A/B round 3 settled -- tightest k-tiles, the XCD groupings, 4 + 0 up-leg, r-assembling norm launch --
# are gone from the library; what is left are the options of include/mg3d.h, here through their environment overrides, plus MG3D_LEGS)
for kv in MG3D_LEGS=0 "MG3D_LEGS=0 MG3D_NO_CARRY=1" MG3D_NO_TINY=1 MG3D_NO_TINY_CYCLE=1 MG3D_LU_REDUCED=0 MG3D_SWEEP_TUNE=0 MG3D_SMALL_MAX=0 MG3D_SMALL_MAX=65 MG3D_FUSE_LEG_MAX=65 MG3D_FUSE_UP_MAX=65 MG3D_FUSE_RST2=0; do
  ( export $kv; run $kv )
done
run default
if [ -f multigrid_parallel_amd/lib/libmg3d_nt0.so ]; then ( export MG3D_LIB_PATH=$GRAFT_REPO_ROOT/multigrid_parallel_amd/lib/libmg3d_nt0.so; run "plain stores (MG3D_NT=0)" ); fi
