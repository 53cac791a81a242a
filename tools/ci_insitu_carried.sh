#!/bin/bash
# usage (on the GPU box): tools/ci_insitu_carried.sh -- chunk length of the two carried-cycle launches varied inside the cycle (bench line + per-launch ms)
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', '%.1f  ms %.3f ' % (d['value'], d['ms_per_step']), ' '.join('%s %.3f' % (x['kernel'], x['ms']) for x in r['finest_level_launches']))"; }
MG3D_SWEEP_TUNE_LOG=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "mg3d sweep" | grep "513x" 
for ci in 0 129 172 257 103 0; do export MG3D_SWEEP_CI_43=$ci; run "TAP_CI=$ci"; done; unset MG3D_SWEEP_CI_43
for ci in 0 129 172 257 103; do export MG3D_SWEEP_CI_12=$ci; run "S1RST_CI=$ci"; done; unset MG3D_SWEEP_CI_12
