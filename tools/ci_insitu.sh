run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', '%.1f  ms %.3f ' % (d['value'], d['ms_per_step']), ' '.join('%s %.3f' % (x['kernel'], x['ms']) for x in r['finest_level_launches']))"; }
for ci in 52 57 65 74 86 103 129 0 65 103; do export MG3D_SWEEP_CI_02=$ci; run "B_CI=$ci"; done; unset MG3D_SWEEP_CI_02
for ci in 74 86 103 129; do export MG3D_SWEEP_CI_21=$ci; run "D_CI=$ci"; done; unset MG3D_SWEEP_CI_21
for ci in 74 86 103 129; do export MG3D_SWEEP_CI_20P=$ci; run "C_CI=$ci"; done; unset MG3D_SWEEP_CI_20P
