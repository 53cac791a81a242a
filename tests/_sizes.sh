mkdir -p gpurun_out/r4n
for cl in "7 7" "11 7" "13 7" "9 8" "9 8"; do set -- $cl; python bench.py --coarse $1 --levels $2 --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c L = $1 $2:', d['config']['workload'][:8], 'carried %.2f ms  plain %.2f ms  legs %.2f ms' % (d['ms_per_step'], d['plain_schedule']['ms_per_step'] if d['plain_schedule'] else 0, d['legs_schedule']['ms_per_step']))"; done > gpurun_out/r4n/sizes.txt 2>&1
cat gpurun_out/r4n/sizes.txt
