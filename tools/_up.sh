for v in 0 65 129 257 0 257; do
  export MG3D_FUSE_UP_MAX=$v
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-alt-schedules 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_UP_MAX=$v', 'V-cycles/s %.1f  ms %.3f' % (d['value'], d['ms_per_step']))"
done
