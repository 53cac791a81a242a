for r in 1 2; do
for v in 0 33 65 129 257; do
  for lv in 7 6; do
    MG3D_FUSE_UP_MAX=$v python bench.py --levels $lv --steps 40 --warmup 3 --no-cpu-baseline --timing-mode 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('up_max $v levels $lv: %.1f V-cycles/s  %.4f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
