#!/bin/bash
# usage (on the GPU box): tools/leg_max_sweep.sh  -- MG3D_FUSE_LEG_MAX (whole legs of the small levels as one launch each) by threshold, bench line at 513^3 and 129^3, twice round-robin
for r in 1 2; do
for v in 0 33 65; do
  for lv in 7 5; do
    MG3D_FUSE_LEG_MAX=$v python bench.py --levels $lv --steps 40 --warmup 3 --no-cpu-baseline --timing-mode 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('leg_max $v levels $lv: %.1f V-cycles/s  %.4f ms' % (d['value'], d['ms_per_step']))"
  done
done
done
