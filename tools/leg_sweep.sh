#!/bin/bash
# usage (on the GPU box): tools/leg_sweep.sh  -- small-level policy: two-rows-per-thread shapes up to MG3D_SMALL_MAX points per
# side, whole legs as one launch up to MG3D_FUSE_LEG_MAX
for lv in 7 5; do
  for cfg in "0 0" "65 0" "0 65" "65 65" "129 65" "65 129" "33 33" "0 0" "65 65"; do
    set -- $cfg
    MG3D_SMALL_MAX=$1 MG3D_FUSE_LEG_MAX=$2 python bench.py --levels $lv --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('levels $lv  small shapes up to $1, fused legs up to $2: %.1f V-cycles/s  %.4f ms' % (d['value'], d['ms_per_step']))"
  done
done
