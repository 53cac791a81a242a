#!/bin/bash
# usage (on the GPU box): tools/profile_all.sh  -- everything profiles/ holds for round 4 in one go (copy the r04_* files from gpurun_out/)
tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -3 gpurun_out/prof_r04.log
tools/profile_legs.sh r04 > gpurun_out/prof_r04_legs.log 2>&1; tail -3 gpurun_out/prof_r04_legs.log
tools/profile_f32.sh r04 > gpurun_out/prof_r04_f32.log 2>&1; tail -3 gpurun_out/prof_r04_f32.log
python tools/dist_loopback_bench.py 2 4 8 > gpurun_out/r04_loopback.txt 2>&1; cat gpurun_out/r04_loopback.txt
python tools/pcie_inclusive.py > gpurun_out/r04_pcie.txt 2>&1; tail -4 gpurun_out/r04_pcie.txt
for L in 5 6 8; do python bench.py --levels $L --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('levels', $L, d['value'], d['ms_per_step'], d['schedule'], d['plain_schedule'] and d['plain_schedule']['value'], d['legs_schedule'] and d['legs_schedule']['value'])"; done > gpurun_out/r04_other_sizes.txt 2>&1; cat gpurun_out/r04_other_sizes.txt
