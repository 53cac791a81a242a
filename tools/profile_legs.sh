#!/bin/bash
# usage (on the GPU box): tools/profile_legs.sh <tag>  -- the one-launch-per-leg schedule (option legs = 1: the default from 450 points
# per side since round 4) beside the carried cycles (MG3D_LEGS=0): bench lines of both, kernel trace + stats of the legs run, one
# steady-state cycle of each as a timeline.
tag=${1:-r04}
[ -n "$GRAFT_REPO_ROOT" ] && [ -d "$GRAFT_REPO_ROOT" ] || { echo "profile_legs.sh: GRAFT_REPO_ROOT is not set (run this through gpurun)"; exit 2; }
out=gpurun_out/prof_${tag}_legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
mkdir -p $out
MG3D_LEGS=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_legs_bench_line.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
MG3D_LEGS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --timing-mode 0 --no-alt-schedules > /dev/null 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_legs_kernel_stats.csv
python3 tools/cycle_timeline.py $(find $out/stats -name "*kernel_trace.csv" | head -1) 2 > $out/${tag}_legs_cycle_timeline.txt 2>&1
MG3D_LEGS=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_carried_bench_line.json 2>> $out/bench.err
MG3D_LEGS=0 rocprofv3 --kernel-trace --output-format csv -d $out/stats2 -o p -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --timing-mode 0 --no-alt-schedules > /dev/null 2> $out/stats2.err
python3 tools/cycle_timeline.py $(find $out/stats2 -name "*kernel_trace.csv" | head -1) 2 > $out/${tag}_carried_cycle_timeline.txt 2>&1
rm -rf $out/stats $out/stats2
ls -la $out
