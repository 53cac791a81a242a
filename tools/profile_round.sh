#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>   -- everything lands in gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/
# 1. PMC traffic passes (tools/pmc_traffic.py)  2. bench.py line  3. rocprofv3 --kernel-trace --stats of the same command  4. SQ counters
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
[ -n "$GRAFT_REPO_ROOT" ] && [ -d "$GRAFT_REPO_ROOT" ] || { echo "profile_round.sh: GRAFT_REPO_ROOT is not set (run this through gpurun)"; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
# the counter passes first: bench.py reads profiles/pmc_traffic.json and drops it when the kernel sources changed since
python3 tools/pmc_traffic.py $tag > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic_finest_level.txt $out/ 2>/dev/null
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_line_profiled.json 2> $out/stats.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
python3 tools/finest_from_trace.py $(find $out/stats -name "*kernel_trace.csv" | head -1) > $out/${tag}_bench_kernel_stats_note.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/sq -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/sq.err
python3 tools/sq_table.py $out/sq > $out/${tag}_pmc_sq_finest_level.txt
rm -rf $out/stats $out/sq gpurun_out/pmc_fetch gpurun_out/pmc_write
ls -la $out
