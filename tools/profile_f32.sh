#!/bin/bash
# usage (on the GPU box): tools/profile_f32.sh <tag>  -- the evidence of the fp32 / damped-Jacobi / F-cycle variant (BASELINE configs[4], parity
# unpinned) at 1025^3 on one GPU: 1. the bench line with its per-launch roofline  2. rocprofv3 --kernel-trace --stats of the same command
# 3. HBM-side bytes per launch from two PMC passes (FETCH_SIZE; WRITE_SIZE -- not in one pass; read bytes = 2 x FETCH_SIZE on gfx950).
# Everything lands in gpurun_out/prof_<tag>_f32/ ; copy what is to be judged into profiles/.
tag=${1:-r04}
[ -n "$GRAFT_REPO_ROOT" ] && [ -d "$GRAFT_REPO_ROOT" ] || { echo "profile_f32.sh: GRAFT_REPO_ROOT is not set (run this through gpurun)"; exit 2; }
out=gpurun_out/prof_${tag}_f32
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
mkdir -p $out
python3 bench.py --f32 --steps 10 --warmup 2 > $out/${tag}_f32_bench_line.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 bench.py --f32 --steps 10 --warmup 2 > $out/${tag}_f32_bench_line_profiled.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_f32_jacobi_1025_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 bench.py --f32 --steps 4 --warmup 1 > /dev/null 2> $out/fetch.err || { tail -5 $out/fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -o p -- python3 bench.py --f32 --steps 4 --warmup 1 > /dev/null 2> $out/write.err || { tail -5 $out/write.err; exit 1; }
python3 tools/pmc_traffic_table.py $out/fetch $out/write > $out/${tag}_f32_pmc_traffic.txt 2>&1
rm -rf $out/stats $out/fetch $out/write
ls -la $out; cat $out/${tag}_f32_pmc_traffic.txt | head -30
