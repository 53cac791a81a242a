#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters>" -- python3 prog args...   (run on the GPU box)
# collects one rocprofv3 --pmc pass (kernel-trace only, csv) into gpurun_out/<outdir>
out=$1; ctr=$2; shift 3
[ -n "$GRAFT_REPO_ROOT" ] && [ -d "$GRAFT_REPO_ROOT" ] || { echo "pmc.sh: GRAFT_REPO_ROOT is not set (run this through gpurun)"; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/$out -o p -- "$@" > gpurun_out/$out.log 2>&1
