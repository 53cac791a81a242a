#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters>" -- python3 prog args...   (run on the GPU box)
# collects one rocprofv3 --pmc pass (kernel-trace only, csv) into gpurun_out/<outdir>
out=$1; ctr=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/$out -o p -- "$@" > gpurun_out/$out.log 2>&1
