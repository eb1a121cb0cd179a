#!/bin/bash
# Kernel-time profile of one bench command on the GPU box: tools/prof_stats.sh NAME <bench args...>
#   -> gpurun_out/prof/NAME_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
set -e -o pipefail
NAME=$1; shift
OUT=gpurun_out/prof/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-legs "$@" > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/prof/${NAME}_kernel_stats.csv
find $OUT -name '*_kernel_trace.csv' -delete || true
find $OUT -name '*_agent_info.csv' -delete || true
