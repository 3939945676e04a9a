#!/bin/bash
# usage: gpurun -- ./tools/gpu_rocprof.sh <tag> <python script (repo-relative)> [args...]   (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}; TAG=$1; SCRIPT=$2; shift; shift; OUT=$R/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/$SCRIPT "$@" > $OUT/out.txt 2> $OUT/err.txt
echo "rc=$?"
for f in $OUT/prof/*/*kernel_stats.csv; do cut -c1-260 $f | head -30; done
