#!/bin/bash
# rocprofv3 kernel-trace of tools/column_probe.py (per-kernel durations of every variant)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT=$R/gpurun_out/prof_column
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/column_probe.py "$@" > $OUT/probe.log 2> $OUT/probe.err
for f in $OUT/stats/*/*kernel_stats.csv; do cut -c1-200 $f | head -20; done
grep variant $OUT/probe.log
