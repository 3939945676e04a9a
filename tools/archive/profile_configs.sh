#!/bin/bash
# rocprofv3 kernel-trace summary of every BASELINE single-GPU configuration (tools/bench_configs.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_configs
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/bench_configs.py > $OUT/run.log 2>&1
grep config $OUT/run.log
for f in $OUT/stats/*/*kernel_stats.csv; do grep -E "interpn::|^\"Name\"" $f | cut -c1-220; done
