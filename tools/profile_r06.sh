#!/bin/bash
# Round-6 profile session (gpurun --timeout 1200 -- bash tools/profile_r06.sh).
#   A1. rocprofv3 --kernel-trace --stats of the driver's bench command with ONLY the headline workload in the process
#       (--no-configs --no-ablate --no-cpu-baseline): the dominant kernel's average is then the headline launches' alone
#       (round-4 review: the all-in-one stats mixed 1e5-point oracle samples and the 128^3 shard into one average)
#   A2. the same of the full default command (every configuration's kernels)
#   B.  per configuration (cfg2, cfg3, cfg5 shard, cfg4): separate --pmc passes FETCH_SIZE | WRITE_SIZE | TCC hit / miss /
#       request counters of tools/bench_configs.py --only <config>, + the known-byte-count stream kernel for the
#       FETCH_SIZE / WRITE_SIZE corrections (MI355X_MICROARCH.md section HBM); cfg2 also with the sweep kernel off
#   (round 6: the automatic path's sampling kernel k_sweep_probe and the gated one-pass launch show up as their own rows in the
#   kernel statistics — a few launches each under the default thinned-out policy)
# Every rocprofv3 call is wrapped in `timeout`; PMC passes use --kernel-trace only; the program follows `--` directly.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
export PROF_TAG=prof_r06
OUT="$R/gpurun_out/$PROF_TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
PY=python3
echo "A1"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- $PY $R/bench.py --steps 20 --warmup 5 --no-configs --no-ablate --no-cpu-baseline --no-live-traffic > $OUT/stats_headline_bench.json 2> $OUT/stats_headline.err || echo "A1 failed"
echo "A2"; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err || echo "A2 failed"
echo "B"; date
for c in FETCH_SIZE WRITE_SIZE; do
  if [ -x $R/tools/tune_linear3d ]; then
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/tools/tune_linear3d 1e8 64 cal > $OUT/cal_$c.log 2>&1
  fi
done
declare -A CFG
CFG[cfg2]="cfg2 3D linear regular 64^3 1e8"
CFG[cfg3]="cfg3 3D linear rectilinear 64^3 1e8"
CFG[cfg5]="cfg5-shard 3D linear regular 128^3 1e8"
CFG[cfg4]="cfg4 4D cubic regular 32^4 1e7 (linearize=false)"
for key in cfg2 cfg3 cfg5 cfg4; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${key}_p$i -- $PY $R/tools/bench_configs.py --only "${CFG[$key]}" > $OUT/${key}_p$i.log 2>&1 || echo "$key pass $i failed/timeout"
  done
  timeout -k 10 200 $PY $R/tools/bench_configs.py --only "${CFG[$key]}" > $OUT/${key}.time 2>&1
done
# cfg2 through the brick kernel (option sweep = 0): the round-4 path, for the same counters side by side
export INTERPN_HIP_SWEEP=0
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/cfg2brick_p$i -- $PY $R/tools/bench_configs.py --only "${CFG[cfg2]}" > $OUT/cfg2brick_p$i.log 2>&1 || echo "cfg2 brick pass $i failed/timeout"
done
timeout -k 10 200 $PY $R/tools/bench_configs.py --only "${CFG[cfg2]}" > $OUT/cfg2brick.time 2>&1
unset INTERPN_HIP_SWEEP
date
$PY $R/tools/profile_r04_summary.py > $OUT/summary.txt 2>&1; tail -40 $OUT/summary.txt
