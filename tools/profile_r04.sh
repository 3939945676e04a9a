#!/bin/bash
# Round-4 profile session (gpurun --timeout 1200 -- bash tools/profile_r04.sh).
#   A. rocprofv3 --kernel-trace --stats of the driver's bench command (every configuration's kernels)
#   B. per configuration (cfg2, cfg3, cfg5 shard, cfg4): separate --pmc passes FETCH_SIZE | WRITE_SIZE |
#      TCC hit / miss / request counters of tools/bench_configs.py --only <config>, + the known-byte-count
#      stream kernel for the FETCH_SIZE / WRITE_SIZE corrections (MI355X_MICROARCH.md section HBM)
#   C. cfg4: SQ / LDS counter passes of the sort kernels and k_cubic_column, and of the tiled kernel in place
# Every rocprofv3 call is wrapped in `timeout`; PMC passes use --kernel-trace only; the program follows `--` directly.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT=$R/gpurun_out/prof_r04
rm -rf "$OUT" && mkdir -p "$OUT"
PY=python3
echo "A" ; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err
echo "B" ; date
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
echo "C" ; date
for mode in 1 0; do
  export INTERPN_HIP_BINNED=$mode
  [ $mode = 1 ] && unset INTERPN_HIP_BINNED
  i=0
  while read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/c4_b${mode}_p$i -- $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}_p$i.log 2>&1 || echo "cfg4 binned=$mode pass $i failed/timeout"
  done <<'CNT'
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY
CNT
  timeout -k 10 200 $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}.time 2>&1
done
unset INTERPN_HIP_BINNED
date
$PY $R/tools/profile_r04_summary.py
