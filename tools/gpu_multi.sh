#!/bin/bash
# usage: gpurun -- bash tools/gpu_multi.sh <tag> '<cmd 1>' '<cmd 2>' ...   (each under timeout, joined with &&; output per command under gpurun_out/<tag>/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:?tag}; shift; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
k=0
for c in "$@"; do
  k=$((k+1))
  echo "## $c" > $OUT/step$k.txt
  timeout -k 10 600 bash -c "$c" >> $OUT/step$k.txt 2>&1
  rc=$?
  echo "step $k rc=$rc: $c"
  tail -12 $OUT/step$k.txt | cut -c1-400
  if [ $rc -ne 0 ]; then exit $rc; fi
done
