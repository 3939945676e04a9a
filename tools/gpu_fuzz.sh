#!/bin/bash
# usage: gpurun -- ./tools/gpu_fuzz.sh <tag> <seconds> <seed>
# Runs the differential fuzzer in chunks of at most 240 s (a run that prints nothing for 7 minutes
# is taken to be hung by the GPU box), one seed per chunk, one summary line per chunk.
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd $R
LEFT=$2; SEED=$3
while [ $LEFT -gt 0 ]; do
  T=$(( LEFT < 240 ? LEFT : 240 ))
  timeout $(( T + 120 )) python3 tools/fuzz_parity.py $T $SEED 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT/fuzz.txt || exit 1
  LEFT=$(( LEFT - T )); SEED=$(( SEED + 1 ))
done
