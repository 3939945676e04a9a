#!/bin/bash
# usage: gpurun -- ./tools/gpu_fuzz.sh <tag> <seconds> <seed>
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout $(( $2 + 120 )) python3 tools/fuzz_parity.py $2 $3 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz.txt | tail -5
