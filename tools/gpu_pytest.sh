#!/bin/bash
# usage: gpurun -- ./tools/gpu_pytest.sh <tag> <pytest args...>
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}; TAG=$1; shift; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd $R
timeout 1500 python3 -m pytest "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest.log
