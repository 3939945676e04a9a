#!/bin/bash
# Generic GPU-box runner: gpurun -- ./tools/gpu_run.sh <tag> <command...>; output -> gpurun_out/<tag>.log
TAG=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
"$@" > gpurun_out/$TAG.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG.log
grep -v amdgpu.ids gpurun_out/$TAG.log | tail -60
