#!/bin/bash
# usage: gpurun -- ./tools/gpu_probe.sh <tag> <python script and args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:?tag}; shift; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
timeout 900 python3 "$@" 2>&1 | grep -v amdgpu.ids | tee $OUT/out.txt
