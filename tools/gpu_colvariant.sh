#!/bin/bash
# usage: gpurun -- ./tools/gpu_colvariant.sh <tag> <library>
# The binned multicubic tests of the 4-D rectilinear shapes through another build of the library (INTERPN_AMD_LIB).
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}; TAG=$1; LIB=$2; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd $R
INTERPN_AMD_LIB=$R/$LIB timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_binned_multicubic_evaluation and rectilinear" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
