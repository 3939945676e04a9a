#!/bin/bash
TAG=${1:-r02c}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 400 python3 tools/fuzz_parity.py 240 11 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/fuzz.log | tee -a $OUT/summary.txt
timeout 600 python3 tools/sweep_linear_nd.py 4 5 6 > $OUT/sweep_linear_nd.txt 2>&1
cat $OUT/sweep_linear_nd.txt
timeout 300 python3 tools/bench_configs.py > $OUT/bench_configs.txt 2>&1
cat $OUT/bench_configs.txt
