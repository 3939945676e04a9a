#!/bin/bash
# Session: parity with the new layouts, N-D layout sweep incl. 4-D cell bricks, cubic tile order.
TAG=${1:-r02b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 900 python3 tools/sweep_linear_nd.py 4 5 6 > $OUT/sweep_linear_nd.txt 2>&1; echo "sweep rc=$?" | tee -a $OUT/summary.txt
cat $OUT/sweep_linear_nd.txt
for ord in planes tiles; do
  for lay in 11 14 44; do
    INTERPN_HIP_CUBIC_ORDER=$ord INTERPN_HIP_BRICKS=$lay timeout 200 python3 tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" 2>&1 | sed "s/^/order=$ord layout=$lay /" | tee -a $OUT/cubic_order.txt
  done
  INTERPN_HIP_CUBIC_ORDER=$ord timeout 200 python3 tools/bench_configs.py --only "extra 3D cubic" 2>&1 | sed "s/^/order=$ord /" | tee -a $OUT/cubic_order.txt
done
