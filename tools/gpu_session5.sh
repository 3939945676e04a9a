#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r02h}; mkdir -p $OUT; cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 300 python3 tools/bench_host_path.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_path.txt
timeout 300 python3 tools/fuzz_parity.py 120 31 2>&1 | tail -1 | tee -a $OUT/summary.txt
