#!/bin/bash
# Split-layout session: parity of the layouts, then the size x split sweep.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-split}; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "brick_layouts" > $OUT/pytest_layouts.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_layouts.log | tee -a $OUT/summary.txt
timeout 900 python3 tools/sweep_split.py ${2:-64} 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_split.txt
