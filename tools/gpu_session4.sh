#!/bin/bash
# Long differential fuzz (several seeds) + the 44-cell throughput matrix.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r02e}; mkdir -p $OUT; cd $R
for seed in ${SEEDS:-21 22 23}; do
  timeout $((${SECS:-600} + 100)) python3 tools/fuzz_parity.py ${SECS:-600} $seed > $OUT/fuzz_$seed.log 2>&1; echo "fuzz seed $seed rc=$?" | tee -a $OUT/summary.txt
  tail -2 $OUT/fuzz_$seed.log | tee -a $OUT/summary.txt
done
timeout 900 python3 tools/bench_matrix.py > $OUT/matrix.txt 2>&1; echo "matrix rc=$?" | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/matrix.txt | tail -50
