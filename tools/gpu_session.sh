#!/bin/bash
# One GPU-box session: parity tests, the driver's bench commands, the 2-rank path on one GPU.
# Usage: gpurun --timeout 2400 -- ./tools/gpu_session.sh [tag]
TAG=${1:-r02g}
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd $R
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 rc=$?" | tee -a $OUT/summary.txt
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --same-device > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; echo "bench n2 gloo rc=$?" | tee -a $OUT/summary.txt
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --same-device > $OUT/bench_n2_torchrun.json 2> $OUT/bench_n2_torchrun.err; echo "bench n2 torchrun rc=$?" | tee -a $OUT/summary.txt
head -c 3000 $OUT/bench_n1.json
echo
head -c 1500 $OUT/bench_n2_gloo.json
