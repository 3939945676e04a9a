#!/usr/bin/env python3
"""The headline workload and nothing else, for a PMC pass: `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3
tools/traffic_child.py <n> <points> [launches]` (bench.py starts it as a child process, one counter per pass, after its
timed region: `roofline.traffic` measured in the run itself)."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("INTERPN_BENCH_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda:0")
g = np.linspace(-1.0, 1.0, n)
vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
gen = torch.Generator(device=dev)
gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
for _ in range(launches):
    it.eval_tensors(obs, out)
    it.finish()
print("kernel", it.kernel_name(), flush=True)
it.close()
