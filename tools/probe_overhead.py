#!/usr/bin/env python3
"""Cost of the automatic 3-D multilinear launch's sampling kernel and gated brick launch on unordered points (cfg2), and what
coherent batches gain, for several shapes of the gated launch:  gpurun -- python3 tools/probe_overhead.py"""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
n = 64; P = 100_000_000
g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
gen = torch.Generator(device=dev); gen.manual_seed(3)
rnd = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
m = 464
ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
lat = [torch.cat([t.reshape(-1), t.reshape(-1)[:P - m ** 3]]).contiguous() for t in torch.meshgrid(ax, ax, ax, indexing="ij")]
out = torch.empty(P, dtype=torch.float64, device=dev)
def t(obs, reps=40):
    for _ in range(6): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
for probe, iters in ((0, 8), (1, 1), (1, 4), (1, 8), (1, 16), (1, 64), (0, 8)):
    it.set_option("sweep", -1); it.set_option("sweep_probe", probe); it.set_option("gated_iters", iters)
    print(json.dumps({"probe": probe, "gated_iters": iters, "random_ms": t(rnd), "lattice_ms": t(lat)}), flush=True)
