#!/usr/bin/env python3
"""Structured observation sets (re-gridding onto a finer lattice: the commonest real use) against
the benchmark's unordered points: 3-D multilinear / multicubic on a 64^3 grid, ~1e8 points."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
n = 64
g = np.linspace(-1, 1, n)
vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
m = 464  # 464^3 = 9.99e7 lattice points
ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
lat = torch.meshgrid(ax, ax, ax, indexing="ij")
P = m ** 3
sets = {
    "lattice, last dim fastest": [t.reshape(-1).contiguous() for t in lat],
    "lattice, first dim fastest": [t.permute(2, 1, 0).reshape(-1).contiguous() for t in lat],
    "random": [torch.rand(P, dtype=torch.float64, device=dev) * 2 - 1 for _ in range(3)],
}
out = torch.empty(P, dtype=torch.float64, device=dev)
for method in ("linear", "cubic", "nearest"):
    it = interpn_amd.Interpolator.regular(method, [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
    for name, obs in sets.items():
        it.eval_tensors(obs, out); it.finish()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ts.append(a.elapsed_time(b))
        t = sorted(ts)[2]
        print(f"{method:8s} {name:28s} {t:7.3f} ms  {P / t / 1e6:8.1f} Gpts/s  {P * 32 / t / 1e9:6.2f} TB/s", flush=True)
    it.close()
