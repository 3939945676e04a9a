#!/usr/bin/env python3
"""Shape of the gated brick launch on the two lattice orders (464^3 over a 64^3 f64 grid, 1e8 points): ms per launch."""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
n = 64; P = 100_000_000; m = 464
g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
lat = torch.meshgrid(ax, ax, ax, indexing="ij")
pad = P - m ** 3
sets = {"lattice_c": [torch.cat([t.reshape(-1), t.reshape(-1)[:pad]]).contiguous() for t in lat],
        "lattice_f": [torch.cat([t.permute(2, 1, 0).reshape(-1), t.permute(2, 1, 0).reshape(-1)[:pad]]).contiguous() for t in lat]}
out = torch.empty(P, dtype=torch.float64, device=dev)
def t(obs, reps=15):
    for _ in range(5): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
for name, obs in sets.items():
    row = {"dist": name}
    it.set_option("sweep", 1); row["sweep"] = t(obs)
    it.set_option("sweep", 0); row["brick"] = t(obs)
    it.set_option("sweep", -1); it.set_option("sweep_probe", 1)
    for gi in (1, 2, 4, 8, 16):
        it.set_option("gated_iters", gi)
        row[f"auto_gi{gi}"] = t(obs)
        row["verdict"] = it.get_option("sweep_probe_took_brick")
    for ipb in (2, 4, 8):
        it.set_option("sweep", 0); it.set_option("iters_per_block", ipb); row[f"brick_ipb{ipb}"] = t(obs)
    it.set_option("iters_per_block", 0)
    print(json.dumps(row), flush=True)
