#!/usr/bin/env python3
"""Interleaved A/B of the automatic 3-D multilinear launch with and without the device-side sample (cfg2, unordered points)."""
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
out = torch.empty(P, dtype=torch.float64, device=dev)
def t(reps=50):
    for _ in range(5): it.eval_tensors(rnd, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(rnd, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
it.set_option("sweep", -1)
res = {0: [], 1: [], 2: []}
for cyc in range(6):
    for probe in (0, 1, 2):
        it.set_option("sweep_probe", probe)
        res[probe].append(t())
print(json.dumps({"probe_off_ms": [round(x, 4) for x in res[0]], "probe_every_launch_ms": [round(x, 4) for x in res[1]], "probe_thinned_ms": [round(x, 4) for x in res[2]],
                  "median_off": round(float(np.median(res[0])), 4), "median_every": round(float(np.median(res[1])), 4), "median_thinned": round(float(np.median(res[2])), 4),
                  "overhead_every_pct": round(100 * (np.median(res[1]) / np.median(res[0]) - 1), 2),
                  "overhead_thinned_pct": round(100 * (np.median(res[2]) / np.median(res[0]) - 1), 2)}))
