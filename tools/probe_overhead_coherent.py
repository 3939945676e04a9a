#!/usr/bin/env python3
"""Coherent batches (a 464^3 lattice on cfg2's grid) through the automatic 3-D multilinear launch: the one-pass kernel
alone (sweep = 0), every launch sampled (sweep_probe = 1: sample + gated pair), and the thinned default (sweep_probe = 2:
after three coherent samples the one-pass kernel alone in the gated launch's shape, sampled every 16th launch)."""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
n = 64; m = 464
g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
ax = torch.linspace(-1.0, 1.0, m, dtype=torch.float64, device=dev)
lat = [t.reshape(-1).contiguous() for t in torch.meshgrid(ax, ax, ax, indexing="ij")]
out = torch.empty(m ** 3, dtype=torch.float64, device=dev)
def t(reps=48):
    for _ in range(6): it.eval_tensors(lat, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(lat, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.mean(ts)), float(np.median(ts))
res = {"one_pass": [], "every": [], "thinned": []}
for cyc in range(5):
    for name, sweep, probe in (("one_pass", 0, 2), ("every", -1, 1), ("thinned", -1, 2)):
        it.set_option("sweep", sweep); it.set_option("sweep_probe", probe)
        res[name].append(t())
o = {k + "_mean_ms": [round(x[0], 4) for x in v] for k, v in res.items()}
med = {k: float(np.median([x[0] for x in v])) for k, v in res.items()}
o.update({"median_of_means": {k: round(v, 4) for k, v in med.items()},
          "every_vs_one_pass_pct": round(100 * (med["every"] / med["one_pass"] - 1), 2),
          "thinned_vs_one_pass_pct": round(100 * (med["thinned"] / med["one_pass"] - 1), 2)})
print(json.dumps(o))
