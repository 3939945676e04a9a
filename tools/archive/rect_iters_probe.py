#!/usr/bin/env python3
"""Rectilinear grids searched with per-bucket records: rows per workgroup (option iters_per_block) —
how much of the gap to the regular kernel is the per-workgroup staging of the records?"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, interpn_amd
dev = torch.device("cuda:0"); P = 100_000_000
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs3 = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
def timed(it, obs, reps=7):
    for _ in range(3): it.eval_tensors(obs, out)
    it.finish(); ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ms.append(a.elapsed_time(b))
    return float(np.median(ms))
for nd, n in ((3, 72), (3, 80), (2, 256), (2, 384)):
    rng = np.random.default_rng(n); g = np.linspace(-1.0, 1.0, n); step = g[1] - g[0]
    grids = []
    for _ in range(nd):
        j = (rng.random(n) - 0.5) * 0.5 * step; j[0] = j[-1] = 0; grids.append(g + j)
    vals = rng.uniform(-1, 1, n**nd); obs = obs3[:nd]
    reg = interpn_amd.Interpolator.regular("linear", [n] * nd, np.full(nd, -1.0), np.full(nd, step), vals, False, 0, np.float64)
    row = {"ndims": nd, "n": n, "regular": round(timed(reg, obs), 4)}; reg.close()
    it = interpn_amd.Interpolator.rectilinear("linear", grids, vals, False, 0, np.float64)
    for iters in (0, 4, 8, 16, 32, 64, 128):
        it.set_option("iters_per_block", iters)
        row[f"iters{iters}"] = round(timed(it, obs), 4)
    it.close()
    print(json.dumps(row), flush=True)
