#!/usr/bin/env python3
"""Binned evaluation of the tiled multicubic kernels (interpn_host.h "Binned evaluation"): the same
handle with option binned = 0 (points evaluated as given), 1 (counting sort by tile position
first; sorted order dealt out to the XCDs: option deal = 1, or not: 0) and -1 (the library's own
choice), batch sizes 2^18 .. 3.5e7 (the last one spans two slices), results compared bitwise.
Times are whole evaluations (sort launches included), HIP events, median of 8."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
PMAX = 36_000_000
gen = torch.Generator(device=dev); gen.manual_seed(11)
obs_all = [torch.rand(PMAX, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(4)]
out = torch.empty(PMAX, dtype=torch.float64, device=dev)


def timed(it, o, res, reps):
    it.eval_tensors(o, res); it.finish()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(o, res); b.record(); ev.append((a, b))
    it.finish()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


cases = [("cubic", "regular", 32, 4), ("cubic", "rectilinear", 32, 4), ("cubic", "regular", 64, 3), ("cubic", "rectilinear", 64, 3), ("cubic", "regular", 512, 2)]
only = sys.argv[1:]
for method, kind, n, N in cases:
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(4).uniform(-1, 1, n ** N)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular(method, [n] * N, np.full(N, -1.0), np.full(N, g[1] - g[0]), vals, linearize_extrapolation=False)
    else:
        rng = np.random.default_rng(2)
        axes = []
        for d in range(N):
            a = g.copy(); a[1:-1] += (rng.uniform(size=n - 2) - 0.5) * 0.5 * (g[1] - g[0]); axes.append(a)
        it = interpn_amd.Interpolator.rectilinear(method, axes, vals, linearize_extrapolation=False)
    for P in (1 << 18, 1 << 20, 3_000_000, 10_000_000, 35_000_000):
        if N == 3 and P > 10_000_000: continue
        o = [x[:P] for x in obs_all[:N]]
        res = out[:P]
        row = {"grid": f"{n}^{N}", "kind": kind, "points": P}
        it.set_option("binned", 0)
        row["direct_ms"] = round(timed(it, o, res, 8), 4)
        ref = res.clone()
        it.set_option("binned", 1)
        row["binned_ms"] = round(timed(it, o, res, 8), 4)
        row["binned_used"] = it.get_option("last_binned")
        row["equal"] = bool(torch.equal(res, ref))
        it.set_option("deal", 0)
        res.zero_()
        row["binned_nodeal_ms"] = round(timed(it, o, res, 8), 4)
        row["nodeal_equal"] = bool(torch.equal(res, ref))
        it.set_option("deal", 1)
        it.set_option("binned", -1)
        row["auto_ms"] = round(timed(it, o, res, 8), 4)
        row["auto_binned"] = it.get_option("last_binned")
        print(json.dumps(row), flush=True)
    it.close()
