#!/usr/bin/env python3
"""Shape of the gated one-pass launch behind an automatic sweep launch: ms per 1e8 points on a fine lattice (the one-pass
kernel runs) and on unordered points (it returns at once) for several values of option gated_iters."""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
def t(it, obs, out, reps=15):
    for _ in range(5): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
for method, dims, P in (("nearest", [128] * 3, 100_000_000), ("nearest", [1000, 1000], 100_000_000), ("linear", [1000, 1000], 100_000_000), ("cubic", [512, 512], 30_000_000), ("linear", [64] * 3, 100_000_000)):
    nd = len(dims)
    vals = np.random.default_rng(1).uniform(-1, 1, int(np.prod(dims)))
    it = interpn_amd.Interpolator.regular(method, dims, np.full(nd, -1.0), np.array([2.0 / (n - 1) for n in dims]), vals, linearize_extrapolation=True)
    m = int(np.floor(P ** (1.0 / nd)))
    ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
    mesh = torch.meshgrid(*([ax] * nd), indexing="ij")
    lat = [torch.cat([x.reshape(-1), x.reshape(-1)[:P - m ** nd]]).contiguous() for x in mesh]
    del mesh
    rnd = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(nd)]
    out = torch.empty(P, dtype=torch.float64, device=dev)
    it.set_option("sweep", 0); base_l = t(it, lat, out)
    it.set_option("sweep", -1); it.set_option("sweep_probe", 0); base_r = t(it, rnd, out)
    it.set_option("sweep_probe", 1)
    row = {"method": method, "dims": dims, "one_pass_lattice_ms": base_l, "sweep_random_ms": base_r}
    for gi in (1, 2, 4, 8, 16, 32):
        it.set_option("gated_iters", gi)
        row[f"gi{gi}"] = [t(it, lat, out), t(it, rnd, out)]
    print(json.dumps(row), flush=True)
    it.close(); del lat, rnd, out
