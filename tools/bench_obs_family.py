#!/usr/bin/env python3
"""The other sweep kernels (nearest 3-D / 2-D, multilinear 2-D, multicubic 3-D / 2-D) against their one-pass kernels on
unordered points and on a fine lattice (last dimension fastest), at sizes their automatic rules take the sweep:
  gpurun -- python3 tools/bench_obs_family.py > gpurun_out/obs_family.jsonl"""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)

def t(it, obs, out, reps=9):
    for _ in range(4): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)

for method, dims, P in (("nearest", [128] * 3, 100_000_000), ("nearest", [1000, 1000], 100_000_000), ("linear", [1000, 1000], 100_000_000),
                        ("cubic", [64] * 3, 30_000_000), ("cubic", [512, 512], 30_000_000)):
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
    for name, obs in (("random", rnd), ("lattice_c", lat)):
        row = {"method": method, "dims": dims, "points": P, "dist": name}
        for key, opt in (("sweep", 1), ("one_pass", 0), ("auto", -1)):
            it.set_option("sweep", opt)
            row[key + "_ms"] = t(it, obs, out)
            if key == "auto":
                row["auto_path"] = it.last_path
        row["auto_over_best"] = round(row["auto_ms"] / min(row["sweep_ms"], row["one_pass_ms"]), 3)
        print(json.dumps(row), flush=True)
    it.close()
    del lat, rnd, out
