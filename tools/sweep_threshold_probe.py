#!/usr/bin/env python3
"""From how many points on does the sweep kernel beat the brick kernel?  (the automatic rule's batch threshold,
k_linear_sweep.hip::sweep_applies)  Product library, same handle, option "sweep" 0 / 1, median of launches."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
dtype = np.float32 if "f32" in sys.argv else np.float64
tdt = torch.float32 if dtype == np.float32 else torch.float64
rect = "rect" in sys.argv
SIZES = (500_000, 1_000_000, 2_000_000, 4_000_000, 6_000_000, 8_000_000, 12_000_000, 16_000_000, 32_000_000, 100_000_000)
if "big" in sys.argv:
    SIZES = (32_000_000, 100_000_000)
for n in [int(v) for v in sys.argv[1:] if v.isdigit()] or [64, 80, 128]:
    g = np.linspace(-1.0, 1.0, n)
    rng = np.random.default_rng(n)
    vals = rng.uniform(-1, 1, n ** 3).astype(dtype)
    if rect:
        grids = []
        for _ in range(3):
            j = (rng.random(n) - 0.5) * 0.5 * (g[1] - g[0]); j[0] = j[-1] = 0.0
            grids.append((g + j).astype(dtype))
        it = interpn_amd.Interpolator.rectilinear("linear", grids, vals)
    else:
        it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0, dtype), np.full(3, g[1] - g[0], dtype), vals)
    for P in SIZES:
        obs = [(torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0).to(tdt) for _ in range(3)]
        out = torch.empty(P, dtype=tdt, device=dev)
        res = {}
        for mode in (0, 1, 0, 1):
            it.set_option("sweep", mode)
            ev = []
            for _ in range(14):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); it.eval_tensors(obs, out); b.record(); ev.append((a, b))
            it.finish()
            res.setdefault(mode, []).extend(a.elapsed_time(b) for a, b in ev[4:])
        it.set_option("sweep", -1)
        it.eval_tensors(obs, out); it.finish()
        print(json.dumps({"grid": n, "dtype": np.dtype(dtype).name, "kind": "rectilinear" if rect else "regular", "points": P, "brick_us": round(float(np.median(res[0])) * 1e3, 1), "sweep_us": round(float(np.median(res[1])) * 1e3, 1),
                          "ratio": round(float(np.median(res[1]) / np.median(res[0])), 3), "automatic_path": it.last_path}), flush=True)
        del obs, out
    it.close()
