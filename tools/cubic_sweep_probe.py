#!/usr/bin/env python3
"""3-D (2-D with "2d") multicubic: the sweep kernel (cubic_sweep.h, option sweep = 1) against the tiled kernel in place (sweep = 0) on the same
handle — bitwise comparison and HIP-event medians.   python tools/cubic_sweep_probe.py [f32] [rect] [2d] n... [points=1e7,...]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(5)
dtype = np.float32 if "f32" in sys.argv else np.float64
tdt = torch.float32 if dtype == np.float32 else torch.float64
rect = "rect" in sys.argv
ND = 2 if "2d" in sys.argv else 3
sizes = [int(float(v.split("=")[1])) for v in sys.argv[1:] if v.startswith("points=")] or [10_000_000, 30_000_000]
for n in [int(v) for v in sys.argv[1:] if v.isdigit()] or [64]:
    g = np.linspace(-1.0, 1.0, n)
    rng = np.random.default_rng(n)
    vals = rng.uniform(-1, 1, n ** ND).astype(dtype)
    if rect:
        grids = []
        for _ in range(ND):
            j = (rng.random(n) - 0.5) * 0.5 * (g[1] - g[0]); j[0] = j[-1] = 0.0
            grids.append((g + j).astype(dtype))
        it = interpn_amd.Interpolator.rectilinear("cubic", grids, vals)
    else:
        it = interpn_amd.Interpolator.regular("cubic", [n] * ND, np.full(ND, -1.0, dtype), np.full(ND, g[1] - g[0], dtype), vals)
    for P in sizes:
        obs = [(torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05).to(tdt) for _ in range(ND)]
        outs, res, names = {}, {}, {}
        for mode in (0, 1, 0, 1):
            it.set_option("sweep", mode)
            out = torch.empty(P, dtype=tdt, device=dev)
            ev = []
            for _ in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); it.eval_tensors(obs, out); b.record(); ev.append((a, b))
            it.finish()
            res.setdefault(mode, []).extend(a.elapsed_time(b) for a, b in ev[4:])
            outs[mode] = out
            names[mode] = (it.last_path, it.kernel_name())
        same = bool(torch.equal(outs[0], outs[1]) or torch.equal(torch.nan_to_num(outs[0], nan=7.0), torch.nan_to_num(outs[1], nan=7.0)))
        it.set_option("sweep", -1)
        it.eval_tensors(obs, out); it.finish()
        print(json.dumps({"grid": n, "dtype": np.dtype(dtype).name, "kind": "rectilinear" if rect else "regular", "points": P, "in_place_ms": round(float(np.median(res[0])), 4),
                          "sweep_ms": round(float(np.median(res[1])), 4), "ratio": round(float(np.median(res[1]) / np.median(res[0])), 3), "bitwise_equal": same,
                          "paths": names, "automatic_path": it.last_path}), flush=True)
        del obs, outs, out
    it.close()
