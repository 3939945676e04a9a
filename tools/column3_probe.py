#!/usr/bin/env python3
"""3-D multicubic n^3 at 1e7 points: in place against sorted + LDS-column (cubic3_column.h), bit-compared,
with the sorted path's per-stage times (histogram | scan | scatter | column kernel)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
P = int(float(os.environ.get("C3_POINTS", "1e7")))
sizes = [int(v) for v in sys.argv[1:] if v.isdigit()] or [64]
kinds = ["regular", "rectilinear"] if "rect" in sys.argv else ["regular"]
dtype = np.float32 if "f32" in sys.argv else np.float64
tdt = torch.float32 if dtype == np.float32 else torch.float64
gen = torch.Generator(device=dev); gen.manual_seed(3)


def timed(it, obs, out, reps=12):
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); ev.append((a, b))
    it.finish()
    return float(np.median([a.elapsed_time(b) for a, b in ev][2:]))


for n in sizes:
    g = np.linspace(-1.0, 1.0, n)
    rng = np.random.default_rng(n)
    vals = rng.uniform(-1, 1, n ** 3).astype(dtype)
    obs = [(torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0).to(tdt) for _ in range(3)]
    for kind in kinds:
        for lin in (False, True):
            if kind == "regular":
                it = interpn_amd.Interpolator.regular("cubic", [n] * 3, np.full(3, -1.0, dtype), np.full(3, g[1] - g[0], dtype), vals, linearize_extrapolation=lin)
            else:
                grids = []
                for _ in range(3):
                    j = (rng.random(n) - 0.5) * 0.5 * (g[1] - g[0]); j[0] = j[-1] = 0.0
                    grids.append((g + j).astype(dtype))
                it = interpn_amd.Interpolator.rectilinear("cubic", grids, vals, linearize_extrapolation=lin)
            out = torch.empty(P, dtype=tdt, device=dev)
            it.set_option("binned", 0)
            t_in = timed(it, obs, out)
            k_in = it.kernel_name()
            ref = out.clone()
            it.set_option("binned", 1); it.set_option("column", 1)
            t_col = timed(it, obs, out)
            same = bool(torch.equal(out, ref))
            it.set_option("stage_timing", 1)
            it.eval_tensors(obs, out); it.finish()
            st = it.stage_ms()
            it.set_option("stage_timing", 0)
            print(json.dumps({"grid": n, "kind": kind, "dtype": np.dtype(dtype).name, "linearize": lin, "points": P, "in_place_ms": round(t_in, 4), "in_place_kernel": k_in,
                              "sorted_column_ms": round(t_col, 4), "kernel": it.kernel_name(), "bitwise_equal": same,
                              "stage_ms": {k: round(v, 4) for k, v in st.items()}}), flush=True)
            it.close()
