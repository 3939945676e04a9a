#!/usr/bin/env python3
"""4-D multicubic on small-per-axis grids (the common shape of 4-D tables): the layout the
heuristic picks for in-place evaluation against fully overlapped tiles (1,1) evaluated binned
(counting sort + LDS-DMA gather), by grid size and batch size; results compared bitwise.
    python tools/cubic4_layout_probe.py [f32]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

DT = np.float32 if "f32" in sys.argv[1:] else np.float64
TDT = torch.float32 if DT == np.float32 else torch.float64
dev = torch.device("cuda:0")
PMAX = 10_000_000
gen = torch.Generator(device=dev); gen.manual_seed(11)
obs_all = [torch.rand(PMAX, dtype=TDT, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(4)]
out = torch.empty(PMAX, dtype=TDT, device=dev)


def timed(it, o, res, reps=8):
    it.eval_tensors(o, res); it.finish()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(o, res); b.record(); ev.append((a, b))
    it.finish()
    return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)


def make(n, lay):
    if lay is None: os.environ.pop("INTERPN_HIP_BRICKS", None)
    else: os.environ["INTERPN_HIP_BRICKS"] = lay
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(4).uniform(-1, 1, n ** 4).astype(DT)
    return interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype=DT), np.full(4, g[1] - g[0], dtype=DT), vals,
                                            linearize_extrapolation=False, dtype=DT)


for n in (8, 10, 12, 16, 20, 24, 28, 32):
    ita, itb = make(n, None), make(n, "11")
    for P in (1 << 18, 1 << 20, 3_000_000, 10_000_000):
        o = [x[:P] for x in obs_all]; res = out[:P]
        row = {"grid": f"{n}^4", "points": P, "auto_layout": list(ita.table_layout()), "t11_MiB": round(itb.table_layout()[0] / 2**20, 1)}
        ita.set_option("binned", 0); row["auto_inplace_ms"] = timed(ita, o, res); ref = res.clone()
        ita.set_option("binned", -1); row["auto_ms"] = timed(ita, o, res); row["auto_binned"] = ita.get_option("last_binned")
        itb.set_option("binned", 0); row["t11_inplace_ms"] = timed(itb, o, res)
        itb.set_option("binned", 1); row["t11_binned_ms"] = timed(itb, o, res); row["equal"] = bool(torch.equal(res, ref))
        print(json.dumps(row), flush=True)
    ita.close(); itb.close()
