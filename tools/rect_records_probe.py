#!/usr/bin/env python3
"""Rectilinear axes too long for the lane-resident search (65..512 coordinates): regular grid vs
rectilinear with the per-bucket records (one LDS access per cell query) vs rectilinear with the
coordinate + table search of round 2.  ms per 1e8 points (2-D, 3-D multilinear; nearest 3-D)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    P = 100_000_000
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    obs3 = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
    out = torch.empty(P, dtype=torch.float64, device=dev)

    def timed(it, obs, reps=9):
        for _ in range(3):
            it.eval_tensors(obs, out)
        it.finish()
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            it.eval_tensors(obs, out)
            b.record()
            it.finish()
            ms.append(a.elapsed_time(b))
        return float(np.median(ms))

    cases = [("linear", 3, n) for n in (65, 72, 80, 100, 128)] + [("linear", 2, n) for n in (128, 256, 384, 512)] + [("nearest", 3, 80)]
    if "forms" in sys.argv:  # full against compact records where both fit
        cases = [("linear", 3, 72), ("linear", 3, 100), ("linear", 2, 256), ("linear", 2, 384)]
    for method, nd, n in cases:
        rng = np.random.default_rng(n)
        g = np.linspace(-1.0, 1.0, n)
        step = g[1] - g[0]
        grids = []
        for _ in range(nd):
            j = (rng.random(n) - 0.5) * 0.5 * step
            j[0] = j[-1] = 0
            grids.append(g + j)
        vals = rng.uniform(-1, 1, n**nd)
        obs = obs3[:nd]
        reg = interpn_amd.Interpolator.regular(method, [n] * nd, np.full(nd, -1.0), np.full(nd, step), vals, False, 0, np.float64)
        t_reg = timed(reg, obs)
        reg.close()
        if "forms" in sys.argv:
            row = {"method": method, "ndims": nd, "n": n, "regular_ms": round(t_reg, 4)}
            for form in (1, 2):
                os.environ["INTERPN_HIP_AXIS_REC_FORM"] = str(form)
                it = interpn_amd.Interpolator.rectilinear(method, grids, vals, False, 0, np.float64)
                row[f"form{form}_mode"] = it.get_option("axis_rec_mode")
                row[f"form{form}_ms"] = round(timed(it, obs), 4)
                it.close()
            os.environ.pop("INTERPN_HIP_AXIS_REC_FORM")
            print(json.dumps(row), flush=True)
            continue
        it = interpn_amd.Interpolator.rectilinear(method, grids, vals, False, 0, np.float64)
        t_rec = timed(it, obs)
        mode = it.get_option("axis_rec_mode")
        ref = out.clone()
        kn = it.kernel_name()
        it.set_option("axis_records", 0)
        t_old = timed(it, obs)
        same = bool(torch.equal(out, ref))
        it.close()
        print(json.dumps({"method": method, "ndims": nd, "n": n, "regular_ms": round(t_reg, 4), "rect_records_ms": round(t_rec, 4),
                          "rect_round2_search_ms": round(t_old, 4), "records_vs_regular": round(t_rec / t_reg, 3),
                          "round2_vs_regular": round(t_old / t_reg, 3), "same_bits": same, "rec_mode": mode, "kernel": kn}), flush=True)


if __name__ == "__main__":
    main()
