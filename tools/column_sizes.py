#!/usr/bin/env python3
"""4-D multicubic on regular n^4 grids, 1e7 points: in place / sorted + tiled kernel / sorted by class
+ LDS-column kernel (K-range phases), each bit-compared with the in-place result.

    python tools/column_sizes.py [n ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import interpn_amd

    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [24, 32, 40, 48, 64]
    kind = "rectilinear" if "rect" in sys.argv else "regular"
    P = 10_000_000
    dev = torch.device("cuda:0")
    for n in sizes:
        rng = np.random.default_rng(4)
        g = np.linspace(-1.0, 1.0, n)
        vals = rng.uniform(-1, 1, n**4)
        if kind == "regular":
            it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0), np.full(4, g[1] - g[0]), vals, False, 0, np.float64)
        else:
            grids = []
            for d in range(4):
                a = g.copy()
                a[1:-1] += (rng.uniform(size=n - 2) - 0.5) * 0.5 * (g[1] - g[0])
                grids.append(a)
            it = interpn_amd.Interpolator.rectilinear("cubic", grids, vals, False, 0, np.float64)
        gen = torch.Generator(device=dev)
        gen.manual_seed(5)
        obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(4)]
        out = torch.empty(P, dtype=torch.float64, device=dev)

        def timed(reps=9):
            for _ in range(2):
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
            return round(float(np.median(ms)), 4)

        row = {"n": n, "kind": kind}
        it.set_option("binned", 0)
        row["in_place_ms"] = timed(3)
        ref = out.clone()
        for name, opts in (("tiled_sorted", {"binned": 1, "column": 0}), ("column", {"binned": 1, "column": 1}), ("auto", {"binned": -1, "column": -1})):
            for k, v in opts.items():
                it.set_option(k, v)
            out.fill_(-3.0)
            row[name + "_ms"] = timed()
            row[name + "_kernel"] = it.kernel_name().replace("interpn::", "")[:40]
            row[name + "_same"] = bool(torch.equal(out, ref))
        print(json.dumps(row), flush=True)
        it.close()
        del obs, out, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
