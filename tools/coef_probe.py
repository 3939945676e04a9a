#!/usr/bin/env python3
"""Column evaluation of 4-D multicubic n^4 grids at 1e7 points: dim 0 from the table values
(`column_coef` 0, the round-3/4 form) against dim 0 from per-part Hermite coefficients
(`column_coef` 1, cubic_column.h "Coefficient columns"), alternating in one process, every result
bit-compared with the in-place evaluation; per-stage times from interpn_hip_stage_ms.

    python tools/coef_probe.py [n ...] [rect] [f32] [lin] [wide]     (wide: points 5 % beyond the grid)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import interpn_amd

    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [32]
    kind = "rectilinear" if "rect" in sys.argv else "regular"
    dtype = np.float32 if "f32" in sys.argv else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    lin = "lin" in sys.argv
    wide = "wide" in sys.argv
    P = 10_000_000
    dev = torch.device("cuda:0")
    for n in sizes:
        rng = np.random.default_rng(4)
        g = np.linspace(-1.0, 1.0, n)
        vals = rng.uniform(-1, 1, n**4).astype(dtype)
        if kind == "regular":
            it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype=dtype), np.full(4, g[1] - g[0], dtype=dtype),
                                                  vals, lin, 0, dtype)
        else:
            grids = []
            for d in range(4):
                a = g.copy()
                a[1:-1] += (rng.uniform(size=n - 2) - 0.5) * 0.5 * (g[1] - g[0])
                grids.append(a.astype(dtype))
            it = interpn_amd.Interpolator.rectilinear("cubic", grids, vals, lin, 0, dtype)
        gen = torch.Generator(device=dev)
        gen.manual_seed(5)
        span, off = (2.1, 1.05) if wide else (2.0, 1.0)
        obs = [torch.rand(P, dtype=tdt, device=dev, generator=gen) * span - off for _ in range(4)]
        out = torch.empty(P, dtype=tdt, device=dev)

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

        it.set_option("binned", 0)
        t0 = timed(3)
        ref = out.clone()
        print(json.dumps({"n": n, "kind": kind, "dtype": np.dtype(dtype).name, "linearize": lin, "wide": wide, "in_place_ms": t0}), flush=True)
        it.set_option("binned", 1)
        it.set_option("column", 1)
        it.set_option("stage_timing", 1)
        shapes = [int(t) for t in os.environ.get("COEF_THREADS", "768").split(",")]
        pads = [int(t) for t in os.environ.get("COEF_PAD", "-1").split(",")]
        tails = [int(t, 0) for t in os.environ.get("COEF_TAIL", "0x84").split(",")]
        parts = [int(t) for t in os.environ.get("COEF_PART", "0").split(",")]
        it.set_option("column_groups", int(os.environ.get("COEF_GROUPS", "1")))
        keys = [int(t) for t in os.environ.get("COEF_KEYS", "1").split(",")]
        for rep in range(2):
            for coef, threads, pad, tail, part, ky in ([(0, shapes[0], pads[0], tails[0], 0, 0)] if "COEF_ONLY" not in os.environ else []) + [(1, t, pd, tl, pt, k) for t in shapes for pd in pads for tl in tails for pt in parts for k in keys]:
                it.set_option("column_coef", coef)
                it.set_option("column_keys", ky)
                it.set_option("column_part", part)
                it.set_option("column_tail", tail)
                it.set_option("column_threads", threads)
                it.set_option("column_pad", pad)
                out.fill_(-3.0)
                ms = timed()
                st = {k: round(v, 4) for k, v in it.stage_ms().items()}
                print(json.dumps({"n": n, "column_coef": coef, "pad": pad, "tail": hex(tail), "part": part, "keys": ky, "ms": ms, "stage_ms": st, "kernel": it.kernel_name().replace("interpn::", "")[:60],
                                  "same": bool(torch.equal(out, ref))}), flush=True)
        it.close()
        del obs, out, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
