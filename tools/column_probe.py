#!/usr/bin/env python3
"""cfg4 (4-D multicubic-regular 32^4, 1e7 points): the binned evaluation's variants side by side —
in place, tiled kernel on sorted points (round 2), LDS-column kernel with scattered stores, LDS-column
kernel with sorted stores + un-permutation — each checked bit for bit against the in-place result.

    python tools/column_probe.py [n] [points] [f32]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import interpn_amd

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    dtype = np.float32 if "f32" in sys.argv else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    g = np.linspace(-1.0, 1.0, n)
    vals = rng.uniform(-1, 1, n**4).astype(dtype)
    it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype=dtype), np.full(4, g[1] - g[0], dtype=dtype),
                                          vals, False, 0, dtype)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    obs = [torch.rand(P, dtype=tdt, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(4)]
    out = torch.empty(P, dtype=tdt, device=dev)

    def timed(reps=15):
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
        return float(np.median(ms)), float(np.min(ms))

    it.set_option("binned", 0)
    t = timed(5)
    ref = out.clone()
    print(json.dumps({"variant": "in place", "ms": round(t[0], 4), "min": round(t[1], 4), "kernel": it.kernel_name()}), flush=True)
    variants = []
    for rep in range(2):
        variants.append(("tiled kernel on sorted points", {"binned": 1, "column": 0}))
        variants.append(("column", {"binned": 1, "column": 1}))
    for name, opts in variants:
        for k, v in opts.items():
            it.set_option(k, v)
        out.fill_(-3.0)
        t = timed()
        same = bool(torch.equal(out, ref))
        print(json.dumps({"points": P, "variant": name, "ms": round(t[0], 4), "min": round(t[1], 4), "kernel": it.kernel_name(),
                          "path": it.last_path, "bit_identical_to_in_place": same}), flush=True)
    it.close()


if __name__ == "__main__":
    main()
