#!/usr/bin/env python3
"""Split brick layouts (linear_brick.h::split_step): 3-D multilinear-regular f64, 1e8 random
points, grid size x split.  "22:S" = steps (2,2) with the first S cells of dimension i at step 1
(S = 0 is the plain (2,2) table, S = n-1 the (1,2) table); "12:S" = steps (1,2) with the first S
cells of dimension j at step 1 (S = n-1 is the (1,1) table).  Interleaved rounds, median; every
layout's output is compared bitwise with the C-order kernel's.

usage: sweep_split.py [grid sizes ...]   (default 64)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
P = 100_000_000
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
ref = torch.empty(P, dtype=torch.float64, device=dev)


def make(n, lay):
    if lay == "auto": os.environ.pop("INTERPN_HIP_BRICKS", None)
    else: os.environ["INTERPN_HIP_BRICKS"] = lay
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
    return interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)


def timed(it, launches=6):
    ms = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); ms.append((a, b))
    it.finish()
    return [a.elapsed_time(b) for a, b in ms]


sizes = [int(a) for a in sys.argv[1:]] or [64]
for n in sizes:
    it0 = make(n, "off")
    it0.eval_tensors(obs, ref); it0.finish(); it0.close()
    lays = ["11", "12", "22"]
    lays += [f"22:{s}" for s in range(4, n - 1, 4)]
    lays += [f"12:{s}" for s in range(8, n - 1, 8)]
    lays += ["auto"]
    its = {}
    for lay in lays:
        it = make(n, lay)
        it.eval_tensors(obs, out); it.finish()
        its[lay] = (it, bool(torch.equal(out, ref)))
    ms = {lay: [] for lay in lays}
    for _ in range(5):
        for lay in lays:
            ms[lay] += timed(its[lay][0])
    for lay in lays:
        it, same = its[lay]
        tb, si, sj = it.table_layout()
        print(json.dumps({"grid": n, "layout": lay, "steps": [si, sj], "split": list(it.table_split()),
                          "table_MiB": round(tb / 2**20, 2), "ms": round(float(np.median(ms[lay])), 4),
                          "ms_min": round(float(np.min(ms[lay])), 4), "bitwise_equal_to_c_order": same,
                          "kernel": it.kernel_name()}), flush=True)
        it.close()
