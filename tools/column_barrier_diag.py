#!/usr/bin/env python3
"""Does the column kernel's source meet s_barrier's contract?  (round-4 review, item 3: the one-group
form on s_barrier hung at 32^4; the product synchronises its wave group with an LDS counter instead.)

Runs a DIAGNOSIS build of the library (cubic_column.h compiled with -DINTERPN_COLUMN_DIAG, see
tools/column_barrier_diag.sh) on cfg4's shape: per part every wave of the workgroup reports how many
group barriers it went through and whether it ever reached one with lanes switched off.  s_barrier
needs (a) the same count from every wave and (b) nothing reached under a partial EXEC that a wave with
an EMPTY exec would also reach.  The diagnosis build keeps the LDS-counter barrier (which cannot hang
on a count mismatch between waves that arrive: it would show here as differing counts)."""
import json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
g = np.linspace(-1.0, 1.0, n)
vals = np.random.default_rng(1).uniform(-1, 1, n ** 4)
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(4)]
for linearize in (False, True):
    it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0), np.full(4, g[1] - g[0]), vals, linearize_extrapolation=linearize)
    ref = it.eval_tensors(obs)
    it.finish()
    parts = 1 << 15
    buf = torch.zeros(parts * 8, dtype=torch.int64, device=dev)
    buf.view(-1, 8)[:, 2] = (1 << 62)
    it.set_option("debug_stamps_bytes", buf.numel() * 8)
    it.set_option("debug_stamps", buf.data_ptr())
    out = it.eval_tensors(obs)
    it.finish()
    it.set_option("debug_stamps", 0)
    st = buf.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] > 0]
    rec = {"grid": n, "points": P, "linearize": linearize, "kernel": it.kernel_name(), "bitwise_equal_to_product_run": bool(torch.equal(out, ref)),
           "parts_reported": int(len(st)), "waves_reporting_per_part": sorted(set(int(v) for v in st[:, 0])),
           "barriers_per_part": sorted(set(int(v) for v in st[:, 1])),
           "parts_where_waves_disagree_on_the_count": int(np.sum(st[:, 1] != st[:, 2])),
           "parts_with_a_barrier_reached_under_partial_exec": int(np.sum(st[:, 3] != 0))}
    print(json.dumps(rec), flush=True)
    it.close()
