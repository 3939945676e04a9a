#!/usr/bin/env python3
"""Automatic 3-D multilinear launches on coherent and on unordered batches: the device's verdict, and the results against the
forced sweep and the forced brick kernel bit for bit (the gated pair must leave exactly one kernel's output)."""
import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
n = 64; P = 20_000_000
g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
for dtype in (np.float64, np.float32):
    td = torch.float64 if dtype == np.float64 else torch.float32
    it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0, dtype), np.full(3, g[1] - g[0], dtype), vals.astype(dtype))
    m = 272
    ax = torch.linspace(-1, 1, m, dtype=td, device=dev)
    lat = [t.reshape(-1).contiguous() for t in torch.meshgrid(ax, ax, ax, indexing="ij")]
    rnd = [torch.rand(P, dtype=td, device=dev) * 2.1 - 1.05 for _ in range(3)]
    for name, obs in (("lattice", lat), ("random", rnd)):
        it.set_option("sweep", 1); a = it.eval_tensors(obs).clone(); it.finish()
        it.set_option("sweep", 0); b = it.eval_tensors(obs).clone(); it.finish()
        it.set_option("sweep", -1)
        out = torch.full_like(a, -7.0)
        it.eval_tensors(obs, out); it.finish()
        print(dtype.__name__, name, "path", it.last_path, "verdict", it.get_option("sweep_probe_took_brick"), "auto==sweep", bool(torch.equal(out, a)), "auto==brick", bool(torch.equal(out, b)), flush=True)
    it.close()
