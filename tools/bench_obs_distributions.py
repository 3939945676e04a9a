#!/usr/bin/env python3
"""The sweep and the one-pass (brick) kernel of 3-D multilinear on cfg2's grid (64^3 f64, 1e8 points) over observation
sets that are not i.i.d. uniform, and what the automatic rule picks for each: one JSON line per (distribution, path).
  gpurun -- python3 tools/bench_obs_distributions.py [n] > gpurun_out/obs.jsonl
Distributions (the same five tests/test_gpu_parity.py::test_sweep_family_on_structured_observation_sets checks for bits):
uniform | one_cell | sorted (by the leading coordinate) | lattice (464^3, last dimension fastest / first dimension fastest) |
on_planes (every coordinate a grid coordinate) | half_nan is a failing batch on regular grids and is timed on the
rectilinear twin of the grid only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = 100_000_000
g = np.linspace(-1.0, 1.0, n)
step = g[1] - g[0]
vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
gen = torch.Generator(device=dev)
gen.manual_seed(3)


def uniform():
    return [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]


def make(dist):
    if dist == "uniform":
        return uniform()
    if dist == "one_cell":
        c = [17, 40, 5]
        return [(torch.rand(P, dtype=torch.float64, device=dev, generator=gen) + c[d]) * step - 1.0 for d in range(3)]
    if dist == "sorted":
        obs = uniform()
        order = torch.argsort(obs[0])
        return [o[order].contiguous() for o in obs]
    if dist.startswith("lattice"):
        m = 464
        ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
        lat = torch.meshgrid(ax, ax, ax, indexing="ij")
        pad = P - m ** 3
        if dist == "lattice_c":
            obs = [t.reshape(-1) for t in lat]
        else:
            obs = [t.permute(2, 1, 0).reshape(-1) for t in lat]
        return [torch.cat([o, o[:pad]]).contiguous() for o in obs]
    if dist == "on_planes":
        gt = torch.from_numpy(g).to(dev)
        return [gt[torch.randint(0, n, (P,), device=dev, generator=gen)].contiguous() for _ in range(3)]
    if dist == "half_nan":
        obs = uniform()
        obs[1][P // 2::2] = float("nan")
        return obs
    raise ValueError(dist)


def time_it(it, obs, out, reps=9):
    for _ in range(4):  # (the sweep kernel's period settles over the first launches through a scratch block)
        it.eval_tensors(obs, out)
        try:
            it.finish()
        except AssertionError:
            pass
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        it.eval_tensors(obs, out)
        b.record()
        try:
            it.finish()
        except AssertionError:
            pass
        b.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))


out = torch.empty(P, dtype=torch.float64, device=dev)
reg = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, step), vals)
PROBE = int(os.environ.get("OBS_SWEEP_PROBE", "1"))  # 1: a sample in front of every automatic launch (the steady-state choice per distribution; the default, 2, thins the samples out and would carry one distribution's history into the next)
gr = [g.copy() for _ in range(3)]
rng = np.random.default_rng(2)
for a in gr:
    a[1:-1] += (rng.random(n - 2) - 0.5) * 0.5 * step
rect = interpn_amd.Interpolator.rectilinear("linear", gr, vals)
for dist in ("uniform", "one_cell", "sorted", "lattice_c", "lattice_f", "on_planes", "half_nan"):
    obs = make(dist)
    for kind, it in (("regular", reg), ("rectilinear", rect)):
        if dist == "half_nan" and kind == "regular":
            continue
        row = {"grid": n, "points": P, "dist": dist, "kind": kind}
        it.set_option("sweep_probe", PROBE)
        for name, opt in (("sweep", 1), ("brick", 0), ("auto", -1)):
            it.set_option("sweep", opt)
            row[name + "_ms"] = round(time_it(it, obs, out), 4)
            if name == "auto":
                took = it.get_option("sweep_probe_took_brick")
                row["auto_path"] = it.last_path if took < 0 else ("brick (device verdict)" if took else "sweep (device verdict)")
        best = min(row["sweep_ms"], row["brick_ms"])
        row["auto_over_best"] = round(row["auto_ms"] / best, 3)
        print(json.dumps(row), flush=True)
    del obs
