#!/usr/bin/env python3
"""Throughput of every (method, kind, N) the ABI accepts, device-resident points, next to the
single-thread CPU oracle on a small sample: a sanity sweep for slow corners (the runtime-N kernel
serving the reference's recursive arms, odd grid sizes...)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
from oracle import pyoracle

dev = torch.device("cuda:0")
DT = np.float32 if os.environ.get("BENCH_MATRIX_DTYPE", "f64") == "f32" else np.float64
TDT = torch.float32 if DT == np.float32 else torch.float64
AX = {1: 4096, 2: 512, 3: 64, 4: 24, 5: 12, 6: 8, 7: 6, 8: 5}
rows = []
for method in ("linear", "cubic", "nearest"):
    for kind in ("regular", "rectilinear"):
        for N in range(1, 9):
            if method == "nearest" and N > 6:
                continue
            n = AX[N]
            P = 4_000_000 if (method != "cubic" or N <= 4) else 200_000
            rng = np.random.default_rng(N)
            grids = [np.linspace(-1.0, 1.0, n).astype(DT) for _ in range(N)]
            if kind == "rectilinear":
                grids = [(g + np.concatenate([[0], (rng.random(n - 2) - 0.5) * 0.4 * (g[1] - g[0]), [0]])).astype(DT) for g in grids]
            vals = rng.uniform(-1, 1, n ** N).astype(DT)
            obs_h = [rng.uniform(-1.02, 1.02, P).astype(DT) for _ in range(N)]
            obs = [torch.from_numpy(o).to(dev) for o in obs_h]
            out = torch.empty(P, dtype=TDT, device=dev)
            dims = [n] * N
            starts = np.full(N, -1.0, dtype=DT); steps = np.full(N, grids[0][1] - grids[0][0], dtype=DT)
            if kind == "regular":
                it = interpn_amd.Interpolator.regular(method, dims, starts, steps, vals, linearize_extrapolation=True)
            else:
                it = interpn_amd.Interpolator.rectilinear(method, grids, vals, linearize_extrapolation=True)
            it.eval_tensors(obs, out); it.finish()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                a.record(); it.eval_tensors(obs, out); b.record(); it.finish()
                best = min(best, a.elapsed_time(b))
            it.close()
            # CPU oracle on a sample
            S = min(P, 200_000 if method != "cubic" or N <= 4 else 20_000)
            sub = [o[:S] for o in obs_h]; o_cpu = np.zeros(S, dtype=DT)
            t0 = time.perf_counter()
            if method == "linear":
                (pyoracle.linear_regular(dims, starts, steps, vals, sub, o_cpu) if kind == "regular" else pyoracle.linear_rectilinear(grids, vals, sub, o_cpu))
            elif method == "cubic":
                (pyoracle.cubic_regular(dims, starts, steps, vals, True, sub, o_cpu) if kind == "regular" else pyoracle.cubic_rectilinear(grids, vals, True, sub, o_cpu))
            else:
                (pyoracle.nearest_regular(dims, starts, steps, vals, sub, o_cpu) if kind == "regular" else pyoracle.nearest_rectilinear(grids, vals, sub, o_cpu))
            tc = time.perf_counter() - t0
            same = bool(np.array_equal(out[:S].cpu().numpy(), o_cpu))
            r = {"method": method, "kind": kind, "N": N, "axis": n, "points": P, "gpu_ms": round(best, 3),
                 "gpu_Mpts": round(P / best / 1e3, 1), "cpu_Mpts": round(S / tc / 1e6, 2),
                 "ratio": round((P / best / 1e3) / (S / tc / 1e6), 1), "bit_identical": same}
            rows.append(r); print(json.dumps(r), flush=True)
