#!/usr/bin/env python3
"""interpn_hip_finish by a status kernel against the 8-byte copy: latency of small device-tensor calls (cfg1's plumbing
case, 1e3 points) and the step of the headline workload (launch + finish), interleaved."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
res = {}
for name, n, nd, P in (("cfg1_1e3", 4, 2, 1000), ("cfg2_1e8", 64, 3, 100_000_000)):
    g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n ** nd)
    it = interpn_amd.Interpolator.regular("linear", [n] * nd, np.full(nd, -1.0), np.full(nd, g[1] - g[0]), vals)
    obs = [torch.rand(P, dtype=torch.float64, device=dev) * 2 - 1 for _ in range(nd)]
    out = torch.empty(P, dtype=torch.float64, device=dev)
    reps = 2000 if P < 10_000 else 60
    rows = {0: [], 1: []}
    for cyc in range(5):
        for fk in (0, 1):
            it.set_option("finish_kernel", fk)
            for _ in range(10): it.eval_tensors(obs, out); it.finish()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): it.eval_tensors(obs, out); it.finish()
            rows[fk].append((time.perf_counter() - t0) / reps * 1e6)
    res[name] = {"copy_us_per_step": round(float(np.median(rows[0])), 2), "kernel_us_per_step": round(float(np.median(rows[1])), 2)}
    it.close()
print(json.dumps(res))
