#!/usr/bin/env python3
"""The sweep kernel's table layout on grids beyond the L2: (1,1) (one line per cell, 5.3x the grid) against (1,2) (1.5 lines
per cell, 2.7x) — ms per 1e8 unordered points, f64 regular, each layout in a child process (INTERPN_HIP_SWEEP_LAYOUT is read
when the handle is made), twice, interleaved; results compared through a checksum."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["AB_ROOT"])
import interpn_amd
dev = torch.device("cuda:0")
P = 100_000_000
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
res = {}
for n in [int(x) for x in os.environ["AB_SIZES"].split(",")]:
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** 3)
    it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals)
    it.set_option("sweep_probe", 0); it.set_option("sweep", 1)
    for _ in range(8): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(24):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    res[str(n)] = {"ms": round(float(np.median(ts)), 4), "layout": it.get_option("sweep_layout"), "kernel": it.kernel_name()[:60],
                   "sum": float(out.view(torch.int64).sum().item() & 0xFFFFFFFFFFFF)}
    it.close()
print("AB " + json.dumps(res), flush=True)
'''
sizes = sys.argv[1] if len(sys.argv) > 1 else "96,128,160"
for rep in range(2):
    for lay in ("11", "12"):
        env = dict(os.environ, INTERPN_HIP_SWEEP_LAYOUT=lay, AB_ROOT=ROOT, AB_SIZES=sizes)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(lay, line[0][3:] if line else ("FAILED " + r.stderr[-600:]), flush=True)
