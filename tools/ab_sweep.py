#!/usr/bin/env python3
"""Same-box A/B of library builds on the 3-D multilinear configurations (ms per 1e8 points, median of 30 after warm-up):
  gpurun -- python3 tools/ab_sweep.py libA.so libB.so ...   (paths relative to the repo; each in a child process, twice, interleaved)"""
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
def t(it, obs, out, reps=30):
    for _ in range(8): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
res = {}
for name, n, rect, dtype in (("cfg2", 64, False, np.float64), ("cfg3", 64, True, np.float64), ("cfg5s", 128, False, np.float64), ("f32_64", 64, False, np.float32), ("f32_rect64", 64, True, np.float32), ("rect80", 80, True, np.float64)):
    td = torch.float64 if dtype == np.float64 else torch.float32
    obs = [torch.rand(P, dtype=td, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
    out = torch.empty(P, dtype=td, device=dev)
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** 3).astype(dtype)
    if rect:
        rng = np.random.default_rng(2); gr = []
        for _ in range(3):
            a = g.copy(); a[1:-1] += (rng.random(n - 2) - 0.5) * 0.5 * (g[1] - g[0]); gr.append(a.astype(dtype))
        it = interpn_amd.Interpolator.rectilinear("linear", gr, vals)
    else:
        it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0, dtype), np.full(3, g[1] - g[0], dtype), vals)
    it.set_option("sweep_probe", 0)
    res[name] = t(it, obs, out)
    it.close(); del obs, out
print("AB " + json.dumps(res), flush=True)
'''
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, INTERPN_AMD_LIB=os.path.join(ROOT, lib), AB_ROOT=ROOT)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(lib.ljust(40), line[0][3:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
