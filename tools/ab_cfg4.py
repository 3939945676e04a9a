#!/usr/bin/env python3
"""Same-box A/B of library builds on cfg4 (4-D multicubic 32^4, 1e7 points, sorted + column kernel) and its rectilinear / f32 twins:
  gpurun -- python3 tools/ab_cfg4.py libA.so libB.so ..."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["AB_ROOT"])
import interpn_amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
P = 10_000_000
def t(it, obs, out, reps=25):
    for _ in range(6): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
res = {}
for name, n, rect, dtype, lin in (("cfg4", 32, False, np.float64, False), ("cfg4_lin", 32, False, np.float64, True), ("rect", 32, True, np.float64, False), ("f32", 32, False, np.float32, False), ("n48", 48, False, np.float64, False)):
    td = torch.float64 if dtype == np.float64 else torch.float32
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** 4).astype(dtype)
    if rect:
        rng = np.random.default_rng(2); gr = []
        for _ in range(4):
            a = g.copy(); a[1:-1] += (rng.random(n - 2) - 0.5) * 0.5 * (g[1] - g[0]); gr.append(a.astype(dtype))
        it = interpn_amd.Interpolator.rectilinear("cubic", gr, vals, linearize_extrapolation=lin)
    else:
        it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype), np.full(4, g[1] - g[0], dtype), vals, linearize_extrapolation=lin)
    obs = [torch.rand(P, dtype=td, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
    out = torch.empty(P, dtype=td, device=dev)
    res[name] = t(it, obs, out)
    assert it.last_path == "binned", it.last_path
    it.close(); del obs, out
print("AB " + json.dumps(res), flush=True)
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INTERPN_AMD_LIB=os.path.join(ROOT, lib), AB_ROOT=ROOT)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(lib.ljust(40), line[0][3:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
