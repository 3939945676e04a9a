#!/usr/bin/env python3
"""Same-box A/B of library builds on the one-pass brick kernel (option sweep = 0): unordered points and a fine lattice,
64^3 and 128^3 f64, 3e7 points, and 4-D 32^4 multilinear:  gpurun -- python3 tools/ab_brick.py libA.so libB.so ..."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["AB_ROOT"])
import interpn_amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
def t(it, obs, out, reps=20):
    for _ in range(6): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
res = {}
P = 30_000_000
for name, n, nd in (("r64", 64, 3), ("r128", 128, 3), ("r32_4d", 32, 4)):
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** nd)
    it = interpn_amd.Interpolator.regular("linear", [n] * nd, np.full(nd, -1.0), np.full(nd, g[1] - g[0]), vals)
    it.set_option("sweep", 0)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(nd)]
    out = torch.empty(P, dtype=torch.float64, device=dev)
    res[name] = t(it, obs, out)
    if nd == 3:
        m = int(P ** (1 / 3)) + 1
        ax = torch.linspace(-1, 1, m, dtype=torch.float64, device=dev)
        lat = [x.reshape(-1)[:P].contiguous() for x in torch.meshgrid(ax, ax, ax, indexing="ij")]
        res[name + "_lattice"] = t(it, lat, out)
        del lat
    it.close(); del obs, out
print("AB " + json.dumps(res), flush=True)
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INTERPN_AMD_LIB=os.path.join(ROOT, lib), AB_ROOT=ROOT)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(lib.ljust(40), line[0][3:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
