#!/usr/bin/env python3
"""Same-box A/B of library builds on the other sweep kernels (forced sweep, unordered points; ms per batch, median of 15):
  gpurun -- python3 tools/ab_family.py libA.so libB.so ..."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["AB_ROOT"])
import interpn_amd
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
def t(it, obs, out, reps=15):
    for _ in range(6): it.eval_tensors(obs, out); it.finish()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); b.synchronize(); ts.append(a.elapsed_time(b))
    return round(float(np.median(ts)), 4)
res = {}
for name, method, dims, P, dtype in (("near3_128", "nearest", [128] * 3, 100_000_000, np.float64), ("near2_1000", "nearest", [1000, 1000], 100_000_000, np.float64),
                                     ("lin2_1000", "linear", [1000, 1000], 100_000_000, np.float64), ("lin2_64", "linear", [64, 64], 100_000_000, np.float64),
                                     ("cub3_64", "cubic", [64] * 3, 30_000_000, np.float64), ("cub3_64_f32", "cubic", [64] * 3, 30_000_000, np.float32),
                                     ("cub2_512", "cubic", [512, 512], 30_000_000, np.float64)):
    nd = len(dims)
    td = torch.float64 if dtype == np.float64 else torch.float32
    vals = np.random.default_rng(1).uniform(-1, 1, int(np.prod(dims))).astype(dtype)
    it = interpn_amd.Interpolator.regular(method, dims, np.full(nd, -1.0, dtype), np.array([2.0 / (n - 1) for n in dims], dtype), vals, linearize_extrapolation=True)
    it.set_option("sweep", 1)
    obs = [torch.rand(P, dtype=td, device=dev, generator=gen) * 2 - 1 for _ in range(nd)]
    out = torch.empty(P, dtype=td, device=dev)
    res[name] = t(it, obs, out)
    assert it.last_path == "sweep"
    it.close(); del obs, out
print("AB " + json.dumps(res), flush=True)
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, INTERPN_AMD_LIB=os.path.join(ROOT, lib), AB_ROOT=ROOT)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
        print(lib.ljust(40), line[0][3:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
