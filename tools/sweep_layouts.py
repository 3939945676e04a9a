#!/usr/bin/env python3
"""Sweep grid sizes x brick layouts for 3-D multilinear f64 (validates the layout heuristic of
abi_layout.hip::maybe_build_bricks).  Run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
import interpn_amd
n = int(sys.argv[1]); P = 50_000_000
dev = torch.device("cuda:0")
g = np.linspace(-1, 1, n); vals = np.random.default_rng(1).uniform(-1, 1, n**3)
it = interpn_amd.Interpolator.regular("linear", [n]*3, np.full(3,-1.0), np.full(3, g[1]-g[0]), vals)
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen)*2-1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
for _ in range(2): it.eval_tensors(obs, out)
it.finish()
ms = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ms.append(a.elapsed_time(b))
print(sorted(ms)[2])
''' % ROOT
for n in (16, 32, 48, 56, 64, 72, 80, 96, 128, 192, 256):
    row = {}
    for lay in ("off", "22", "12", "11", ""):
        env = dict(os.environ); env["INTERPN_HIP_BRICKS"] = lay
        if lay == "": env.pop("INTERPN_HIP_BRICKS")
        try:
            r = subprocess.run([sys.executable, "-c", code, str(n)], env=env, capture_output=True, text=True, timeout=120)
            row[lay or "auto"] = round(float(r.stdout.strip().splitlines()[-1]), 3)
        except Exception as e:
            row[lay or "auto"] = "err"
    print(n, json.dumps(row), flush=True)
