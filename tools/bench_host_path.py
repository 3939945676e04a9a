#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (numpy in, numpy out): never the
headline metric, reported in DESIGN.md section 5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interpn_amd

n, P = 64, 20_000_000
g = np.linspace(-1, 1, n)
rng = np.random.default_rng(0)
vals = rng.uniform(-1, 1, n**3)
obs = [rng.uniform(-1, 1, P) for _ in range(3)]
out = np.zeros(P)
dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0])
it = interpn_amd.MultilinearRegular.new(dims, starts, steps, vals)
it.eval(obs, out)
for name, fn in (("class .eval (resident grid)", lambda: it.eval(obs, out)),
                 ("raw one-shot", lambda: interpn_amd.raw.interpn_linear_regular_f64(dims, starts, steps, vals, obs, out))):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {P/best/1e6:.1f} Mpts/s, {P*32/best/1e9:.1f} GB/s over PCIe (32 B/pt), {best*1e3:.1f} ms for {P} pts", flush=True)
