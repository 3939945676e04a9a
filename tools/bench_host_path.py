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

# Small-call latency (BASELINE configs[0]: 2-D 4x4 grid, 1e3 points): the per-call floor of the
# host entry points.
g2 = np.linspace(-1, 1, 4)
v2 = rng.uniform(-1, 1, 16)
o2 = [rng.uniform(-1, 1, 1000) for _ in range(2)]
out2 = np.zeros(1000)
d2, s2, st2 = [4, 4], np.full(2, -1.0), np.full(2, g2[1] - g2[0])
it2 = interpn_amd.MultilinearRegular.new(d2, s2, st2, v2)
it2.eval(o2, out2)
for name, fn in (("class .eval (resident grid), 1e3 pts", lambda: it2.eval(o2, out2)),
                 ("raw one-shot, 1e3 pts", lambda: interpn_amd.raw.interpn_linear_regular_f64(d2, s2, st2, v2, o2, out2)),
                 ("raw one-shot 64^3 grid, 1e3 pts", lambda: interpn_amd.raw.interpn_linear_regular_f64(dims, starts, steps, vals, [o[:1000] for o in obs], out2))):
    fn()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: median {ts[len(ts)//2]*1e6:.0f} us, min {ts[0]*1e6:.0f} us", flush=True)
