#!/usr/bin/env python3
"""1-D multilinear on a rectilinear axis: the per-bucket record kernel (k_linear1_records.hip)
against the general rectilinear kernel (INTERPN_HIP_BRICKS=off) and the regular-grid kernel on the
same number of points per axis; 1e8 random points, f64, median ms; outputs compared bitwise."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
P = 100_000_000
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05]
out = torch.empty(P, dtype=torch.float64, device=dev)
ref = torch.empty(P, dtype=torch.float64, device=dev)


def timed(it, res):
    it.eval_tensors(obs, res); it.finish()
    ev = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, res); b.record(); ev.append((a, b))
    it.finish()
    return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)


rng = np.random.default_rng(3)
for name, n in (("jitter", 512), ("jitter", 4096), ("jitter", 8192), ("jitter", 20000), ("jitter", 65536), ("jitter", 500000), ("log", 4096),
                ("two_scales", 4096), ("two_scales", 50000)):
    g = np.linspace(-1.0, 1.0, n)
    if name == "jitter":
        if n > 2: g[1:-1] += (rng.uniform(size=n - 2) - 0.5) * 0.5 * (g[1] - g[0])
    elif name == "log":
        g = -1.0 + 2.0 * (np.logspace(0, 3, n) - 1.0) / 999.0          # spacing ratio 1000 : 1 end to end
    else:
        g = np.concatenate([np.linspace(-1.0, -0.9, n // 2, endpoint=False), np.linspace(-0.9, 1.0, n - n // 2)])
    vals = rng.uniform(-1, 1, n)
    row = {"axis": name, "n": n}
    os.environ["INTERPN_HIP_BRICKS"] = "off"
    it0 = interpn_amd.Interpolator.rectilinear("linear", [g], vals)
    row["general_ms"] = timed(it0, ref); row["general_kernel"] = it0.kernel_name().split("<")[0]; it0.close()
    os.environ.pop("INTERPN_HIP_BRICKS")
    it1 = interpn_amd.Interpolator.rectilinear("linear", [g], vals)
    row["auto_ms"] = timed(it1, out); row["auto_kernel"] = it1.kernel_name()
    row["table_KiB"] = round(it1.table_layout()[0] / 1024, 1)
    a, b = out.cpu().numpy(), ref.cpu().numpy()
    row["equal"] = bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]))
    it1.close()
    it2 = interpn_amd.Interpolator.regular("linear", [n], np.array([-1.0]), np.array([2.0 / (n - 1)]), vals)
    row["regular_ms"] = timed(it2, out); it2.close()
    print(json.dumps(row), flush=True)
