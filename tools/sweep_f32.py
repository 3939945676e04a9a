#!/usr/bin/env python3
"""3-D multilinear f32: grid size x brick layout (is the f64-derived layout rule right for f32?)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd
dev = torch.device("cuda:0")
P = 100_000_000
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs = [torch.rand(P, dtype=torch.float32, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float32, device=dev)
for n in (32, 48, 56, 64, 68, 72, 80, 96, 112, 128, 160):
    row = {}
    for lay in ("22", "12", "11", "j4", "auto"):
        os.environ.pop("INTERPN_HIP_PPL", None)
        if lay == "auto": os.environ.pop("INTERPN_HIP_BRICKS", None)
        elif lay == "j4ppl2":
            os.environ["INTERPN_HIP_BRICKS"] = "j4"
            os.environ["INTERPN_HIP_PPL"] = "2"
        else: os.environ["INTERPN_HIP_BRICKS"] = lay
        g = np.linspace(-1, 1, n).astype(np.float32)
        vals = np.random.default_rng(1).uniform(-1, 1, n ** 3).astype(np.float32)
        it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0, np.float32), np.full(3, g[1] - g[0], np.float32), vals, dtype=np.float32)
        for _ in range(2): it.eval_tensors(obs, out)
        it.finish()
        ms = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ms.append(a.elapsed_time(b))
        row[lay] = round(sorted(ms)[3], 3)
        if lay == "auto": row["auto_layout"] = it.table_layout()
        it.close()
    print(n, json.dumps(row), flush=True)
