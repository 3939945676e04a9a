"""Rectilinear multicubic through the automatic paths, records of the axes' cells on (default) and off
(INTERPN_HIP_CUBIC_RECORDS=0 in the environment): ms per batch, kernel.
  gpurun -- python3 tools/rect_cubic_probe.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import numpy as np
from bench_configs import run

for dtype in (np.float64, np.float32):
    tag = "f64" if dtype == np.float64 else "f32"
    run(f"2D cubic rectilinear 512^2 {tag}", "cubic", "rectilinear", 512, 2, 10_000_000, dtype=dtype)
    run(f"3D cubic rectilinear 64^3 {tag}", "cubic", "rectilinear", 64, 3, 10_000_000, dtype=dtype)
    run(f"3D cubic rectilinear 128^3 {tag}", "cubic", "rectilinear", 128, 3, 10_000_000, dtype=dtype)
    run(f"4D cubic rectilinear 32^4 {tag}", "cubic", "rectilinear", 32, 4, 10_000_000, dtype=dtype)
    run(f"3D cubic regular 64^3 {tag}", "cubic", "regular", 64, 3, 10_000_000, dtype=dtype)
