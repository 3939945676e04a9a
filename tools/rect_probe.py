import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
from bench_configs import run
import numpy as np
for n in (64, 65, 96, 128):
    run(f"3D linear regular {n}^3", "linear", "regular", n, 3, 100_000_000)
    run(f"3D linear rectilinear {n}^3", "linear", "rectilinear", n, 3, 100_000_000)
run("2D linear regular 64^2", "linear", "regular", 64, 2, 100_000_000)
run("2D linear rectilinear 64^2", "linear", "rectilinear", 64, 2, 100_000_000)
run("2D linear regular 512^2", "linear", "regular", 512, 2, 100_000_000)
run("2D linear rectilinear 512^2", "linear", "rectilinear", 512, 2, 100_000_000)
run("3D nearest regular 64^3", "nearest", "regular", 64, 3, 100_000_000)
run("3D nearest rectilinear 64^3", "nearest", "rectilinear", 64, 3, 100_000_000)
run("4D linear regular 32^4", "linear", "regular", 32, 4, 100_000_000)
run("4D linear rectilinear 32^4", "linear", "rectilinear", 32, 4, 100_000_000)
