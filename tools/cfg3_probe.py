import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
from bench_configs import run
for rep in range(3):
    run("cfg2 3D linear regular 64^3", "linear", "regular", 64, 3, 100_000_000)
    run("cfg3 3D linear rectilinear 64^3", "linear", "rectilinear", 64, 3, 100_000_000)
run("nearest rect 64^3", "nearest", "rectilinear", 64, 3, 100_000_000)
run("nearest regular 64^3", "nearest", "regular", 64, 3, 100_000_000)
run("2D rect 64^2", "linear", "rectilinear", 64, 2, 100_000_000)
run("2D regular 64^2", "linear", "regular", 64, 2, 100_000_000)
run("4D rect 32^4", "linear", "rectilinear", 32, 4, 100_000_000)
run("4D regular 32^4", "linear", "regular", 32, 4, 100_000_000)
