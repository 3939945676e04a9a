"""Regular vs rectilinear / nearest / 2-D quick timings (1e8 random points)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
from bench_configs import run
for rep in range(2):
    for ppl in ("0", "1"):
        os.environ["INTERPN_HIP_PPL"] = ppl
        run(f"nearest regular 64^3 ppl={ppl}", "nearest", "regular", 64, 3, 100_000_000)
        run(f"nearest rect 64^3 ppl={ppl}", "nearest", "rectilinear", 64, 3, 100_000_000)
        run(f"2D regular 64^2 ppl={ppl}", "linear", "regular", 64, 2, 100_000_000)
        run(f"2D rect 64^2 ppl={ppl}", "linear", "rectilinear", 64, 2, 100_000_000)
        run(f"2D regular 512^2 ppl={ppl}", "linear", "regular", 512, 2, 100_000_000)
        run(f"2D regular 1000^2 ppl={ppl}", "linear", "regular", 1000, 2, 100_000_000)
    os.environ.pop("INTERPN_HIP_PPL")
