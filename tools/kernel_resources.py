#!/usr/bin/env python3
"""Tabulate -Rpass-analysis=kernel-resource-usage output (stderr of hipcc) per kernel:
VGPRs, AGPRs, scratch bytes/lane, SGPRs, SGPR/VGPR spills, occupancy.

    hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip -o /dev/null 2> remarks.txt
    python tools/kernel_resources.py remarks.txt [substring ...]
"""
import re
import subprocess
import sys


def parse(path):
    txt = open(path).read()
    rows = []
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        rows.append(dict(name=name, vgpr=g("    VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         sgpr=g("TotalSGPRs"), sspill=g("SGPRs Spill"), vspill=g("VGPRs Spill"),
                         occ=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r["demangled"] = n.replace("interpn::", "")
    return rows


if __name__ == "__main__":
    rows = parse(sys.argv[1])
    pats = sys.argv[2:]
    print(f"{'vgpr':>5} {'agpr':>5} {'scr':>5} {'sgpr':>5} {'sspl':>5} {'vspl':>5} {'occ':>4} {'lds':>6}  kernel")
    for r in rows:
        if pats and not all(p in r["demangled"] for p in pats):
            continue
        print(f"{r['vgpr']:5d} {r['agpr']:5d} {r['scratch']:5d} {r['sgpr']:5d} {r['sspill']:5d} {r['vspill']:5d} {r['occ']:4d} {r['lds']:6d}  {r['demangled'][:110]}")
