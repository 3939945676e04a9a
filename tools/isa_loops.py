#!/usr/bin/env python3
"""List the loops (backward branches) of one kernel in a hipcc --save-temps .s file with their
instruction mix: VALU / f64 VALU / SALU / LDS / VMEM / scratch / v_mov.  Usage:
  isa_loops.py file.s kernel-symbol-substring"""
import re, sys, collections

def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sym in l and ": ;" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    label_at = {}
    insts = []  # (idx, text)
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        insts.append(s.split(";")[0].strip())
    def kind(t):
        op = t.split()[0]
        if op.startswith("scratch_"): return "scratch"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "vmov"
        if op.startswith("v_") and ("f64" in op): return "valu64"
        if op.startswith("v_"): return "valu"
        if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
        if op.startswith("s_"): return "salu"
        if op.startswith("ds_"): return "lds"
        if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"): return "vmem"
        return "other"
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i, m.group(1)))
    print(f"kernel instructions: {len(insts)}")
    tot = collections.Counter(kind(t) for t in insts)
    print("whole kernel:", dict(tot))
    for a, b, lab in sorted(loops):
        c = collections.Counter(kind(t) for t in insts[a:b + 1])
        print(f"loop {lab}: insts {a}..{b} ({b - a + 1})", dict(c))
    if len(sys.argv) > 3:
        a, b = int(sys.argv[3]), int(sys.argv[4])
        ops = collections.Counter(t.split()[0] for t in insts[a:b + 1])
        for op, n in ops.most_common(60):
            print(f"  {op:32s} {n}")

main()
