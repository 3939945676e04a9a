"""Build-time resource check of the recursive-arm kernels (the row-vector form of
k_generic_n returned wrong results where hipcc had spilled its state into AGPRs).

The rule enforced by interpn_amd/csrc/k_generic.hip::generic_vec_ok is: a row-vector (VEC = true)
instantiation exists only where the compiler reports no AGPRs and no scratch.  This test re-derives
that from the compiler's own -Rpass-analysis=kernel-resource-usage remarks on every build, so a new
ROCm that spills a shape now enabled fails here instead of on the GPU."""

import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_row_vector_generic_kernels_do_not_spill(tmp_path):
    from tools.kernel_resources import parse

    src = os.path.join(ROOT, "interpn_amd", "csrc", "k_generic.hip")
    remarks = tmp_path / "remarks.txt"
    with open(remarks, "w") as err:
        subprocess.check_call(
            [HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
             "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "k_generic.o")],
            stderr=err, cwd=os.path.dirname(src))
    rows = [r for r in parse(str(remarks)) if "k_generic_n<" in r["demangled"]]
    vec = [r for r in rows if r["demangled"].split("(")[0].rstrip(">").endswith("true")]
    one = [r for r in rows if r["demangled"].split("(")[0].rstrip(">").endswith("false")]
    assert len(vec) >= 40 and len(one) >= 40, (len(vec), len(one))
    bad = [(r["demangled"], r["vgpr"], r["agpr"], r["scratch"]) for r in vec if r["agpr"] != 0 or r["scratch"] != 0]
    assert not bad, bad
    # the shapes round 1 found broken / spilled must not be compiled in the row-vector form at all
    names = " ".join(r["demangled"] for r in vec)
    for shape in ("k_generic_n<double, 1, 0, true, 8, true>", "k_generic_n<double, 1, 0, true, 7, true>",
                  "k_generic_n<double, 1, 1, true, 6, true>", "k_generic_n<double, 1, 1, false, 8, true>"):
        assert shape not in names, shape
    # and every shape still has its one-tree form, free of scratch
    assert all(r["scratch"] == 0 for r in one)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_column_kernel_barrier_form_and_sweep_kernel_registers(tmp_path):
    """(1) Which barrier the product's column kernel instantiates: its persistent loop synchronises the wave
    group with the LDS-counter barrier (cubic_column.h::col_group_barrier), so the ISA of the product shape
    holds exactly ONE s_barrier — the one in front of the loop.  (The one-group form on s_barrier hung at
    32^4 in round 4; round 5's diagnosis build — tools/column_barrier_diag.py, profiles/NOTES.md section B —
    shows the source meets s_barrier's contract, so the cause is not in the source's barrier sequence; the
    s_barrier form stays behind -DINTERPN_COLUMN_SBARRIER until it has been run under a watchdog.)
    (2) The sweep kernel's product shape must not spill: 12 rows of points live in its registers
    (linear_sweep.h; four more wait in LDS) and three waves per SIMD leave it 168."""
    csrc = os.path.join(ROOT, "interpn_amd", "csrc")
    one = tmp_path / "one.hip"
    one.write_text('#include "cubic_column.h"\n#include "linear_sweep.h"\nusing namespace interpn;\n'
                   "template __global__ void interpn::k_cubic_column<double, false, true, 768, 1, false>(const CubicColumnArgs<double>);\n"
                   "template __global__ void interpn::k_linear_sweep<double, false, true, 1, 1, 12, 768, 0, false, 0, 4>(const SweepArgs<double>);\n")
    asm = tmp_path / "one.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-I", csrc,
                           "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(asm), str(one)], cwd=csrc)
    import re

    text = asm.read_text()
    kernels = {}
    name = None
    for line in text.splitlines():
        lab = re.match(r"^(_Z\w*k_\w+):", line)
        if lab:
            name = lab.group(1)
            kernels[name] = []
        elif name and line.strip().startswith(("s_barrier", "scratch_", ".vgpr_spill_count", ".vgpr_count")):
            kernels[name].append(line.strip())
    col = [k for k in kernels if "k_cubic_column" in k]
    swp = [k for k in kernels if "k_linear_sweep" in k]
    assert len(col) == 1 and len(swp) == 1, list(kernels)
    assert sum(1 for l in kernels[col[0]] if l.startswith("s_barrier")) == 1, kernels[col[0]]
    assert sum(1 for l in kernels[swp[0]] if l.startswith("s_barrier")) == 1, kernels[swp[0]]  # before any wave takes work
    assert not [l for l in kernels[swp[0]] if l.startswith("scratch_")], "the sweep kernel spills"
    m = re.search(r"\.name:\s+_ZN7interpn14k_linear_sweep.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    if m:
        assert int(m.group(1)) <= 168 and int(m.group(2)) == 0, m.groups()
