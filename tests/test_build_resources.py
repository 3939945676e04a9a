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
