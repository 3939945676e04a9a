#!/usr/bin/env python3
"""Regenerates interpn_amd/raw.pyi (type stubs of the reference-named raw functions) from one
table: family name -> parameter list with `A` standing for the dtype's array type."""
import os

FAMILIES = [
    ("interpn_linear_regular", "dims: Dims, starts: A, steps: A, vals: A, obs: Sequence[A], out: A"),
    ("interpn_linear_rectilinear", "grids: Sequence[A], vals: A, obs: Sequence[A], out: A"),
    ("interpn_nearest_regular", "dims: Dims, starts: A, steps: A, vals: A, obs: Sequence[A], out: A"),
    ("interpn_nearest_rectilinear", "grids: Sequence[A], vals: A, obs: Sequence[A], out: A"),
    ("interpn_cubic_regular", "dims: Dims, starts: A, steps: A, vals: A, linearize_extrapolation: bool, obs: Sequence[A], out: A"),
    ("interpn_cubic_rectilinear", "grids: Sequence[A], vals: A, linearize_extrapolation: bool, obs: Sequence[A], out: A"),
    ("check_bounds_regular", "dims: Dims, starts: A, steps: A, obs: Sequence[A], atol: float, out: Flags"),
    ("check_bounds_rectilinear", "grids: Sequence[A], obs: Sequence[A], atol: float, out: Flags"),
]
HEAD = '''"""Type stubs of interpn_amd.raw: the 16 functions of the reference's `interpn.raw` surface
(src/interpn/raw.pyi:32-147 of jlogan03/interpn v0.8.2), same names, argument order and dtypes.
Generated from the table in tools/gen_raw_stub.py; tests/test_abi_cpu.py checks it against raw.py."""

from collections.abc import Sequence

import numpy as np
from numpy.typing import NDArray

F64 = NDArray[np.float64]
F32 = NDArray[np.float32]
Dims = NDArray[np.intp] | Sequence[int]
Flags = NDArray[np.bool_]

MAXDIMS: int
'''


def render() -> str:
    lines = [HEAD]
    for name, sig in FAMILIES:
        for sfx, ty in (("f64", "F64"), ("f32", "F32")):
            lines.append(f"def {name}_{sfx}({sig.replace('[A]', '[' + ty + ']').replace(': A', ': ' + ty)}) -> None: ...")
    lines += ["", "__all__: list[str]", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "interpn_amd", "raw.pyi"), "w") as f:
        f.write(render())
