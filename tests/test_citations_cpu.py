"""The `file:line` citations of the reference that the oracle, the C ABI header and the kernels carry (the judge checks parity
through them): every cited file exists in the reference tree and has the cited lines; where the oracle cites a range for a
fused multiply-add site, the reference's text in that range says `mul_add`.  Reads the reference as TEXT only, here in the
build container; skipped where /root/reference does not exist (the GPU box)."""
import glob
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"((?:src/|test/|benches/)?(?:[a-z_]+/)*[a-z_]+\.(?:rs|py|pyi|toml)):(\d+)(?:-(\d+))?")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference tree is not mounted here")


def _candidates(path):
    """Files of the reference a citation may mean: the path as written, under src/, or — for a bare file name — any file of
    that name (the surrounding text says which module; the check accepts the citation if ANY candidate has the lines)."""
    out = []
    for base in ("", "src/", "src/interpn/"):
        p = os.path.join(REF, base + path)
        if os.path.isfile(p):
            out.append(p)
    if not out and "/" not in path:
        out = [p for p in glob.glob(os.path.join(REF, "**", path), recursive=True) if os.path.isfile(p)]
    elif "/" not in path:
        out += [p for p in glob.glob(os.path.join(REF, "src", "**", path), recursive=True) if os.path.isfile(p) and p not in out]
    return out


def _sources():
    pats = ["oracle/*.cpp", "oracle/*.py", "include/*.h", "include/*.hpp", "interpn_amd/*.py", "interpn_amd/csrc/*.h", "interpn_amd/csrc/*.hip"]
    for pat in pats:
        for f in sorted(glob.glob(os.path.join(ROOT, pat))):
            yield f


def test_every_cited_reference_line_exists():
    bad, total = [], 0
    for f in _sources():
        for ln, text in enumerate(open(f, errors="replace"), 1):
            for m in CITE.finditer(text):
                path, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
                if path.endswith(".toml") and not os.path.isfile(os.path.join(REF, path)):
                    continue
                cands = _candidates(path)
                total += 1
                if not cands:
                    bad.append((os.path.relpath(f, ROOT), ln, m.group(0), "no such file in the reference"))
                    continue
                if not any(sum(1 for _ in open(c, errors="replace")) >= max(a, b) for c in cands) or b < a:
                    bad.append((os.path.relpath(f, ROOT), ln, m.group(0), "beyond the end of the file"))
    assert total > 300, total  # the product and the oracle cite the reference a few hundred times
    assert not bad, bad[:20]


def test_oracle_fma_sites_cite_lines_that_fuse():
    """Lines of the oracle that apply the `fma` flavour (`mul_add<FMA>` / `std::fma` under FMA) and cite a range: the
    reference's text there (a few lines of slack for the `#[cfg(feature = "fma")]` attribute) mentions mul_add."""
    f = os.path.join(ROOT, "oracle", "interpn_oracle.cpp")
    checked = 0
    for ln, text in enumerate(open(f), 1):
        if "mul_add<FMA>" not in text and "std::fma(" not in text:
            continue
        for m in CITE.finditer(text.split("//", 1)[1] if "//" in text else ""):
            path, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            ok = False
            for c in _candidates(path):
                lines = open(c, errors="replace").read().split("\n")
                if "mul_add" in "\n".join(lines[max(0, a - 4):b + 3]):
                    ok = True
            assert ok, (ln, m.group(0))
            checked += 1
    assert checked >= 4, checked
