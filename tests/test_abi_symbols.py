"""The C-ABI library loads on a GPU-less host and exports every symbol include/colorid_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "colorid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cid_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from colorid_amd import _lib
    lib = _lib.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/colorid_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.cid_abi_version() == 4


def test_the_library_exports_its_abi_and_nothing_else():
    """-fvisibility=hidden + csrc/export.map: the dynamic symbols the library DEFINES are exactly the header's cid_* names (round 3
    leaked 743 mangled C++ symbols), and the compressed fat binary keeps the file small (27 MB uncompressed)."""
    import subprocess
    for name in ("libcolorid_hip.so", "libcolorid_hip_tune.so"):
        path = os.path.join(ROOT, "colorid_amd", name)
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
        defined = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
        assert defined == _declared_symbols(), [d for d in defined if not d.startswith("cid_")][:10]
        assert os.path.getsize(path) < 15 * (1 << 20)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import colorid_amd
    with pytest.raises(colorid_amd.CidError) as ei:
        colorid_amd.Context(0)
    assert ei.value.code == -2  # CID_ERR_HIP: the product never computes on the CPU


def test_product_never_imports_oracle():
    """Nothing under colorid_amd/ (sources, bindings, Makefile) may mention the oracle, and bench.py may load it only in its cpu_baseline*() functions."""
    pkg = os.path.join(ROOT, "colorid_amd")
    seen = 0
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read().lower()
                seen += 1
                assert "oracle" not in txt and "liborc" not in txt and "import orc" not in txt, (dp, f)
    assert seen >= 20
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"^\s*(from oracle|import oracle|from orc|import orc)\b", bench, flags=re.M)]
    assert uses, "bench.py's cpu_baseline leg times the oracle"
    # the cpu_baseline leg = the functions named cpu_baseline* (the k-mer search's, and the read_id side record's)
    legs = []
    for mm in re.finditer(r"^def (cpu_baseline\w*)\(", bench, flags=re.M):
        nxt = re.search(r"^def |^if __name__", bench[mm.start() + 1:], flags=re.M)
        legs.append((mm.start(), mm.start() + 1 + nxt.start() if nxt else len(bench)))
    assert legs
    assert all(any(lo < u < hi for lo, hi in legs) for u in uses), "the oracle is loaded outside bench.py's cpu_baseline*() functions"
