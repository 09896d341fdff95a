import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test timeout (pytest-timeout)")
    # The built artefacts are git-ignored: a fresh checkout has none.  Build them in-tree once (hipcc cross-compiles
    # without a GPU; make is a no-op when everything is up to date).
    lib = os.path.join(ROOT, "colorid_amd", "libcolorid_hip.so")
    cli = os.path.join(ROOT, "colorid_amd", "bin", "colorid")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def hip_ctx():
    import colorid_amd
    ctx = colorid_amd.Context(0)
    yield ctx
    ctx.close()


def _switch_table():
    """cid_switches.def: {ENV_NAME: (tune name, kind, default)} — the library's switches, one list for the code, the README and the tests"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for m in re.finditer(r'^CID_SWITCH\((\w+),\s*"(\w+)",\s*(\w),\s*([-\w]+),', open(os.path.join(root, "colorid_amd", "csrc", "cid_switches.def")).read(), re.M):
        out[m.group(2)] = (m.group(1), m.group(3), m.group(4))
    return out


@pytest.fixture
def tune(hip_ctx):
    """tune("CID_KMERSET_MSD_MIN", 1) or tune("kmerset_msd_min", 1): a switch of the session's context for one test (cid_ctx_tune), put
    back to its default afterwards.  (Until round 5 the tests set environment variables the library read at every call; the library now
    reads its environment once, when a context is made.)"""
    table = _switch_table()
    by_name = {v[0]: v for v in table.values()}
    touched = []

    def set_(name, value):
        nm, kind, dflt = table[name] if name in table else by_name[name]
        hip_ctx.tune(nm, int(value))
        touched.append((nm, int(dflt)))
    yield set_
    for nm, dflt in touched:
        hip_ctx.tune(nm, dflt)
