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
