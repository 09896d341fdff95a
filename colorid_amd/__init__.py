"""colorid_amd — MI355X (gfx950) implementation of colorid's BIGSI query hot path.

The product is the C-ABI library ``libcolorid_hip.so`` (include/colorid_hip.h) plus the C++ host
(colorid_amd/csrc/host).  This Python package is only the ctypes binding used by tests and bench.py;
it contains no compute and no CPU fallback: without the built HIP library every call raises.
"""
from ._lib import CidError, load_library  # noqa: F401
from .hip import Context, FastqReader, Group, Index, KmerSet  # noqa: F401
