"""The build's dependency tracking: after __graft_entry__.build() nothing is out of date, and editing any header — also the host
decoders that an earlier hand-written header list forgot (host/par_gunzip.hpp, host/fast_inflate.hpp) — makes the objects that
include it stale.  `make -q` only asks; nothing is rebuilt here and the header's timestamp is put back."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "colorid_amd", "csrc")


def up_to_date():
    return subprocess.run(["make", "-q", "-C", CSRC, "all", "tune"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()


def test_nothing_to_do_after_build(built):
    assert up_to_date()


@pytest.mark.parametrize("header", ["host/par_gunzip.hpp", "host/fast_inflate.hpp", "host/colorid_host.hpp", "cid_gather.hpp",
                                    "cid_partition.hpp", "../../include/colorid_hip.h"])
def test_touching_a_header_makes_the_build_stale(built, header):
    path = os.path.join(CSRC, header)
    st = os.stat(path)
    try:
        os.utime(path, None)
        assert not up_to_date(), f"{header} changed and make saw nothing to do"
    finally:
        os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert up_to_date()


def test_cli_binary_holds_the_current_decoders(built):
    """the round-3 symptom of the stale binary: the BMI2 build of the marker decoder (par_gunzip.hpp) missing from bin/colorid"""
    out = subprocess.run(["nm", "-C", os.path.join(ROOT, "colorid_amd", "bin", "colorid")], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "MarkerInflate::decode_bmi2" in out
