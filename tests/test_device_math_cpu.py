"""CPU check of the integer arithmetic the gfx950 kernels execute (same headers, compiled with g++):
XXH3-64 seeds 0..n-1 out of a byte image at arbitrary alignment, and the exact `% bloom_size`."""
import ctypes as C
import json
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "hash_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "cpu_shim", "hash_shim.cpp")])
    L = C.CDLL(so)
    L.shim_hash_seeds.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.shim_hash_canonical_code.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_char_p,
                                           C.POINTER(C.c_uint64)]
    L.shim_minimizer.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.POINTER(C.c_uint64)]
    L.shim_mod.restype = C.c_uint64
    L.shim_mod.argtypes = [C.c_uint64, C.c_uint64]
    L.shim_row_stride_words.restype = C.c_uint32
    L.shim_row_stride_words.argtypes = [C.c_uint32]
    return L


def test_device_hash_matches_published_kats(shim):
    with open(os.path.join(HERE, "golden", "xxh3_kat.json")) as f:
        vecs = json.load(f)["vectors"]
    by_input = {}
    for v in vecs:
        if v["seed"] < 8:
            by_input.setdefault(v["hex"], {})[v["seed"]] = v["h"]
    n_checked = 0
    for hx, seeds in by_input.items():
        b = bytes.fromhex(hx)
        if not 1 <= len(b) <= 128:
            continue
        for off in (0, 1, 2, 3, 5, 31, 62):
            out = (C.c_uint64 * 8)()
            shim.shim_hash_seeds(b, len(b), off, 8, out)
            for s, h in seeds.items():
                assert out[s] == h, (len(b), off, s)
                n_checked += 1
    assert n_checked > 3000


def test_device_mod_is_exact(shim):
    rnd = random.Random(9)
    # every bloom_size the library accepts: 1 .. 2^32 (cid_index_create refuses more; mod_m relies on it)
    ms = [1, 2, 3, 5, 6, 7, 12, 1000, 750000, 30_000_000, 50_000_000, 250_000_000, 2**30, 2**32 - 1, 2**32, 2**32 - 5,
          2**31 + 1, 2**31 - 1, 2**31, 10**9 + 7, 3 * 2**30] + [rnd.getrandbits(rnd.randint(2, 32)) | 1 for _ in range(80)]
    for m in ms:
        hs = [0, 1, m - 1, m, (m + 1) % 2**64, 2**64 - 1, 2**63, 2**32] + [rnd.getrandbits(64) for _ in range(500)]
        for h in hs:
            assert shim.shim_mod(h, m) == h % m, (h, m)


def test_row_stride_rule(shim):
    want = {1: 1, 4: 1, 46: 1, 64: 1, 65: 2, 128: 2, 129: 4, 255: 4, 256: 4, 257: 8, 512: 8, 1024: 16, 1025: 32,
            4096: 64, 8192: 128, 8193: 256, 16384: 256, 16385: 384, 20000: 384, 65536: 1024}
    for c, rs in want.items():
        assert shim.shim_row_stride_words(c) == rs


def test_packed_code_path_matches_string_path(shim, orc):
    """2-bit window code -> canonical choice -> ASCII re-expansion -> XXH3: equals hashing the canonical string
    the reference would build (min(window, revcomp) on bytes, ties -> revcomp), for every k <= 32."""
    import xxhash
    rnd = random.Random(4)
    for k in range(1, 33):
        for trial in range(60):
            if trial == 0:
                s = (b"ACGT" * 9)[:k]
            elif trial == 1:
                s = b"A" * k
            elif trial == 2:
                half = bytes(rnd.choice(b"ACGT") for _ in range(k // 2))
                s = (half + orc.revcomp(half))[:k] if k % 2 == 0 else bytes(rnd.choice(b"ACGT") for _ in range(k))
            else:
                s = bytes(rnd.choice(b"ACGT") for _ in range(k))
            rc = orc.revcomp(s)
            want = s if s < rc else rc
            out = (C.c_uint64 * 4)()
            canon = C.create_string_buffer(k)
            msb = C.c_uint64(0)
            shim.shim_hash_canonical_code(s, k, 4, out, canon, C.byref(msb))
            assert canon.raw == want, (k, s)
            assert msb.value == int("".join(str(b"ACGT".index(c)) for c in want), 4)
            for sd in range(4):
                assert out[sd] == xxhash.xxh3_64_intdigest(want, seed=sd), (k, s, sd)


def test_minimizer_code_matches_find_minimizer(shim, orc):
    """device minimizer of a 2-bit code == find_minimizer (kmer.rs:971-986) of the string, and its hash == XXH3 of that string"""
    import xxhash
    rnd = random.Random(8)
    for k, m in ((31, 15), (27, 15), (21, 11), (32, 16), (15, 15), (16, 1), (31, 31), (20, 9), (12, 4)):
        for trial in range(200):
            s = bytes(rnd.choice(b"ACGT") for _ in range(k)) if trial else b"A" * k
            want = orc.find_minimizer(s, m)
            out = C.create_string_buffer(m)
            hs = (C.c_uint64 * 3)()
            shim.shim_minimizer(s, k, m, 3, out, hs)
            assert out.raw == want, (k, m, s)
            for sd in range(3):
                assert hs[sd] == xxhash.xxh3_64_intdigest(want, seed=sd)
