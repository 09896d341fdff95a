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


# ---------------------------------------------------------------------------------------------- the v0.7 draft (candidate variant)

_M64 = (1 << 64) - 1
_SECRET16 = bytes.fromhex("b8fe6c3923a44bbe7c01812cf721ad1cded46de9839097db7240a4a4b7b3671f")   # first 32 bytes of the default secret


def _py_xxh3_17to32(b, seed, avalanche_mult):
    """Third, independent restatement (Python ints) of XXH3's 17..32-byte path; the avalanche multiplier is the parameter."""
    def rd(x, o):
        return int.from_bytes(x[o:o + 8], "little")

    def mix(p, o, key_o):
        lo = rd(p, o) ^ ((rd(_SECRET16, key_o) + seed) & _M64)
        hi = rd(p, o + 8) ^ ((rd(_SECRET16, key_o + 8) - seed) & _M64)
        prod = lo * hi
        return (prod & _M64) ^ (prod >> 64)
    acc = (len(b) * 0x9E3779B185EBCA87 + mix(b, 0, 0) + mix(b, len(b) - 16, 16)) & _M64
    acc ^= acc >> 37
    acc = (acc * avalanche_mult) & _M64
    return acc ^ (acc >> 32)


def test_v07_draft_variant_device_matches_oracle_and_structure(shim, orc):
    """CID_HASH_XXH3_V07 (candidate for crate xxh3 0.1.x, unverified against it): the device header and the oracle's separately
    written restatement agree for every length 1..128 at every image alignment; and for the k-mer lengths that matter (17..32) a
    third restatement in Python, validated with the v0.8 multiplier against python-xxhash, gives the same values with PRIME64_3."""
    import xxhash
    rnd = random.Random(12)
    shim.shim_set_hash_variant.argtypes = [C.c_uint32]
    shim.shim_set_hash_variant(1)
    try:
        for ln in list(range(1, 129)):
            for trial in range(6):
                b = bytes(rnd.getrandbits(8) for _ in range(ln)) if trial else (b"ACGT" * 32)[:ln]
                for off in (0, 1, 3, 6):
                    out = (C.c_uint64 * 6)()
                    shim.shim_hash_seeds(b, ln, off, 6, out)
                    for s in range(6):
                        assert out[s] == orc.xxh3_v07(b, s), (ln, off, s)
                if 17 <= ln <= 32:
                    for s in (0, 1, 3):
                        assert _py_xxh3_17to32(b, s, 0x165667919E3779F9) == xxhash.xxh3_64_intdigest(b, seed=s)   # the structure is right
                        assert _py_xxh3_17to32(b, s, 0x165667B19E3779F9) == orc.xxh3_v07(b, s)                   # the draft = same, other multiplier
                        assert orc.xxh3_v07(b, s) != orc.xxh3(b, s)
    finally:
        shim.shim_set_hash_variant(0)
    # the oracle's switch routes every index hash through the chosen variant
    with orc.hash_variant(1):
        assert orc.xxh3(b"ACGTACGTACGTACGTACGTA", 2) == orc.xxh3_v07(b"ACGTACGTACGTACGTACGTA", 2)
    assert orc.xxh3(b"ACGTACGTACGTACGTACGTA", 2) == xxhash.xxh3_64_intdigest(b"ACGTACGTACGTACGTACGTA", seed=2)
