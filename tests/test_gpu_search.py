"""Parity of the HIP path (through the C ABI) with the oracle: bit-exact integers on the same seeded inputs.
Covers a5 (cid_search_count) and a4 (cid_search_perfect) over row widths, hash counts, k-mer lengths,
ragged tiles, absent rows and empty inputs; plus put_rows/get_rows round trips."""
import os

import numpy as np
import pytest

from util import plant, random_index, random_kmers, to_hip_index

pytestmark = pytest.mark.gpu

# (n_colors, n_hash, k, bloom_size)
LAYOUTS = [
    (4, 4, 27, 750_000),        # test.sh parameters, rs=1 (8-byte rows)
    (1, 1, 31, 4099),
    (46, 4, 31, 100_003),       # ref_file_example.txt colour count
    (64, 3, 21, 1 << 16),       # power-of-two bloom size (mask path)
    (65, 2, 21, 50_021),        # rs=2 with one live bit in word 1
    (128, 4, 31, 40_009),
    (129, 4, 31, 40_009),       # w64=3 -> rs=4, dead lanes past the row's width
    (255, 2, 21, 30_011),
    (256, 4, 31, 1 << 20),      # the headline row width: 32-byte rows, 2 lanes per row
    (300, 5, 31, 20_011),       # generic n_hash > 4 path
    (512, 4, 31, 20_011),
    (1024, 4, 31, 10_007),      # 128-byte rows, 8 lanes per row
    (1500, 3, 31, 5_003),
    (4096, 3, 31, 3_001),       # 512-byte rows
    (8192, 2, 31, 1_009),       # 1 KiB rows: the whole wave covers one row
    (8193, 2, 31, 1_009),       # "wide" rows (> 8192 colours): 2 KiB stride, one wave walks a row in 1-KiB steps
    (20_000, 3, 21, 503),       # 3 KiB rows, GTDB-scale colour count
]


@pytest.mark.parametrize("n_colors,n_hash,k,m", LAYOUTS)
def test_search_count_parity(orc, hip_ctx, n_colors, n_hash, k, m):
    rng = np.random.default_rng(n_colors * 31 + n_hash)
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.25, zero_row_frac=0.15)
    K = 3000 + n_colors % 61
    kmers = random_kmers(rng, K, k)
    plant(oix, rng, kmers, frac=0.5, max_colours=min(3, n_colors))
    freq = rng.integers(1, 1000, size=K).astype(np.uint32)
    hx = to_hip_index(hip_ctx, oix)
    want = oix.search_count(kmers, freq)
    got = hx.search_count(kmers, freq)
    for w, g, name in zip(want, got, ("hits", "n_unique", "sum_unique_freq", "unique_colour")):
        assert np.array_equal(w, g), name
    assert want[0].sum() > 0
    # outputs are optional: hits only
    hits_only = hx.search_count(kmers, None, want_unique=False, want_unique_colour=False)[0]
    assert np.array_equal(hits_only, want[0])
    # freq == NULL counts every unique hit once
    w2 = oix.search_count(kmers, None)
    g2 = hx.search_count(kmers, None)
    assert np.array_equal(w2[2], g2[2]) and np.array_equal(w2[1], g2[2])
    hx.close()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 7, 8, 9, 12, 16, 17, 21, 27, 31, 32, 33, 48, 64, 65, 96, 97, 127, 128])
def test_kmer_lengths(orc, hip_ctx, k):
    rng = np.random.default_rng(1000 + k)
    alphabet = np.frombuffer(b"ACGTacgtN", np.uint8)        # raw bytes are hashed as they are (App. B Q2)
    oix = random_index(orc, rng, 20_011, 3, k, 100, density=0.4, zero_row_frac=0.0)
    kmers = random_kmers(rng, 777, k, alphabet)
    hx = to_hip_index(hip_ctx, oix)
    want = oix.search_count(kmers, None)
    got = hx.search_count(kmers, None)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    hx.close()


@pytest.mark.parametrize("n_hash", [1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 32])
def test_hash_counts(orc, hip_ctx, n_hash):
    rng = np.random.default_rng(2000 + n_hash)
    oix = random_index(orc, rng, 9_973, n_hash, 31, 256, density=0.8, zero_row_frac=0.01)
    kmers = random_kmers(rng, 1500, 31)
    plant(oix, rng, kmers, frac=0.7)
    hx = to_hip_index(hip_ctx, oix)
    for w, g in zip(oix.search_count(kmers, None), hx.search_count(kmers, None)):
        assert np.array_equal(w, g)
    pw, pm = oix.search_perfect(kmers[:5])
    gw, gm = hx.search_perfect(kmers[:5])
    assert pm == gm and np.array_equal(pw, gw)
    hx.close()


@pytest.mark.parametrize("K", [0, 1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 4097])
def test_ragged_and_empty_batches(orc, hip_ctx, K):
    rng = np.random.default_rng(3000 + K)
    oix = random_index(orc, rng, 30_011, 4, 31, 256, density=0.5, zero_row_frac=0.05)
    kmers = random_kmers(rng, K, 31)
    plant(oix, rng, kmers, frac=0.9)
    freq = rng.integers(1, 5, size=K).astype(np.uint32)
    hx = to_hip_index(hip_ctx, oix)
    for w, g in zip(oix.search_count(kmers, freq), hx.search_count(kmers, freq)):
        assert np.array_equal(w, g)
    hx.close()


def test_duplicates_and_dense_hits(orc, hip_ctx):
    # every k-mer in every colour: AND words are all-ones (dense counting path), duplicates are counted per entry
    rng = np.random.default_rng(77)
    oix = orc.Index(5_003, 3, 21, 200)
    oix.rows()[:] = 0xFFFFFFFF
    oix.rows()[:, -1] = (1 << (200 - 192)) - 1
    kmers = np.repeat(random_kmers(rng, 50, 21), 7, axis=0)
    hx = to_hip_index(hip_ctx, oix)
    want, got = oix.search_count(kmers, None), hx.search_count(kmers, None)
    assert want[0].tolist() == [350] * 200
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    hx.close()


@pytest.mark.parametrize("n_colors,n_hash,k,m", LAYOUTS)
def test_search_perfect_parity(orc, hip_ctx, n_colors, n_hash, k, m):
    rng = np.random.default_rng(n_colors * 17 + n_hash + 5)
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.2, zero_row_frac=0.1)
    kmers = random_kmers(rng, 700, k)
    # colours 0 and C-1 hold every k-mer; colour C//2 holds all but one
    for j, km in enumerate(kmers):
        oix.insert(0, km.tobytes())
        oix.insert(n_colors - 1, km.tobytes())
        if j != 333:
            oix.insert(n_colors // 2, km.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    ww, wm = oix.search_perfect(kmers)
    gw, gm = hx.search_perfect(kmers)
    assert not wm and not gm
    assert np.array_equal(ww, gw)
    assert ww[0] & 1 and ww[(n_colors - 1) // 32] >> ((n_colors - 1) % 32) & 1
    # a k-mer with an absent row => "No perfect hits!" (perfect_search.rs:38-39)
    rows = oix.rows()
    absent = int(orc.xxh3(kmers[10].tobytes(), n_hash - 1) % m)
    saved = rows[absent].copy()
    rows[absent] = 0
    hx2 = to_hip_index(hip_ctx, oix)
    ww, wm = oix.search_perfect(kmers)
    gw, gm = hx2.search_perfect(kmers)
    assert wm and gm and not gw.any() and not ww.any()
    rows[absent] = saved
    for K in (1, 63, 65):
        a, b = oix.search_perfect(kmers[:K]), hx.search_perfect(kmers[:K])
        assert a[1] == b[1] and np.array_equal(a[0], b[0])
    hx.close(); hx2.close()


def test_put_get_rows_roundtrip(orc, hip_ctx):
    import colorid_amd
    rng = np.random.default_rng(11)
    for n_colors in (4, 33, 64, 65, 256, 257, 1000):
        m = 10_007
        hx = colorid_amd.Index(hip_ctx, m, 2, 21, n_colors)
        w32 = (n_colors + 31) // 32
        ids = rng.choice(m, size=500, replace=False).astype(np.uint64)
        words = rng.integers(0, 2**32, size=(500, w32), dtype=np.uint64).astype(np.uint32)
        if n_colors % 32:
            words[:, -1] &= np.uint32((1 << (n_colors % 32)) - 1)
        hx.put_rows(ids, words)
        hx.finalize()
        assert np.array_equal(hx.get_rows(ids), words)
        others = np.setdiff1d(np.arange(m, dtype=np.uint64), ids)[:300]
        assert not hx.get_rows(others).any()                 # rows never put stay absent (all-zero)
        hx.close()


def test_error_behaviour(hip_ctx):
    import colorid_amd
    with pytest.raises(colorid_amd.CidError):
        colorid_amd.Index(hip_ctx, 1000, 2, 129, 8)          # k_size > 128
    with pytest.raises(colorid_amd.CidError):
        colorid_amd.Index(hip_ctx, 1000, 2, 21, 8, hash_variant=7)
    with pytest.raises(colorid_amd.CidError):
        colorid_amd.Index(hip_ctx, 1000, 2, 21, (1 << 20) + 1)  # colour limit
    hx = colorid_amd.Index(hip_ctx, 1000, 2, 21, 8)
    with pytest.raises(colorid_amd.CidError):                # not finalized
        hx.search_count(np.zeros((1, 21), np.uint8))
    with pytest.raises(colorid_amd.CidError):                # row id out of range
        hx.put_rows(np.array([1000], np.uint64), np.array([[1]], np.uint32))
    with pytest.raises(colorid_amd.CidError):                # bits beyond n_colors
        hx.put_rows(np.array([1], np.uint64), np.array([[256]], np.uint32))
    hx.close()


def test_insert_kmers_matches_oracle_bloom(orc, hip_ctx):
    import torch

    import colorid_amd
    rng = np.random.default_rng(21)
    for n_colors, k, m, n_hash in ((4, 27, 75_011, 4), (256, 31, 1 << 16, 4), (1000, 21, 9_001, 2)):
        kmers = random_kmers(rng, 5000, k)
        cols = rng.integers(0, n_colors, size=5000).astype(np.uint32)
        oix = orc.Index(m, n_hash, k, n_colors)
        for km, c in zip(kmers, cols):
            oix.insert(int(c), km.tobytes())
        hx = colorid_amd.Index(hip_ctx, m, n_hash, k, n_colors)
        dk = torch.from_numpy(kmers.reshape(-1)).cuda()
        dc = torch.from_numpy(cols.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        hx.insert_kmers_dev(dk.data_ptr(), dc.data_ptr(), 5000)
        hx.finalize()
        got = hx.get_rows(np.arange(m, dtype=np.uint64))
        assert np.array_equal(got, oix.rows())
        hx.close()


def test_put_records_parses_bxi_rows_on_the_device(orc, hip_ctx, tmp_path):
    """cid_index_put_records takes the .bxi file's own row records; the matrix equals the one built row by row, and malformed
    records are refused (bigsi.rs:59-63: the reference's deserialiser would panic)."""
    import struct

    import colorid_amd
    rng = np.random.default_rng(8)
    for n_colors in (5, 32, 33, 100, 300):
        oix = random_index(orc, rng, 20_011, 3, 21, n_colors, density=0.3, zero_row_frac=0.4)
        path = str(tmp_path / f"c{n_colors}.bxi")
        oix.save(path)
        raw = open(path, "rb").read()
        off = 32
        for _ in range(n_colors):
            (ln,) = struct.unpack_from("<Q", raw, off + 8)
            off += 16 + ln
        (n_rows,) = struct.unpack_from("<Q", raw, off)
        off += 8
        rec = 24 + 4 * oix.w32
        records = raw[off:off + n_rows * rec]
        hx = colorid_amd.Index(hip_ctx, oix.m, oix.n_hash, oix.k, n_colors)
        cut = (n_rows // 3) * rec
        hx.put_records(records[:cut])            # in two calls, any order
        hx.put_records(records[cut:])
        hx.finalize()
        assert np.array_equal(hx.get_rows(np.arange(oix.m, dtype=np.uint64)), oix.rows())
        # and back: the non-zero rows formatted on the device are the file's bytes (cid_index_get_records), also piecewise
        assert hx.get_records(0, oix.m) == records
        mid = oix.m // 2 + 3
        assert hx.get_records(0, mid) + hx.get_records(mid, oix.m - mid) == records and hx.get_records(7, 0) == b""
        hx.close()
        # malformed: wrong word count / wrong bit count / row beyond bloom_size / a bit beyond n_colors
        bad = []
        b = bytearray(records[:rec]); struct.pack_into("<Q", b, 8, oix.w32 + 1); bad.append(bytes(b))
        b = bytearray(records[:rec]); struct.pack_into("<Q", b, 16 + 4 * oix.w32, n_colors + 1); bad.append(bytes(b))
        b = bytearray(records[:rec]); struct.pack_into("<Q", b, 0, oix.m); bad.append(bytes(b))
        if n_colors % 32:
            b = bytearray(records[:rec]); b[16 + 4 * oix.w32 - 1] |= 0x80; bad.append(bytes(b))
        for rb in bad:
            hx = colorid_amd.Index(hip_ctx, oix.m, oix.n_hash, oix.k, n_colors)
            with pytest.raises(Exception):
                hx.put_records(records[:5 * rec] + rb)
            hx.close()


def test_index_shared_by_two_contexts_on_two_threads(orc, hip_ctx):
    """A finalized index is read-only: a second context (own stream and scratch) on another host thread searches and classifies
    against it at the same time as the first, and both get the oracle's answers."""
    import threading

    import colorid_amd
    from test_gpu_readid import pack_reads
    rng = np.random.default_rng(17)
    oix = random_index(orc, rng, 60_013, 3, 27, 200, density=0.25, zero_row_frac=0.1)
    kmers = random_kmers(rng, 40_000, 27)
    plant(oix, rng, kmers[:8000], frac=1.0)
    freq = rng.integers(1, 9, len(kmers)).astype(np.uint32)
    hx = to_hip_index(hip_ctx, oix)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    reads = [[kmers[i:i + 5].tobytes()] for i in range(0, 3000, 5)]
    bases, so, r0 = pack_reads(reads)
    want_r = oix.readid_counts(bases, so, r0, 1, 3)
    ctx2 = colorid_amd.Context(0)
    errors = []

    def worker(ctx, n_iter):
        try:
            lib = ctx.lib
            for _ in range(n_iter):
                hits = np.zeros(200, np.uint64); nu = np.zeros(200, np.uint64); sf = np.zeros(200, np.uint64)
                uc = np.zeros(len(kmers), np.uint32)
                colorid_amd.hip.check(lib.cid_search_count(ctx.h, hx.h, kmers.ctypes.data, freq.ctypes.data, len(kmers), hits.ctypes.data,
                                                           nu.ctypes.data, sf.ctypes.data, uc.ctypes.data))
                assert all(np.array_equal(a, b) for a, b in zip(want, (hits, nu, sf, uc)))
                rep = np.zeros((len(reads), 201), np.uint32); nk = np.zeros(len(reads), np.uint32); st = np.zeros(len(reads), np.uint8)
                colorid_amd.hip.check(lib.cid_readid_count(ctx.h, hx.h, bases.ctypes.data, so.ctypes.data, len(so) - 1, r0.ctypes.data,
                                                           len(reads), 1, 3, rep.ctypes.data, nk.ctypes.data, st.ctypes.data))
                assert np.array_equal(rep, want_r[0]) and np.array_equal(nk, want_r[1])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(c, 15)) for c in (hip_ctx, ctx2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    ctx2.close()
    hx.close()


def test_more_error_behaviour(orc, hip_ctx):
    """Every misuse comes back as an error code + message (nothing aborts across the ABI): state errors, argument errors,
    the perfect search without k-mers (src/perfect_search.rs:22-23), inconsistent read batches."""
    import ctypes as C

    import colorid_amd
    lib = hip_ctx.lib
    rng = np.random.default_rng(2)
    oix = random_index(orc, rng, 5003, 2, 21, 40, density=0.3, zero_row_frac=0.1)
    hx = to_hip_index(hip_ctx, oix)
    with pytest.raises(colorid_amd.CidError):                 # finalized indices are immutable
        hx.put_rows(np.array([1], np.uint64), np.array([[1, 0]], np.uint32))
    with pytest.raises(colorid_amd.CidError):
        hx.put_records(b"\x00" * 32)
    with pytest.raises(colorid_amd.CidError):                 # "needs at least one k-mer"
        hx.search_perfect(np.zeros((0, 21), np.uint8))
    hits = np.zeros(40, np.uint64)
    assert lib.cid_search_count(hip_ctx.h, hx.h, None, None, 5, hits.ctypes.data, None, None, None) < 0      # null k-mers
    assert lib.cid_search_count(hip_ctx.h, hx.h, None, None, 0, hits.ctypes.data, None, None, None) == 0     # empty batch is fine
    assert lib.cid_search_count(None, hx.h, None, None, 0, hits.ctypes.data, None, None, None) < 0
    assert b"null" in lib.cid_last_error()
    # read batches: offsets must be monotonic and inside the arrays, the stride positive
    bases = np.frombuffer(b"ACGT" * 50, np.uint8)
    rep = np.zeros((1, 41), np.uint32); nk = np.zeros(1, np.uint32); st = np.zeros(1, np.uint8)

    def readid(seq_off, read0, n_seqs, n_reads, d=1):
        so = np.array(seq_off, np.uint64); r0 = np.array(read0, np.uint64)
        return lib.cid_readid_count(hip_ctx.h, hx.h, bases.ctypes.data, so.ctypes.data, n_seqs, r0.ctypes.data, n_reads, d, 3,
                                    rep.ctypes.data, nk.ctypes.data, st.ctypes.data)
    assert readid([0, 200], [0, 1], 1, 1) == 0
    assert readid([0, 200], [0, 1], 1, 1, d=0) < 0            # stride 0
    assert readid([200, 0], [0, 1], 1, 1) < 0                 # seq_off not monotonic
    assert readid([0, 200], [0, 2], 1, 1) < 0                 # read_seq0 points past n_seqs
    assert readid([0, 200], [1, 0], 1, 1) < 0                 # read_seq0 not monotonic
    with pytest.raises(colorid_amd.CidError):                 # rows outside the index
        hx.get_records(5000, 10)
    mx = colorid_amd.Index(hip_ctx, 5003, 2, 21, 40)
    with pytest.raises(colorid_amd.CidError):                 # minimizer longer than k
        mx.set_minimizer(22)
    mx.close()
    # the failed calls left the index usable
    kmers = random_kmers(rng, 100, 21)
    assert np.array_equal(hx.search_count(kmers)[0], oix.search_count(kmers, None)[0])
    hx.close()


@pytest.fixture(scope="module")
def tune_ctx():
    """a context of libcolorid_hip_tune.so: the shipped library + the two rejected schedulings of k_search_count (`make tune`)"""
    import colorid_amd
    from colorid_amd import _lib
    if not os.path.exists(_lib.TUNE_LIB_PATH):
        pytest.skip("libcolorid_hip_tune.so is not built (make -C colorid_amd/csrc tune)")
    ctx = colorid_amd.Context(0, lib=_lib.open_library(_lib.TUNE_LIB_PATH))
    yield ctx
    ctx.close()


@pytest.mark.parametrize("tunable", ["search_persist", "search_mixed"])
@pytest.mark.parametrize("n_colors,n_hash", [(256, 4), (200, 3), (129, 2), (64, 4), (1024, 4)])
def test_alternative_schedulings_are_bit_exact(orc, hip_ctx, tune_ctx, tunable, n_colors, n_hash):
    """The two measured-and-not-adopted variants of k_search_count stay bit-exact in the TUNE build (cid_ctx_tune): the persistent
    grid with one work queue per XCD, and (32-byte rows) the last row of each k-mer fetched through the scalar cache.  The shipped
    library does not contain them and says so."""
    import colorid_amd
    with pytest.raises(colorid_amd.CidError) as ei:
        hip_ctx.tune(tunable, 1)
    assert ei.value.code == -4 and "TUNE=1" in str(ei.value)
    rng = np.random.default_rng(n_colors * 7 + n_hash)
    oix = random_index(orc, rng, 60_013, n_hash, 31, n_colors, density=0.2, zero_row_frac=0.05)
    kmers = random_kmers(rng, 70_001, 31)            # > 2^16: the persistent path engages
    plant(oix, rng, kmers[:5000], frac=0.9)
    freq = rng.integers(1, 20, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    hx = to_hip_index(tune_ctx, oix)
    tune_ctx.tune(tunable, 1)
    try:
        got = hx.search_count(kmers, freq)
    finally:
        tune_ctx.tune(tunable, 0)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    got = hx.search_count(kmers, freq)               # and the TUNE build's default scheduling
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    hx.close()


@pytest.mark.parametrize("unroll", [1, 2])
@pytest.mark.parametrize("n_colors,n_hash", [(512, 3), (320, 4), (1024, 4), (700, 2), (1000, 5)])
def test_search_count_unroll_is_bit_exact(orc, hip_ctx, unroll, n_colors, n_hash):
    """Rows of 64 and 128 bytes: k_search_count with one and with two sub-passes' row loads in flight (cid_ctx_tune "search_unroll";
    2 is the default) against the oracle, ragged last tile, colours that do not fill the row (320, 700, 1000), 5 seeds."""
    rng = np.random.default_rng(n_colors * 11 + n_hash)
    oix = random_index(orc, rng, 40_009, n_hash, 31, n_colors, density=0.25, zero_row_frac=0.05)
    kmers = random_kmers(rng, 20_003, 31)
    plant(oix, rng, kmers[:4000], frac=0.9)
    freq = rng.integers(1, 20, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    hx = to_hip_index(hip_ctx, oix)
    hip_ctx.tune("search_unroll", unroll)
    try:
        got = hx.search_count(kmers, freq)
    finally:
        hip_ctx.tune("search_unroll", 2)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    hx.close()
