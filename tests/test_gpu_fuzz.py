"""Randomised differential tests (FUZZ_SEED0 / FUZZ_N in the environment run a longer campaign): random index shapes (k, hashes, colours, Bloom size, minimizers) and random, awkward reads
(lengths around k, empty mates, N runs, lower case, repeats, a few reads long enough for the sort-based path) through
cid_readid_count and cid_search_count / cid_search_perfect, each compared bit for bit with the oracle."""
import os

import numpy as np
import pytest

from test_gpu_readid import check
from util import to_hip_index

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", np.uint8)


def random_read(rng, genome, k):
    kind = rng.integers(0, 12)
    if kind == 0:
        L = int(rng.choice([0, 1, k - 1, k, k + 1, 2 * k - 1]))
    elif kind == 1:
        L = int(rng.integers(2_000, 9_000))           # beyond two waves' worth of LDS for many shapes: sort-based path
    else:
        L = int(rng.integers(k, 400))
    L = min(L, len(genome) - 1)
    st = int(rng.integers(0, len(genome) - L))
    a = np.frombuffer(genome[st:st + L], np.uint8).copy()
    if kind == 2 and L:
        a[rng.integers(0, L, max(1, L // 20))] = ord("N")
    if kind == 3 and L > 10:
        s0 = int(rng.integers(0, L - 5))
        a[s0:s0 + int(rng.integers(1, 40))] = ord("N")
    if kind == 4:
        a = np.frombuffer(a.tobytes().lower(), np.uint8).copy()
    if kind == 5 and L:
        low = rng.random(L) < 0.3
        a[low] |= 0x20
    if kind == 6 and L:
        a = np.tile(a[:max(1, min(L, int(rng.integers(1, 12))))], 40)[:max(L, k + 5)]
    if kind == 7 and L:
        a[rng.integers(0, L, 3)] = np.frombuffer(b"RY-", np.uint8)
    return a.tobytes()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEED0", 0)), int(os.environ.get("FUZZ_SEED0", 0)) + int(os.environ.get("FUZZ_N", 24))))
def test_readid_random_shapes(orc, hip_ctx, seed):
    import colorid_amd
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([5, 11, 16, 21, 27, 31, 32, 33, 40, 64]))
    n_hash = int(rng.integers(1, 7))
    C = int(rng.choice([1, 2, 31, 33, 64, 65, 128, 200, 513, 1000]))
    m = int(rng.choice([1 << 12, 4099, 30011, 1 << 15]))
    msz = int(rng.integers(max(3, k // 3), k + 1)) if seed % 3 == 2 else 0
    genomes = [ACGT[rng.integers(0, 4, 12_000)].tobytes() for _ in range(3)]
    oix = orc.Index(m, n_hash, k, C)
    if msz:
        oix.set_minimizer(msz)
    for c in range(C):
        oix.set_color(c, f"a{c}", 100)
    for gi, g in enumerate(genomes):
        km = orc.Kmers(k)
        km.kmerize_vector(g[:int(rng.integers(3000, 12_000))], 1)
        for key in km.keys()[::int(rng.integers(1, 4))]:
            oix.insert(int(rng.integers(0, C)) if gi else gi % C, key.tobytes())
    hx = colorid_amd.Index(hip_ctx, m, n_hash, k, C)
    if msz:
        hx.set_minimizer(msz)
    hx.put_dense(oix.rows())
    hx.finalize()
    for _ in range(2):
        d, S = int(rng.integers(1, 13)), int(rng.integers(0, 7))
        reads = []
        for _ in range(int(rng.integers(1, 160))):
            g = genomes[rng.integers(0, 3)]
            n_mates = int(rng.choice([1, 1, 2, 2, 3]))
            reads.append([random_read(rng, g, k) for _ in range(n_mates)])
        check(oix, hx, reads, d, S)
    hx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEED0", 0)), int(os.environ.get("FUZZ_SEED0", 0)) + int(os.environ.get("FUZZ_N", 24)) // 2))
def test_search_random_shapes(orc, hip_ctx, seed):
    rng = np.random.default_rng(2000 + seed)
    k = int(rng.choice([1, 7, 16, 17, 31, 32, 33, 48, 97, 128]))
    n_hash = int(rng.integers(1, 9))
    C = int(rng.choice([1, 5, 32, 63, 64, 100, 256, 300, 1025, 4097]))
    m = int(rng.choice([257, 1 << 10, 10007, 1 << 16]))
    oix = orc.Index(m, n_hash, k, C)
    for c in range(C):
        oix.set_color(c, f"a{c}", 100)
    alphabet = np.frombuffer(b"ACGTacgtN", np.uint8) if seed % 2 else ACGT
    K = int(rng.integers(1, 3000))
    kmers = alphabet[rng.integers(0, len(alphabet), (K, k))]
    planted = np.flatnonzero(rng.random(K) < 0.7)
    for j in planted:
        for c in rng.choice(C, size=min(C, int(rng.integers(1, 4))), replace=False):
            oix.insert(int(c), kmers[j].tobytes())
    freq = rng.integers(1, 1000, K).astype(np.uint32)
    hx = to_hip_index(hip_ctx, oix)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    got = hx.search_count(kmers, freq)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    assert want[0].sum() > 0 or len(planted) == 0
    pw, pm = oix.search_perfect(kmers)
    gw, gm = hx.search_perfect(kmers)
    assert pm == gm and np.array_equal(pw, gw)
    sub = np.flatnonzero(want[3] != 0xFFFFFFFF)[:50]      # k-mers with exactly one colour: their AND is that colour, never missing
    if len(sub):
        pw, pm = oix.search_perfect(kmers[sub[:1]])
        gw, gm = hx.search_perfect(kmers[sub[:1]])
        assert pm == gm and not gm and np.array_equal(pw, gw)
    hx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEED0", 0)), int(os.environ.get("FUZZ_SEED0", 0)) + int(os.environ.get("FUZZ_N", 24)) // 2))
def test_kmerset_random_inputs(orc, hip_ctx, seed):
    """GPU k-mer counting (cid_kmerset) on random sequence sets: FASTA mode (mixed case allowed, upper-cased keys) in several
    add_seqs calls against kmerize_vector, with clean_map, the multiplicity histogram and the device-side report (hits / unique /
    sum / mode) of a search over the set; k = 1..32 as 2-bit codes, 33..128 as byte strings."""
    import colorid_amd
    rng = np.random.default_rng(3000 + seed)
    k = int(rng.integers(1, 33)) if seed % 3 else int(rng.integers(33, 129))     # every third case: byte-string keys (k > 32)
    alphabets = [b"ACGT", b"ACGTN", b"ACGTacgt", b"ACGTacgtNn", b"AC", b"ACGTRY"]
    seqs = []
    for _ in range(int(rng.integers(1, 12))):
        L = int(rng.choice([0, 1, k - 1, k, k + 1, 50, 300, 2047 + k, 2048 + k, 2049 + k, 5000]))
        a = alphabets[rng.integers(0, len(alphabets))]
        s = np.frombuffer(a, np.uint8)[rng.integers(0, len(a), max(L, 0))].tobytes()
        if rng.random() < 0.3 and len(s) > 10:
            s = s[:len(s) // 2] * 3
        seqs.append(s)
    want = orc.Kmers(k)
    for s in seqs:
        want.kmerize_vector(s, 1)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    cut = int(rng.integers(0, len(seqs) + 1))
    ks.add_seqs(seqs[:cut], 0)
    ks.add_seqs(seqs[cut:], 0)
    assert ks.finalize() == len(want)
    assert ks.as_dict() == want.as_dict()
    if len(want):
        vals, cnts = ks.histogram()
        wc = want.counts()
        assert dict(zip(vals.tolist(), cnts.tolist())) == {int(v): int((wc == v).sum()) for v in np.unique(wc)}
        if seed % 2 == 0 and len(want) > 3:     # search the set + the whole report on the device
            from util import random_index, to_hip_index
            C = int(rng.choice([3, 64, 70, 300]))
            oix = random_index(orc, rng, 4001, int(rng.integers(1, 5)), k, C, density=0.05, zero_row_frac=0.1)
            keys = want.keys()
            for j in rng.choice(len(keys), size=min(len(keys), 200), replace=False):
                oix.insert(int(rng.integers(0, C)), keys[j].tobytes())
            hx = to_hip_index(hip_ctx, oix)
            km, cnt = ks.download()
            w = oix.search_count(km, cnt.astype(np.uint64))
            modes = orc.unique_modes(w[3], cnt.astype(np.uint64), C)
            hits, nu, sf, md = ks.search_count_report(hx)
            assert np.array_equal(hits, w[0]) and np.array_equal(nu, w[1]) and np.array_equal(sf, w[2]) and np.array_equal(md, modes)
            hx.close()
        t = int(rng.integers(0, 4))
        ks.clean(t)
        assert ks.as_dict() == want.clean_map(t).as_dict()
    ks.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEED0", 0)), int(os.environ.get("FUZZ_SEED0", 0)) + int(os.environ.get("FUZZ_N", 24)) // 2))
def test_kmerset_for_an_index_random_inputs(orc, hip_ctx, seed, monkeypatch, tune):
    """A k-mer set built FOR an index (cid_kmerset_set_target_index) on random sequence sets and index sizes — one row, a handful of rows
    (every window in a few runs: the per-run fallbacks), powers of two, primes — through the partition kernels (CID_KMERSET_MSD_MIN=1) or
    the two LSD sorts, in one batch or merged from several: same contents as the oracle's map, (first-row key, code) order, the same
    report as the code-ordered set."""
    import colorid_amd
    from util import random_index, to_hip_index
    tune("CID_KMERSET_TARGET_SMALL", 1)   # (by default a set for an index below 2^20 rows keeps code order)
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(1, 33))
    m = int(rng.choice([1, 2, 7, 64, 4001, 65_536, 1_000_003, (1 << 20) + 3, 50_000_017]))
    if seed % 2:
        tune("CID_KMERSET_MSD_MIN", 1)
    if seed % 3 == 0:
        tune("CID_KMERSET_COMPACT_WINDOWS", int(rng.choice([500, 3000, 20000])))
    alphabets = [b"ACGT", b"ACGTN", b"ACGTacgt", b"AC", b"ACGTRY"]
    seqs = []
    for _ in range(int(rng.integers(1, 10))):
        L = int(rng.choice([0, 1, k - 1, k, k + 1, 50, 300, 2047 + k, 2049 + k, 5000, 40_000]))
        a = alphabets[rng.integers(0, len(alphabets))]
        s = np.frombuffer(a, np.uint8)[rng.integers(0, len(a), max(L, 0))].tobytes()
        if rng.random() < 0.3 and len(s) > 10:
            s = s[:len(s) // 2] * 3
        seqs.append(s)
    want = orc.Kmers(k)
    for s in seqs:
        want.kmerize_vector(s, 1)
    C = int(rng.choice([3, 64, 70, 300]))
    n_hash = int(rng.integers(1, 5))
    if m <= 1_000_003:
        oix = random_index(orc, rng, m, n_hash, k, C, density=0.05, zero_row_frac=0.1)
        keys = want.keys()
        for j in rng.choice(len(keys), size=min(len(keys), 200), replace=False) if len(keys) else []:
            oix.insert(int(rng.integers(0, C)), keys[j].tobytes())
        hx = to_hip_index(hip_ctx, oix)
    else:
        oix = None
        hx = colorid_amd.Index(hip_ctx, m, n_hash, k, C)
        hx.finalize()
    sets = []
    for targeted in (True, False):
        ks = colorid_amd.KmerSet(hip_ctx, k)
        if targeted:
            ks.set_target_index(hx)
        cut = int(rng.integers(0, len(seqs) + 1))
        ks.add_seqs(seqs[:cut], 0)
        ks.add_seqs(seqs[cut:], 0)
        assert ks.finalize() == len(want)
        sets.append(ks)
    tk, pk = sets
    km, cnt = tk.download()
    assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
    scale = 0xFFFFFFFF00000000 // m
    order = [(((orc.xxh3(bytes(r), 0) % m) * scale) >> 32, bytes(r)) for r in km[:3000]]
    assert order == sorted(order)
    if len(want):
        rt, rp = tk.search_count_report(hx), pk.search_count_report(hx)
        for a, b in zip(rt, rp):
            assert np.array_equal(a, b)
        if oix is not None:
            w = oix.search_count(km, cnt.astype(np.uint64))
            g = tk.search_count(hx)
            for a, b in zip(w, g):
                assert np.array_equal(a, b)
        t = int(rng.integers(0, 4))
        tk.clean(t)
        assert tk.as_dict() == want.clean_map(t).as_dict()
    tk.close(); pk.close(); hx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUZZ_SEED0", 0)), int(os.environ.get("FUZZ_SEED0", 0)) + int(os.environ.get("FUZZ_N", 24)) // 2))
def test_group_random_shapes(orc, seed):
    """The same random shapes through the multi-rank calls (1-4 ranks sharing the one GPU): a replicated index with the query
    sharded (cid_group_search_*), the k-mer set counted over the ranks (cid_group_kmerset + _search_*_parts), and the index cut into
    colour stripes (cid_group_stripes_*: search, perfect search, read_id) — every result against the oracle on the whole index."""
    import colorid_amd
    from test_gpu_group_stripes import _striped
    from test_gpu_readid import pack_reads
    rng = np.random.default_rng(4000 + seed)
    n_ranks = int(rng.integers(1, 5))
    k = int(rng.choice([7, 16, 21, 31, 32]))
    n_hash = int(rng.integers(1, 6))
    C = int(rng.choice([64 * n_ranks, 64 * n_ranks + 1, 300, 513, 1000, 8300])) if n_ranks > 1 else int(rng.choice([1, 63, 300]))
    C = max(C, 64 * (n_ranks - 1) + 1)                       # at least one 64-colour word per rank
    m = int(rng.choice([1031, 4099, 30011]))
    oix = orc.Index(m, n_hash, k, C)
    for c in range(C):
        oix.set_color(c, f"a{c}", 100)
    genomes = [ACGT[rng.integers(0, 4, 3000)].tobytes() for _ in range(4)]
    for gi, gen in enumerate(genomes):
        km = orc.Kmers(k)
        km.kmerize_vector(gen, 1)
        for key in km.keys():
            for c in {gi % C, (gi * 97 + 64) % C, C - 1 - gi % C}:
                oix.insert(int(c), key.tobytes())
    seqs = [genomes[int(rng.integers(0, 4))][int(s):int(s) + int(rng.integers(k, 400))] for s in rng.integers(0, 2500, 40)]
    seqs += [ACGT[rng.integers(0, 4, 200)].tobytes() for _ in range(5)]          # k-mers the index has never seen
    want_set = orc.Kmers(k)
    for s in seqs:
        want_set.kmerize_vector(s, 1)
    keys, cnts = want_set.keys(), want_set.counts()
    order = np.lexsort(keys.T[::-1])                                              # ascending, the device set's order
    keys, cnts = keys[order], cnts[order]
    want = oix.search_count(keys, cnts)
    wp = oix.search_perfect(keys)
    devices = [0] * n_ranks
    # replicated index: host k-mers sharded; the set counted over the ranks and searched in parts
    g = colorid_amd.Group(devices)
    hx = colorid_amd.Index(g.ctxs[0], m, n_hash, k, C)
    hx.put_dense(oix.rows()); hx.finalize(); g.replicate(hx)
    got = g.search_count(keys, cnts.astype(np.uint32))
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    gs = g.kmerset(k)
    gs.add_seqs(seqs[:17], 0); gs.add_seqs(seqs[17:], 0)
    assert gs.finalize() == len(keys)
    km, c2 = gs.download()
    assert np.array_equal(km, keys) and np.array_equal(c2, cnts.astype(np.uint32))
    got = gs.search_count()
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    hits, nu, sf, md = gs.search_count_report()
    assert np.array_equal(hits, want[0]) and np.array_equal(nu, want[1]) and np.array_equal(sf, want[2])
    assert np.array_equal(md, orc.unique_modes(want[3], cnts, C))
    gw, gm = gs.search_perfect()
    assert gm == wp[1] and np.array_equal(gw, wp[0])
    g.close()
    # colour stripes
    g = colorid_amd.Group(devices)
    st = _striped(g, oix, via_records=bool(seed % 2))
    got = st.search_count(keys, cnts.astype(np.uint32))
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    gw, gm = st.search_perfect(keys)
    assert gm == wp[1] and np.array_equal(gw, wp[0])
    one = keys[np.flatnonzero(want[0].sum() >= 0)[:1]] if len(keys) else keys
    if len(one):
        pw, pm = oix.search_perfect(one)
        gw, gm = st.search_perfect(one)
        assert gm == pm and np.array_equal(gw, pw)
    if True:   # stripes of any width, reads of any length (cid_readid_stripe_zero / _count route per stripe)
        reads = [[s] if i % 3 else [s, seqs[(i + 1) % len(seqs)][:150]] for i, s in enumerate(seqs[:30])]
        reads = [[x[:300] for x in r] for r in reads]
        if seed % 2:   # reads too long for a wave's LDS (sort-based path), one of them with lower-case bases, mixed with the short ones
            reads += [[genomes[0]], [genomes[1] + genomes[2][:1500]], [genomes[3][:2900].lower(), genomes[0][:120]], [genomes[2][:40] * 80]]
        bases, seq_off, read_seq0 = pack_reads(reads)
        d, S = int(rng.choice([1, 3])), int(rng.choice([0, 3]))
        w = oix.readid_counts(bases, seq_off, read_seq0, d, S)
        rs, col, cnt, nk, stt = st.readid_count_sparse(bases, seq_off, read_seq0, d, S)
        rows, cols = np.nonzero(w[0])
        assert np.array_equal(nk, w[1]) and np.array_equal(stt, w[2])
        assert np.array_equal(col, cols.astype(np.uint32)) and np.array_equal(cnt, w[0][rows, cols])
    g.close()
