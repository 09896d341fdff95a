"""A k-mer set built FOR an index (cid_kmerset_set_target_index): the set's own sort takes the leading bits of each window's
first-row key as its partition digits and finishes the runs on (key, code), so the set comes out ordered by (first row in the
index, code) at no extra pass.  Contents, multiplicities and every search result must be those of the code-ordered set and of the
oracle (the reference iterates a hash map: src/batch_search_pe.rs:45-84 does not depend on the order); the order itself is checked
against the key restated here from the oracle's hash."""
import numpy as np
import pytest

from util import plant, random_index, to_hip_index

pytestmark = pytest.mark.gpu


def rand_seq(rng, n, alphabet=b"ACGT"):
    return np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)].tobytes()


def row0_key(orc, kmer: bytes, m: int) -> int:
    """cid_kmerset.hip row0_key: the first hash's row, scaled to 32 bits"""
    return ((orc.xxh3(kmer, 0) % m) * (0xFFFFFFFF00000000 // m)) >> 32


def assert_target_order(orc, km, m):
    keys = [(row0_key(orc, bytes(r), m), bytes(r)) for r in km]
    assert keys == sorted(keys)      # ascending (first-row key, code); code order == ASCII order for ACGT


@pytest.fixture(autouse=True)
def _small_indices_are_targeted_too(tune):
    """the library leaves a set for an index below 2^20 rows in code order (nothing to gain, crowded runs); the tests' indices are small"""
    tune("CID_KMERSET_TARGET_SMALL", 1)


def empty_index(hip_ctx, m, n_hash, k, n_colors=8):
    import colorid_amd
    hx = colorid_amd.Index(hip_ctx, m, n_hash, k, n_colors)
    hx.finalize()
    return hx


@pytest.mark.parametrize("m", [50_021, 65_536, 4_000_037])
@pytest.mark.parametrize("k", [6, 21, 31, 32])
@pytest.mark.parametrize("flavour", ["random", "deep", "repeats", "one_kmer", "ns", "mixed", "shared_prefix", "deep_errors"])
def test_targeted_msd_path_equals_the_oracle(orc, hip_ctx, monkeypatch, k, flavour, m, tune):
    """the pair kernels of cid_partition.hpp on inputs small enough for the oracle (CID_KMERSET_MSD_MIN=1): evenly spread keys (the LDS
    bucket sort), deep coverage (crowded buckets: k_run_dedupe_sort, the radix kernel for what is not copies), one row holding most windows (runs beyond a workgroup's LDS: the
    per-run LSD sorts), windows without a k-mer (dropped by the first level)"""
    import colorid_amd
    if k == 32 and m != 50_021:
        pytest.skip("one index size is enough for the 64-bit codes")
    tune("CID_KMERSET_MSD_MIN", 1)
    rng = np.random.default_rng(k * 31 + len(flavour))
    if flavour == "random":
        seqs = [rand_seq(rng, 60_000), rand_seq(rng, 45_000)]
    elif flavour == "deep":
        g = rand_seq(rng, 2000)
        seqs = [g[s:s + 150] for s in rng.integers(0, len(g) - 150, 2000)]
    elif flavour == "repeats":
        unit = np.frombuffer((rand_seq(rng, 7) * 9000)[:60_000], np.uint8).copy()
        hit = rng.random(len(unit)) < 0.002
        unit[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
        seqs = [unit.tobytes(), b"A" * 30_000, b"AC" * 10_000]
    elif flavour == "one_kmer":
        seqs = [b"A" * (k + 20_000)]
    elif flavour == "shared_prefix":                           # thousands of DIFFERENT k-mers that agree in their first 28 bits: one crowded bucket
        head = b"A" * min(14, k - 3)                           # that is not copies (k_run_dedupe_sort gives such a run to the radix kernel)
        seqs = [head + rand_seq(rng, k - len(head)) for _ in range(3500)] + [rand_seq(rng, 20_000)]
    elif flavour == "deep_errors":                             # coverage with read errors: every true k-mer many times, its one-off variants beside
        g = np.frombuffer(rand_seq(rng, 3000), np.uint8)       # it in the same buckets (the everyday input: reads of an isolate)
        seqs = []
        for s0 in rng.integers(0, len(g) - 150, 3000):
            r = g[s0:s0 + 150].copy()
            hit = rng.random(150) < 0.01
            r[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
            seqs.append(r.tobytes())
    elif flavour == "ns":
        seqs = [rand_seq(rng, 50_000, b"ACGTN"), b"N" * 5000, rand_seq(rng, 20_000, b"ACGTNNNN")]
    else:
        g = rand_seq(rng, 3000)
        seqs = [rand_seq(rng, 40_000), b"T" * 9000, rand_seq(rng, 10_000, b"ACGTN")] + [g[s:s + 200] for s in rng.integers(0, 2800, 300)]
    hx = empty_index(hip_ctx, m, 3, k)
    for mode in (0, 1):
        want = orc.Kmers(k)
        for s in seqs:
            if mode == 0:
                want.kmerize_vector(s, 1)
            else:
                want.kmerize_fq_read(s, b"I" * len(s), 0)
        ks = colorid_amd.KmerSet(hip_ctx, k)
        ks.set_target_index(hx)
        ks.add_seqs(seqs, mode)
        assert ks.finalize() == len(want)
        km, cnt = ks.download()
        assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
        assert_target_order(orc, km, m)
        assert int(cnt.sum()) == int(want.counts().sum())
        ks.close()
    hx.close()


def test_small_sets_take_the_two_lsd_sorts(orc, hip_ctx, monkeypatch, tune):
    """below CID_KMERSET_MSD_MIN windows (the cold path, cid_kmerset_cold.hip): sort by code, then stably by key"""
    import colorid_amd
    tune("CID_KMERSET_MSD_MIN", 1000000000)
    rng = np.random.default_rng(5)
    g = rand_seq(rng, 5000)
    seqs = [g[s:s + 150] for s in rng.integers(0, len(g) - 150, 300)] + [rand_seq(rng, 700, b"ACGTN"), b"ACG"]
    hx = empty_index(hip_ctx, 30_011, 2, 21)
    want = orc.Kmers(21)
    for s in seqs:
        want.kmerize_vector(s, 1)
    ks = colorid_amd.KmerSet(hip_ctx, 21)
    ks.set_target_index(hx)
    ks.add_seqs(seqs, 0)
    assert ks.finalize() == len(want)
    km, cnt = ks.download()
    assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
    assert_target_order(orc, km, 30_011)
    ks.close(); hx.close()


@pytest.mark.parametrize("msd", [False, True])
def test_incremental_merge_keeps_the_target_order(orc, hip_ctx, monkeypatch, msd, tune):
    import colorid_amd
    tune("CID_KMERSET_COMPACT_WINDOWS", 20000)   # a merge every few batches
    tune("CID_KMERSET_MSD_MIN", 1 if msd else 1000000000)
    rng = np.random.default_rng(3)
    genome = rand_seq(rng, 30000)
    batches = [[genome[s:s + 400] for s in rng.integers(0, len(genome) - 400, 200)] for _ in range(6)]
    hx = empty_index(hip_ctx, 100_003, 4, 21)
    want = orc.Kmers(21)
    ks = colorid_amd.KmerSet(hip_ctx, 21)
    ks.set_target_index(hx)
    for b in batches:
        for s in b:
            want.kmerize_vector(s, 1)
        ks.add_seqs(b, 0)
    assert ks.finalize() == len(want)
    km, cnt = ks.download()
    assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
    assert_target_order(orc, km, 100_003)
    assert want.counts().max() > 5
    ks.clean(1)                                                   # clean_map keeps the order
    km2, cnt2 = ks.download()
    assert {bytes(km2[i]): int(cnt2[i]) for i in range(len(cnt2))} == {k: v for k, v in want.as_dict().items() if v > 1}
    assert_target_order(orc, km2, 100_003)
    ks.close(); hx.close()


@pytest.mark.parametrize("n_colors,n_hash,k", [(4, 4, 27), (256, 4, 31), (1024, 3, 21), (65, 2, 32)])
def test_search_over_a_targeted_set(orc, hip_ctx, monkeypatch, n_colors, n_hash, k, tune):
    """every output of the proportional and the perfect search over a targeted set == the oracle on the same k-mers in the set's
    order == (the order-free ones) the code-ordered set's"""
    import colorid_amd
    tune("CID_KMERSET_MSD_MIN", 1)
    rng = np.random.default_rng(n_colors + k)
    genome = rand_seq(rng, 20000)
    reads = [genome[s:s + 150] for s in rng.integers(0, len(genome) - 150, 1500)]
    want = orc.Kmers(k)
    for s in reads:
        want.kmerize_vector(s, 1)
    oix = random_index(orc, rng, 60_013, n_hash, k, n_colors, density=0.2, zero_row_frac=0.1)
    plant(oix, rng, want.keys(), frac=0.6, max_colours=min(3, n_colors))
    hx = to_hip_index(hip_ctx, oix)
    plain = colorid_amd.KmerSet(hip_ctx, k)
    plain.add_seqs(reads, 0)
    plain.finalize()
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.set_target_index(hx)
    ks.add_seqs(reads, 0)
    ks.finalize()
    km, cnt = ks.download()
    assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
    assert_target_order(orc, km, 60_013)
    w = oix.search_count(km, cnt.astype(np.uint64))
    g = ks.search_count(hx)
    for a, b in zip(w, g):
        assert np.array_equal(a, b)
    p = plain.search_count(hx)
    for a, b in zip(g[:3], p[:3]):
        assert np.array_equal(a, b)
    assert w[0].sum() > 0
    rep_t, rep_p = ks.search_count_report(hx), plain.search_count_report(hx)
    for a, b in zip(rep_t, rep_p):
        assert np.array_equal(a, b)
    pw, pm = oix.search_perfect(km[:400])
    sub = colorid_amd.KmerSet(hip_ctx, k)
    sub.set_target_index(hx)
    sub.add_seqs([genome[:300]], 0)
    sub.finalize()
    skm, _ = sub.download()
    pw, pm = oix.search_perfect(skm)
    gw, gm = sub.search_perfect(hx)
    assert pm == gm and np.array_equal(pw, gw)
    hx.close(); ks.close(); plain.close(); sub.close()


def test_default_path_at_a_million_windows(orc, hip_ctx):
    """no test switches: 1.3 M windows take the two-level partition + the LDS bucket sort as the bench's sets do"""
    import colorid_amd
    rng = np.random.default_rng(11)
    genome = rand_seq(rng, 400_000)
    reads = [genome[s:s + 150] for s in rng.integers(0, len(genome) - 150, 11_000)]
    want = orc.Kmers(31)
    for s in reads:
        want.kmerize_fq_read(s, b"I" * len(s), 0)
    hx = empty_index(hip_ctx, 50_000_000, 4, 31, n_colors=64)
    ks = colorid_amd.KmerSet(hip_ctx, 31)
    ks.set_target_index(hx)
    ks.add_seqs(reads, 1)
    assert ks.finalize() == len(want)
    km, cnt = ks.download()
    assert {bytes(km[i]): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
    sample = np.sort(rng.integers(0, len(km), 20_000))
    keys = [(row0_key(orc, bytes(km[i]), 50_000_000), bytes(km[i])) for i in sample]
    assert keys == sorted(keys)
    ks.close(); hx.close()


def test_target_comes_before_the_first_sequences(orc, hip_ctx):
    import colorid_amd
    hx = empty_index(hip_ctx, 10_007, 2, 21)
    ks = colorid_amd.KmerSet(hip_ctx, 21)
    ks.add_seqs([b"ACGT" * 20], 0)
    with pytest.raises(colorid_amd.CidError) as ei:
        ks.set_target_index(hx)
    assert ei.value.code == -5
    other = empty_index(hip_ctx, 10_007, 2, 27)
    ks2 = colorid_amd.KmerSet(hip_ctx, 21)
    with pytest.raises(colorid_amd.CidError) as ei:
        ks2.set_target_index(other)
    assert ei.value.code == -1
    ks.close(); ks2.close(); hx.close(); other.close()
