"""GPU k-mer counting (cid_kmerset) against the oracle's k-mer maps: kmerize_vector semantics
(mode 0: N filter, orientation chosen on raw bytes, upper-cased) and the fastq body (mode 1: raw case), clean_map,
the multiplicity histogram auto_cutoff needs, and searches over the device-resident set."""
import os

import numpy as np
import pytest

from util import plant, random_index, to_hip_index

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REFS = os.path.join(HERE, "golden", "refs")


def oracle_map(orc, seqs, k, mode):
    km = orc.Kmers(k)
    for s in seqs:
        if mode == 0:
            km.kmerize_vector(s, 1)
        elif len(s) >= k:
            km.kmerize_skip_n_set(s, 1)     # same window walk as the fastq body, but a set: use counts below
    return km


def fastq_counts(orc, seqs, k):
    """the fastq body (kmer.rs:481-503) without quality masking: q = 0"""
    km = orc.Kmers(k)
    for s in seqs:
        km.kmerize_fq_read(s, b"I" * len(s), 0)
    return km


def rand_seq(rng, n, alphabet=b"ACGT"):
    return np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)].tobytes()


@pytest.mark.parametrize("k", [1, 4, 15, 16, 17, 21, 27, 31, 32])
def test_fasta_mode_matches_kmerize_vector(orc, hip_ctx, k):
    import colorid_amd
    rng = np.random.default_rng(k)
    seqs = [rand_seq(rng, 5000), rand_seq(rng, 3), rand_seq(rng, k), rand_seq(rng, 2047 + k), rand_seq(rng, 2048 + k),
            rand_seq(rng, 7000, b"ACGTN"), rand_seq(rng, 3000, b"ACGTacgt"),          # mixed case: raw-byte orientation
            rand_seq(rng, 2000, b"acgtn"), b"A" * 300, (b"ACGT" * 100), b"", rand_seq(rng, 900, b"ACGTRYKM-")]
    seqs.append(seqs[0][100:900])                                                      # repeats -> multiplicities > 1
    want = orc.Kmers(k)
    for s in seqs:
        want.kmerize_vector(s, 1)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(seqs[:5], 0)
    ks.add_seqs(seqs[5:], 0)
    assert ks.finalize() == len(want)
    assert ks.as_dict() == want.as_dict()
    vals, cnts = ks.histogram()
    wc = want.counts()
    assert dict(zip(vals.tolist(), cnts.tolist())) == {int(v): int((wc == v).sum()) for v in np.unique(wc)}
    ks.clean(1)
    assert ks.as_dict() == want.clean_map(1).as_dict()
    ks.close()


def test_fastq_mode_and_lowercase_refusal(orc, hip_ctx):
    import colorid_amd
    rng = np.random.default_rng(9)
    seqs = [rand_seq(rng, 150, b"ACGTN" if i % 7 == 0 else b"ACGT") for i in range(3000)] + [b"ACG", b""]
    seqs += seqs[:500]
    want = fastq_counts(orc, seqs, 27)
    ks = colorid_amd.KmerSet(hip_ctx, 27)
    ks.add_seqs(seqs, 1)
    ks.finalize()
    assert ks.as_dict() == want.as_dict()
    ks.close()
    ks = colorid_amd.KmerSet(hip_ctx, 27)
    with pytest.raises(colorid_amd.CidError) as ei:          # case-preserving count cannot pack a lower-case k-mer
        ks.add_seqs([rand_seq(rng, 150, b"ACGTacgt")], 1)
    assert ei.value.code == -4
    ks.close()
    with pytest.raises(colorid_amd.CidError):
        colorid_amd.KmerSet(hip_ctx, 129)


@pytest.mark.parametrize("mode", [0, 1])
def test_add_seqs_dev_equals_add_seqs(orc, hip_ctx, mode):
    """cid_kmerset_add_seqs_dev: reads already in HBM (the FASTQ front end's packed batch) give the set the host-pointer call gives —
    ragged reads, reads shorter than k, empty ones, N runs, several calls growing the window buffer; refusals: a sequence longer than a
    segment, a byte-string set (k > 32), a lower-case base in fastq mode."""
    import torch

    import colorid_amd
    rng = np.random.default_rng(31 + mode)
    k = 27
    seqs = [rand_seq(rng, int(rng.integers(0, 400)), b"ACGTN" if i % 5 == 0 else b"ACGT") for i in range(4000)] + [b"", b"ACGT" * 5]
    if mode == 0:
        seqs += [rand_seq(rng, 120, b"ACGTacgt") for _ in range(50)]          # FASTA mode upper-cases
    a = colorid_amd.KmerSet(hip_ctx, k)
    b = colorid_amd.KmerSet(hip_ctx, k)
    dev = torch.device("cuda", 0)
    for part in (seqs[:1500], seqs[1500:1501], seqs[1501:]):
        a.add_seqs(part, mode)
        off = np.zeros(len(part) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in part])
        d_bases = torch.from_numpy(np.frombuffer(b"".join(part) + b"\0", np.uint8).copy()).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        torch.cuda.synchronize()
        b.add_seqs_dev(d_bases.data_ptr(), d_off.data_ptr(), len(part), max(len(x) for x in part), mode)
    a.finalize(); b.finalize()
    assert len(a) == len(b) > 100_000 and a.as_dict() == b.as_dict()
    c = colorid_amd.KmerSet(hip_ctx, k)
    long_seq = rand_seq(rng, 5000, b"ACGT")
    d_bases = torch.from_numpy(np.frombuffer(long_seq, np.uint8).copy()).to(dev)
    d_off = torch.tensor([0, 5000], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with pytest.raises(colorid_amd.CidError) as ei:
        c.add_seqs_dev(d_bases.data_ptr(), d_off.data_ptr(), 1, 5000, mode)
    assert ei.value.code == -4
    if mode == 1:
        low = rand_seq(rng, 150, b"acgt")
        d_bases = torch.from_numpy(np.frombuffer(low, np.uint8).copy()).to(dev)
        d_off = torch.tensor([0, 150], dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(colorid_amd.CidError) as ei:
            c.add_seqs_dev(d_bases.data_ptr(), d_off.data_ptr(), 1, 150, 1)
        assert ei.value.code == -4 and "lower-case" in str(ei.value)
    g = colorid_amd.KmerSet(hip_ctx, 40)
    with pytest.raises(colorid_amd.CidError) as ei:
        g.add_seqs_dev(d_bases.data_ptr(), d_off.data_ptr(), 1, 150, mode)
    assert ei.value.code == -4
    for s_ in (a, b, c, g):
        s_.close()


@pytest.mark.parametrize("k", [33, 40, 48, 49, 64, 100, 128])
def test_byte_string_sets_k_above_32(orc, hip_ctx, k):
    """k > 32 (src/kmer.rs:87-125 and :461-510 at k >= 33): keys are byte strings — sorted on a 4-bit-per-base image, run-length
    counted, kept as n x k ASCII.  FASTA mode upper-cases after the raw-byte orientation choice, fastq mode keeps the case."""
    import colorid_amd
    rng = np.random.default_rng(k)
    seqs = [rand_seq(rng, 4000), rand_seq(rng, k - 1), rand_seq(rng, k), rand_seq(rng, 2047 + k), rand_seq(rng, 2048 + k), rand_seq(rng, 5000, b"ACGTN"),
            rand_seq(rng, 3000, b"ACGTacgt"), rand_seq(rng, 1500, b"acgtn"), b"A" * 400, (b"ACGT" * 150), b"", rand_seq(rng, 900, b"ACGTRYKM-")]
    seqs.append(seqs[0][100:1500])
    seqs.append(seqs[6][50:900])
    want = orc.Kmers(k)
    for s in seqs:
        want.kmerize_vector(s, 1)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(seqs[:5], 0)
    ks.add_seqs(seqs[5:], 0)
    assert ks.finalize() == len(want)
    assert ks.as_dict() == want.as_dict()
    vals, cnts = ks.histogram()
    wc = want.counts()
    assert dict(zip(vals.tolist(), cnts.tolist())) == {int(v): int((wc == v).sum()) for v in np.unique(wc)}
    # searches over the device-resident byte-string set == the oracle over the same k-mers
    oix = random_index(orc, rng, 20_011, 3, k, 100, density=0.2, zero_row_frac=0.0)
    keys = want.keys()
    plant(oix, rng, keys[:2000], frac=0.8)
    hx = to_hip_index(hip_ctx, oix)
    km, cnt = ks.download()
    w = oix.search_count(km, cnt.astype(np.uint64))
    g = ks.search_count(hx)
    assert all(np.array_equal(a, b) for a, b in zip(w, g))
    pw, pm = oix.search_perfect(km)
    gw, gm = ks.search_perfect(hx)
    assert pm == gm and np.array_equal(pw, gw)
    hx.close()
    ks.clean(1)
    assert ks.as_dict() == want.clean_map(1).as_dict()
    ks.close()
    # fastq mode keeps the case: lower-case k-mers are their own keys
    fq = [rand_seq(rng, 150, b"ACGTN" if i % 5 == 0 else (b"ACGTacgt" if i % 11 == 0 else b"ACGT")) for i in range(1500)] + [b"ACG", b""]
    fq += fq[:300]
    wantq = fastq_counts(orc, fq, k)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(fq[:700], 1)
    ks.add_seqs(fq[700:], 1)
    ks.finalize()
    assert ks.as_dict() == wantq.as_dict()
    # Bloom insert of the set == the oracle's inserts
    o2 = orc.Index(30_011, 3, k, 4)
    for key in wantq.keys():
        o2.insert(2, key.tobytes())
    h2 = colorid_amd.Index(hip_ctx, 30_011, 3, k, 4)
    from colorid_amd._lib import check
    check(hip_ctx.lib.cid_index_insert_kmerset(h2.h, ks.h, 2))
    h2.finalize()
    assert np.array_equal(h2.get_rows(np.arange(30_011, dtype=np.uint64)), o2.rows())
    h2.close()
    ks.close()


@pytest.mark.parametrize("cold_sort,cold_merge", [(False, False), (True, False), (False, True)])
def test_incremental_merge_path(orc, hip_ctx, monkeypatch, cold_sort, cold_merge, tune):
    """later batches are merged into the set (cid_merge.hpp: merge path, twins joined, one pass); cold_sort: the batches themselves through
    rocPRIM's LSD sorts; cold_merge: rocPRIM's merge + reduce_by_key (both in cid_kmerset_cold.hip)"""
    import colorid_amd
    tune("CID_KMERSET_COMPACT_WINDOWS", 20000)   # a merge every few batches
    if cold_sort:
        tune("CID_KMERSET_MSD_MIN", 1000000000)
    if cold_merge:
        tune("CID_KMERSET_COLD_MERGE", 1)
    rng = np.random.default_rng(3)
    genome = rand_seq(rng, 30000)
    batches = [[genome[s:s + 400] for s in rng.integers(0, len(genome) - 400, 200)] for _ in range(6)]
    want = orc.Kmers(21)
    ks = colorid_amd.KmerSet(hip_ctx, 21)
    for b in batches:
        for s in b:
            want.kmerize_vector(s, 1)
        ks.add_seqs(b, 0)
    assert ks.finalize() == len(want)
    assert ks.as_dict() == want.as_dict()
    assert want.counts().max() > 5
    ks.close()


def test_histogram_with_multiplicities_beyond_the_lds_bins(orc, hip_ctx):
    """cid_kmerset_count_histogram: multiplicities below 4096 are counted in LDS, the rest listed — a 300-base unit read 9000 times"""
    import colorid_amd
    rng = np.random.default_rng(17)
    unit, rare = rand_seq(rng, 300), rand_seq(rng, 5000)
    ks = colorid_amd.KmerSet(hip_ctx, 21)
    want = orc.Kmers(21)
    seqs = [unit] * 9000 + [unit[:150]] * 700 + [rare]
    for s in (unit, unit[:150], rare):
        want.kmerize_vector(s, 1)
    ks.add_seqs(seqs, 0)
    assert ks.finalize() == len(want)
    got = ks.as_dict()
    assert max(got.values()) >= 9700
    import collections
    hist = collections.Counter(got.values())
    vals, cnts = ks.histogram()
    assert list(vals) == sorted(hist) and [int(c) for c in cnts] == [hist[v] for v in sorted(hist)]
    ks.clean(4095)
    assert len(ks) == sum(c for v, c in hist.items() if v > 4095) and all(v > 4095 for v in ks.as_dict().values())
    ks.close()


@pytest.mark.parametrize("targeted", [False, True])
def test_merge_of_large_lists_with_every_kind_of_overlap(orc, hip_ctx, monkeypatch, targeted, tune):
    """cid_merge.hpp beyond one tile: a set of ~400 k k-mers joined by batches that repeat it wholly, partly and not at all (twins at tile and
    thread borders occur by the thousand), in code order and built for an index; against one set counted in one go"""
    import colorid_amd
    tune("CID_KMERSET_TARGET_SMALL", 1)
    rng = np.random.default_rng(23)
    g1, g2, g3 = rand_seq(rng, 200_000), rand_seq(rng, 150_000), rand_seq(rng, 120_000)
    batches = [[g1, g2], [g2], [g3, g1[:50_000]], [g1, g2, g3], [rand_seq(rng, 10)], [g3[60_000:]]]
    k = 25
    hx = colorid_amd.Index(hip_ctx, 3_000_017, 3, k, 8)
    hx.finalize()
    one = colorid_amd.KmerSet(hip_ctx, k)
    if targeted:
        one.set_target_index(hx)
    one.add_seqs([s for b in batches for s in b], 0)
    n_one = one.finalize()
    tune("CID_KMERSET_COMPACT_WINDOWS", 1000)   # every batch is merged into the set as it comes
    many = colorid_amd.KmerSet(hip_ctx, k)
    if targeted:
        many.set_target_index(hx)
    for b in batches:
        many.add_seqs(b, 0)
    assert many.finalize() == n_one and n_one > 400_000
    k1, c1 = one.download()
    k2, c2 = many.download()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2)     # same k-mers, same order, same multiplicities
    assert int(c1.max()) >= 3
    one.close(); many.close(); hx.close()


def test_phage_fixture_counts(orc, hip_ctx):
    import colorid_amd
    for name, k in (("Listeria_phage_B056.fasta", 27), ("Listeria_phage_B021.fasta", 31)):
        seqs = orc.read_fasta(os.path.join(REFS, name))
        want = orc.Kmers(k)
        for s in seqs:
            want.kmerize_vector(s, 1)
        ks = colorid_amd.KmerSet(hip_ctx, k)
        ks.add_seqs(seqs, 0)
        n = ks.finalize()
        assert n == len(want) and (name != "Listeria_phage_B056.fasta" or n == 32634)   # SURVEY §6 sizing KAT
        assert ks.as_dict() == want.as_dict()
        ks.close()


@pytest.mark.parametrize("n_colors,n_hash,k", [(4, 4, 27), (256, 4, 31), (1024, 3, 21), (65, 2, 32), (100, 3, 40), (256, 2, 64)])
def test_search_over_device_set(orc, hip_ctx, n_colors, n_hash, k):
    import colorid_amd
    rng = np.random.default_rng(n_colors + k)
    genome = rand_seq(rng, 20000)
    reads = [genome[s:s + 150] for s in rng.integers(0, len(genome) - 150, 1500)]
    want = orc.Kmers(k)
    for s in reads:
        want.kmerize_vector(s, 1)
    oix = random_index(orc, rng, 60_013, n_hash, k, n_colors, density=0.2, zero_row_frac=0.1)
    plant(oix, rng, want.keys(), frac=0.6, max_colours=min(3, n_colors))
    hx = to_hip_index(hip_ctx, oix)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(reads, 0)
    ks.finalize()
    for reorder in (False, True):
        if reorder:
            ks.order_for_index(hx)
        km, cnt = ks.download()
        assert {km[i].tobytes(): int(cnt[i]) for i in range(len(cnt))} == want.as_dict()
        w = oix.search_count(km, cnt.astype(np.uint64))         # the oracle on the same k-mers in the set's order
        g = ks.search_count(hx)
        for a, b in zip(w, g):
            assert np.array_equal(a, b)
        assert w[0].sum() > 0
    sub = colorid_amd.KmerSet(hip_ctx, k)
    sub.add_seqs([genome[:300]], 0)
    sub.finalize()
    skm, _ = sub.download()
    for key in skm:
        oix.insert(0, key.tobytes())
    hx2 = to_hip_index(hip_ctx, oix)
    pw, pm = oix.search_perfect(skm)
    gw, gm = sub.search_perfect(hx2)
    assert pm == gm and np.array_equal(pw, gw) and (pm or pw[0] & 1)
    hx.close(); hx2.close(); ks.close(); sub.close()


@pytest.mark.parametrize("n_colors", [256, 77])
def test_report_modes_of_one_colour_and_one_multiplicity(orc, hip_ctx, n_colors):
    """Whole waves of k-mers that hit one colour uniquely at one multiplicity (an isolate searched against an index that holds it once): the
    mode table's wave-wide shortcut (k_mode_hist adds such a wave's count once) next to waves of mixed cells."""
    import colorid_amd
    k = 31
    rng = np.random.default_rng(n_colors)
    oix = random_index(orc, rng, 200_003, 3, k, n_colors, density=0.0, zero_row_frac=0.0)
    long_seq, three = rand_seq(rng, 30_000), rand_seq(rng, 9_000)
    seqs = [long_seq] + [three] * 3 + [rand_seq(rng, 200) for _ in range(40)]
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(seqs, 0)
    ks.finalize()
    km, cnt = ks.download()
    for colour, s_ in ((5, long_seq), (n_colors - 1, three)):
        one = orc.Kmers(k)
        one.kmerize_vector(s_, 1)
        for key in one.keys():
            oix.insert(colour, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    w = oix.search_count(km, cnt.astype(np.uint64))
    modes = orc.unique_modes(w[3], cnt.astype(np.uint64), n_colors)
    hits, nu, sf, md = ks.search_count_report(hx)
    assert np.array_equal(hits, w[0]) and np.array_equal(nu, w[1]) and np.array_equal(sf, w[2]) and np.array_equal(md, modes)
    assert md[5] == 1 and md[n_colors - 1] == 3 and nu[5] > 29_000 and nu[n_colors - 1] > 8_000
    hx.close()
    ks.close()


@pytest.mark.parametrize("n_colors,k", [(256, 31), (46, 27), (3000, 21), (20000, 25), (100, 40)])
def test_report_outputs_on_the_device(orc, hip_ctx, n_colors, k):
    """cid_search_count_set_report: hits / n_unique / sum / MODE per colour (reports.rs:65-77, ties -> smallest value) computed on the
    device == the oracle's unique_modes over the per-k-mer results; multiplicities both below and above the LDS table's range."""
    import colorid_amd
    rng = np.random.default_rng(n_colors + k)
    m = 30_011 if n_colors < 5000 else 3001
    oix = random_index(orc, rng, m, 3, k, n_colors, density=0.01 if n_colors < 5000 else 0.001, zero_row_frac=0.0)
    mults = [1, 2, 3, 7, 30, 63, 64, 65, 130, 2, 64, 5]             # both tiers of the mode histogram (table: f < 64 at 256 colours)
    base = [rand_seq(rng, 300) for _ in mults]
    seqs = []
    for s_, mu in zip(base, mults):
        seqs += [s_] * mu
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs(seqs, 0)
    ks.finalize()
    km, cnt = ks.download()
    # colour i (< 12) holds k-mers of base sequence i alone (mode = that sequence's multiplicity), further colours share them
    # (not unique); colour 20 holds exactly five k-mers each of sequences 1 and 2: a tie, resolved to the smaller multiplicity
    for i, s_ in enumerate(base):
        one = orc.Kmers(k)
        one.kmerize_vector(s_, 1)
        cols = [c for c in range(min(n_colors, 40)) if c % len(mults) == i and c != 20]
        for j, key in enumerate(one.keys()):
            if i in (1, 2) and j < 5:
                oix.insert(20, key.tobytes())
                continue
            for c in cols[:1] if rng.random() < 0.5 else cols:
                oix.insert(int(c), key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    w = oix.search_count(km, cnt.astype(np.uint64))
    modes = orc.unique_modes(w[3], cnt.astype(np.uint64), n_colors)
    hits, nu, sf, md = ks.search_count_report(hx)
    assert np.array_equal(hits, w[0]) and np.array_equal(nu, w[1]) and np.array_equal(sf, w[2])
    assert np.array_equal(md, modes), (np.flatnonzero(md != modes)[:5], md[md != modes][:5], modes[md != modes][:5])
    assert (nu > 0).sum() >= 10 and len(np.unique(modes)) > 4 and modes.max() >= 64 and (n_colors <= 20 or modes[20] == 2)
    hx.close()
    ks.close()


def test_many_add_calls_on_a_used_context(orc, hip_ctx):
    """The window buffer of a set grows across cid_kmerset_add_seqs calls; the old contents must be carried over before the old
    block is recycled (a null-stream device-to-device hipMemcpy once let this call's own base upload overtake that copy: ASCII
    bases turned up as k-mer codes).  Six calls of growing size, three rounds on a context whose scratch blocks have been used."""
    import colorid_amd
    rng = np.random.default_rng(99)
    for k in (32, 21):
        for rnd in range(3):
            seqs = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(200, 900))).astype(np.uint8)) for _ in range(700)]
            ks = colorid_amd.KmerSet(hip_ctx, k)
            at = 0
            for n in (5, 20, 60, 115, 200, 300):
                ks.add_seqs(seqs[at:at + n], 0)
                at += n
            ks.finalize()
            om = orc.Kmers(k)
            for s in seqs[:at]:
                om.kmerize_vector(s, 1)
            assert ks.as_dict() == om.as_dict(), (k, rnd)
            ks.close() if hasattr(ks, "close") else None


@pytest.mark.parametrize("k", [6, 11, 21, 27, 31])
@pytest.mark.parametrize("flavour", ["random", "deep", "repeats", "one_kmer", "ns", "mixed", "shared_prefix", "deep_errors"])
def test_msd_sort_path_equals_the_oracle(orc, hip_ctx, monkeypatch, k, flavour, tune):
    """The k-mer set's own sort (cid_partition.hpp: MSD radix partition passes, then every run finished in LDS) on inputs small enough
    for the oracle — CID_KMERSET_MSD_MIN=1 sends them through the kernels large sets take: evenly spread codes (the LDS bucket sort),
    every k-mer many times (deep coverage: k_run_dedupe_sort for crowded runs, the radix kernel for what is not copies), low-complexity sequence (runs beyond a workgroup's LDS:
    the per-run fallback), a single k-mer, N runs (sentinels dropped by the first pass), and all of it at once."""
    import colorid_amd
    tune("CID_KMERSET_MSD_MIN", 1)
    rng = np.random.default_rng(k * 31 + len(flavour))
    if flavour == "random":
        seqs = [rand_seq(rng, 60_000), rand_seq(rng, 45_000)]
    elif flavour == "deep":                                    # 2 kb of sequence read 150 times over
        g = rand_seq(rng, 2000)
        seqs = [g[s:s + 150] for s in rng.integers(0, len(g) - 150, 2000)]
    elif flavour == "repeats":                                 # a 7-base unit with rare substitutions: few distinct, similar k-mers
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
    for mode in (0, 1):
        want = orc.Kmers(k)
        for s in seqs:
            if mode == 0:
                want.kmerize_vector(s, 1)
            else:
                want.kmerize_fq_read(s, b"I" * len(s), 0)
        ks = colorid_amd.KmerSet(hip_ctx, k)
        ks.add_seqs(seqs, mode)
        assert ks.finalize() == len(want)
        km, cnt = ks.download()
        assert ks.as_dict() == want.as_dict()
        order = [bytes(r) for r in km]
        assert order == sorted(order)                         # ascending code == ascending ASCII (A < C < G < T)
        assert int(cnt.sum()) == int(want.counts().sum())
        ks.close()
