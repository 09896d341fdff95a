"""K-mer counting over the ranks of a group (cid_group_kmerset, SURVEY.md §8e.1 caveat / §8f.1): every rank counts the windows of its
share of the sequences, the code space is cut into ranges, the ranges travel to their owners and are merged.  The result must be
the single-GPU cid_kmerset of the same sequences — same k-mers, same multiplicities, same order — and the searches over the parts
must equal the single-set searches and the oracle.  1-4 ranks share the one GPU."""
import numpy as np
import pytest

from util import plant, random_index, random_kmers

pytestmark = pytest.mark.gpu


def _seqs(rng, n, L, dup=0.3):
    base = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(L // 2, L + 1))).astype(np.uint8)) for _ in range(n)]
    out = []
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i, s in enumerate(base):
        out.append(s)
        if rng.random() < dup:                       # the same stretch again elsewhere (another rank's shard), as its reverse complement
            out.append(s[len(s) // 4:].translate(comp)[::-1])
        if i % 7 == 0:
            a = bytearray(s[:60]); a[10] = ord("N"); out.append(bytes(a))
        if i % 11 == 0:
            out.append(b"ACGT")                      # shorter than k
    order = rng.permutation(len(out))
    return [out[i] for i in order]


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0], [0], [0, 0, 0]])
@pytest.mark.parametrize("k,mode", [(31, 0), (21, 1), (32, 0), (5, 0)])
def test_group_kmerset_equals_single_gpu_set(orc, hip_ctx, devices, k, mode):
    import colorid_amd
    rng = np.random.default_rng(k * 10 + len(devices))
    seqs = _seqs(rng, 300, 400)
    one = colorid_amd.KmerSet(hip_ctx, k)
    one.add_seqs(seqs[:200], mode)
    one.add_seqs(seqs[200:], mode)
    one.finalize()
    g = colorid_amd.Group(devices)
    gs = g.kmerset(k)
    gs.add_seqs(seqs[:200], mode)
    gs.add_seqs(seqs[200:], mode)
    n = gs.finalize()
    assert n == len(one) == len(gs)
    km1, c1 = one.download()
    km2, c2 = gs.download()
    assert np.array_equal(km1, km2) and np.array_equal(c1, c2)          # same k-mers, same counts, same order
    parts = gs.part_sizes()
    assert int(parts.sum()) == n
    if len(devices) > 1 and k > 5:
        assert parts.min() > 0.5 * n / len(devices), parts               # the quantile splitters balance the ranks
    h1, h2 = one.histogram(), gs.histogram()
    assert np.array_equal(h1[0], h2[0]) and np.array_equal(h1[1], h2[1])
    one.clean(1); gs.clean(1)
    km1, c1 = one.download(); km2, c2 = gs.download()
    assert (len(c1) < n or k == 5) and np.array_equal(km1, km2) and np.array_equal(c1, c2)
    # the oracle's k-mer map of the same sequences (FASTA mode upper-cases; fastq mode keeps case — all upper here)
    om = orc.Kmers(k)
    for s in seqs:
        if mode == 0:
            om.kmerize_vector(s, 1)
        else:
            om.kmerize_fq_read(s, b"I" * len(s), 15)
    want = {key.tobytes(): int(c) for key, c in zip(om.keys(), om.counts()) if c > 1}
    got = {km2[i].tobytes(): int(c2[i]) for i in range(len(c2))}
    assert got == want
    g.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_search_over_parts_equals_single_set_and_oracle(orc, hip_ctx, devices):
    import colorid_amd
    k, C = 31, 300
    rng = np.random.default_rng(len(devices))
    oix = random_index(orc, rng, 40_009, 4, k, C, density=0.15, zero_row_frac=0.05)
    seqs = _seqs(rng, 120, 500)
    g = colorid_amd.Group(devices)
    gs = g.kmerset(k)
    gs.add_seqs(seqs, 0)
    gs.finalize()
    km, cnt = gs.download()
    plant(oix, rng, km[: len(km) // 2], frac=0.8)
    hx = colorid_amd.Index(g.ctxs[0], oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.put_dense(oix.rows())
    hx.finalize()
    g.replicate(hx)
    want = oix.search_count(km, cnt.astype(np.uint64))
    got = gs.search_count()
    for w, x in zip(want, got):
        assert np.array_equal(w, x)
    pw, pm = oix.search_perfect(km)
    gw, gm = gs.search_perfect()
    assert gm == pm and np.array_equal(gw, pw)
    # the whole report without anything per k-mer: the ranks' (colour, multiplicity) histograms added up -> the modes
    modes = orc.unique_modes(want[3], cnt.astype(np.uint64), C)
    hits, nu, sf, md = gs.search_count_report()
    assert np.array_equal(hits, want[0]) and np.array_equal(nu, want[1]) and np.array_equal(sf, want[2]) and np.array_equal(md, modes)
    assert (want[1] > 0).sum() > 10 and len(set(md.tolist())) > 1
    # a set whose k-mers all sit in two colours: a perfect hit over parts held by different ranks
    sub = seqs[:5]
    gs2 = g.kmerset(k)
    gs2.add_seqs(sub, 0)
    gs2.finalize()
    km2, _ = gs2.download()
    for x in km2:
        oix.insert(3, x.tobytes()); oix.insert(C - 2, x.tobytes())
    g.close()
    g = colorid_amd.Group(devices)
    hx = colorid_amd.Index(g.ctxs[0], oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.put_dense(oix.rows()); hx.finalize(); g.replicate(hx)
    gs2 = g.kmerset(k)
    gs2.add_seqs(sub, 0)
    gs2.finalize()
    pw, pm = oix.search_perfect(km2)
    gw, gm = gs2.search_perfect()
    assert not pm and gm == pm and np.array_equal(gw, pw) and pw.any()
    g.close()


def test_group_kmerset_edges_and_misuse(orc, hip_ctx):
    import ctypes as C

    import colorid_amd
    from colorid_amd._lib import vp
    g = colorid_amd.Group([0, 0, 0])
    h = vp()
    assert g.lib.cid_group_kmerset_create(g.h, 33, C.byref(h)) == -4 and b"1..32" in g.lib.cid_last_error()
    gs = g.kmerset(21)
    assert gs.finalize() == 0 and len(gs) == 0                       # empty set: nothing to exchange
    assert gs.histogram()[0].size == 0
    gs = g.kmerset(21)
    gs.add_seqs([b"ACGTACGTACGTACGTACGTACGTA"], 0)                  # fewer sequences than ranks, 5 windows, a periodic sequence
    assert gs.finalize() == len(set(gs.download()[0][i].tobytes() for i in range(len(gs))))
    gs = g.kmerset(21)
    with pytest.raises(colorid_amd.CidError):                        # a lower-case base in case-keeping (fastq) mode
        gs.add_seqs([b"ACGTACGTACGTACGTACGTACGTA", b"ACGTACGTACGtACGTACGTACGTA", b"ACGTACGTACGTACGTACGTACGTA"], 1)
    gs = g.kmerset(21)
    gs.add_seqs([b"ACGTACGTACGTACGTACGTACGTAGG"], 0)
    gs.finalize()
    with pytest.raises(colorid_amd.CidError):
        gs.add_seqs([b"ACGTACGTACGTACGTACGTACGTAGG"], 0)             # already finalized
    g.close()
