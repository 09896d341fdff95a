"""Colour-striped search on one GPU: an index split into 2-3 stripes (separate cid_index objects) must give exactly the
whole-index results — per-colour hits, the exactly-one-colour statistics and the perfect-search AND / absent-row flag."""
import numpy as np
import pytest
import torch

from util import plant, random_index, random_kmers

pytestmark = pytest.mark.gpu


def stripe_indices(ctx, orc, oix, bounds):
    import colorid_amd
    rows = oix.rows()
    out = []
    for lo, hi in bounds:
        assert lo % 64 == 0
        nc = hi - lo
        w32 = (nc + 31) // 32
        sub = np.zeros((oix.m, w32), np.uint32)
        src = rows[:, lo // 32:(hi + 31) // 32].copy()
        if nc % 32:
            src[:, -1] &= np.uint32((1 << (nc % 32)) - 1)
        sub[:, :src.shape[1]] = src
        hx = colorid_amd.Index(ctx, oix.m, oix.n_hash, oix.k, nc)
        hx.put_dense(sub)
        out.append((hx.finalize(), lo))
    return out


@pytest.mark.parametrize("n_colors,bounds,m", [(300, [(0, 128), (128, 300)], 30_011), (1000, [(0, 320), (320, 640), (640, 1000)], 30_011),
                                                (130, [(0, 64), (64, 128), (128, 130)], 30_011),
                                                # a stripe wider than 8192 colours runs the wide-row kernels (rows of whole KiB)
                                                (9100, [(0, 8832), (8832, 9100)], 3001), (17000, [(0, 8448), (8448, 17000)], 2003)])
def test_striped_equals_whole(orc, hip_ctx, n_colors, bounds, m):
    from colorid_amd.striped import StripedIndex
    rng = np.random.default_rng(n_colors)
    oix = random_index(orc, rng, m, 3, 31, n_colors, density=0.02 if n_colors < 8192 else 0.002, zero_row_frac=0.3)
    kmers = random_kmers(rng, 4000, 31)
    plant(oix, rng, kmers, frac=0.8, max_colours=2)
    for km in kmers[:300]:                                  # a perfect-search subset present in colours 1 and C-1
        oix.insert(1, km.tobytes())
        oix.insert(n_colors - 1, km.tobytes())
    freq = rng.integers(1, 50, size=len(kmers)).astype(np.uint32)
    stripes = stripe_indices(hip_ctx, orc, oix, bounds)
    si = StripedIndex(hip_ctx, stripes, n_colors)
    dk = torch.from_numpy(kmers.reshape(-1)).cuda().reshape(len(kmers), 31)
    df = torch.from_numpy(freq.astype(np.int32)).cuda()
    hits, nu, sf, uc = si.search_count(dk, df)
    w = oix.search_count(kmers, freq.astype(np.uint64))
    assert np.array_equal(hits.cpu().numpy().astype(np.uint64), w[0])
    assert np.array_equal(nu.cpu().numpy().astype(np.uint64), w[1])
    assert np.array_equal(sf.cpu().numpy().astype(np.uint64), w[2])
    assert np.array_equal(uc.cpu().numpy().view(np.uint32), w[3])
    assert w[1].sum() > 100                                  # unique hits that straddle stripe boundaries are exercised
    # perfect search: present subset, then one with an absent row
    w64_total = (n_colors + 63) // 64
    for sel in (slice(0, 300), slice(0, 1000)):
        sub = kmers[sel]
        ds = torch.from_numpy(sub.reshape(-1)).cuda().reshape(len(sub), 31)
        aw, missing = si.search_perfect(ds, w64_total, lambda base: base // 64)
        pw, pm = oix.search_perfect(sub)
        got32 = aw.cpu().numpy().view(np.uint32)[:oix.w32]
        assert missing == pm and np.array_equal(got32, pw)
    for hx, _ in stripes:
        hx.close()


def test_stripe_calls_refuse_minimizer_index(orc, hip_ctx):
    """`search` is not defined on .mxi indices (src/main.rs:569-573): the stripe entry points refuse them like the plain ones."""
    import colorid_amd
    from colorid_amd._lib import vp
    hx = colorid_amd.Index(hip_ctx, 1009, 2, 21, 64)
    assert hip_ctx.lib.cid_index_set_minimizer(hx.h, 11) == 0
    hx.finalize()
    k = torch.zeros((4, 21), dtype=torch.uint8, device="cuda") + 65
    h = torch.zeros(64, dtype=torch.int64, device="cuda")
    a = torch.zeros(4, dtype=torch.int32, device="cuda")
    b = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert hip_ctx.lib.cid_search_count_stripe_dev(hip_ctx.h, hx.h, vp(k.data_ptr()), None, 4, 0, vp(h.data_ptr()), vp(a.data_ptr()),
                                                   vp(b.data_ptr())) == -4
    assert hip_ctx.lib.cid_search_perfect_stripe_dev(hip_ctx.h, hx.h, vp(k.data_ptr()), None, 4, vp(h.data_ptr()), vp(a.data_ptr())) == -4
    hx.close()


@pytest.mark.parametrize("n_colors,bounds", [(300, [(0, 128), (128, 300)]), (700, [(0, 256), (256, 512), (512, 700)])])
@pytest.mark.parametrize("paired", [False, True])
def test_readid_over_stripes_equals_whole_index(orc, hip_ctx, n_colors, bounds, paired):
    """read_id with the index cut into colour stripes == the oracle on the whole index: the absent-row stop (a row that is zero in
    one stripe but set in another is NOT absent), -B S sampling, -d strides, too-short and all-N reads."""
    from colorid_amd.striped import StripedIndex
    from test_gpu_readid import pack_reads
    rng = np.random.default_rng(n_colors + paired)
    k, n, m = 21, 3, 40_009
    oix = random_index(orc, rng, m, n, k, n_colors, density=0.02, zero_row_frac=0.004)    # rows often zero in one stripe and set in another; ~1 % of the k-mers meet a truly absent row
    genome = bytes(rng.choice(list(b"ACGT"), size=30_000).astype(np.uint8))
    for pos in range(0, len(genome) - k, 3):                                                # plant part of the genome in a few colours
        km = orc.Kmers(k)
        km.kmerize_vector(genome[pos:pos + k], 1)
        for key in km.keys():
            for c in rng.choice(n_colors, size=2, replace=False):
                oix.insert(int(c), key.tobytes())
    reads = []
    for i in range(400):
        st = int(rng.integers(0, len(genome) - 400))
        L = int(rng.integers(60, 151))
        mates = [genome[st:st + L]]
        if paired:
            mates.append(orc.revcomp(genome[st + 150:st + 150 + int(rng.integers(30, 151))]))
        reads.append(mates)
    reads += [[b"ACG"] + ([genome[:100]] if paired else []), [b"N" * 80] + ([b"N" * 5] if paired else []),
              [bytes(rng.choice(list(b"ACGT"), size=120).astype(np.uint8))] + ([bytes(rng.choice(list(b"ACGT"), size=90).astype(np.uint8))] if paired else [])]
    bases, seq_off, read_seq0 = pack_reads(reads)
    stripes = stripe_indices(hip_ctx, orc, oix, bounds)
    si = StripedIndex(hip_ctx, stripes, n_colors)
    db = torch.from_numpy(bases.copy()).cuda()
    dso = torch.from_numpy(seq_off.astype(np.int64)).cuda()
    dr0 = torch.from_numpy(read_seq0.astype(np.int64)).cuda()
    max_bytes = max(sum(len(s) for s in r) for r in reads)
    stopped_somewhere, total_counts = False, 0
    for d, S in ((1, 3), (1, 0), (4, 2), (10, 3)):
        max_win = max(sum(((len(s) - k) // d + 1) if len(s) >= k else 0 for s in r) for r in reads)
        want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
        rep, nk, st = si.readid_count(db, dso, dr0, len(reads), d, S, max_bytes, max_win)
        assert np.array_equal(st.cpu().numpy(), want[2]) and np.array_equal(nk.cpu().numpy().view(np.uint32), want[1])
        got = rep.cpu().numpy().view(np.uint32)
        bad = np.flatnonzero((got != want[0]).any(axis=1))
        assert len(bad) == 0, (d, S, bad[:5], want[0][bad[:1]], got[bad[:1]])
        stopped_somewhere |= bool(want[0][:, n_colors].any())
        total_counts += int(want[0][:, :n_colors].sum())
    assert stopped_somewhere and total_counts > 5000     # the absent-row stop was exercised, and colours were counted
    for hx, _ in stripes:
        hx.close()


def _genome_index(orc, rng, m, n_hash, k, n_colors, genomes, colours_of):
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.01, zero_row_frac=0.02)
    for gi, g in enumerate(genomes):
        km = orc.Kmers(k)
        km.kmerize_vector(g, 1)
        for key in km.keys():
            for c in colours_of(gi):
                oix.insert(int(c), key.tobytes())
    return oix


@pytest.mark.parametrize("n_colors,bounds,k", [(300, [(0, 128), (128, 300)], 21),                # narrow stripes of different width
                                               (9100, [(0, 8832), (8832, 9100)], 21),             # a wide-row stripe (> 8192 colours) + a narrow one
                                               (17000, [(0, 8448), (8448, 17000)], 21),           # two wide-row stripes
                                               (200, [(0, 64), (64, 200)], 40)])                  # k > 32: byte-string keys
def test_readid_routed_over_stripes_any_length(orc, hip_ctx, n_colors, bounds, k):
    """cid_readid_stripe_zero / _count: reads of any length over stripes of any width == the oracle on the whole index.  Long reads
    (whole genomes, a chimera, long pairs) take the sort-based path, short ones the LDS kernels — per stripe, so the narrow and the
    wide stripe of one index route the same read differently — and the k-mer masks of both land in one array."""
    from colorid_amd.striped import StripedIndex
    from test_gpu_readid import pack_reads
    rng = np.random.default_rng(n_colors + k)
    m, n_hash = 40_009, 2
    genomes = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 30_000)].tobytes() for _ in range(3)]
    oix = _genome_index(orc, rng, m, n_hash, k, n_colors, [g[:25_000] for g in genomes],
                        lambda gi: (gi, n_colors - 1 - gi, bounds[0][1] + gi))        # colours on both sides of the stripe cut
    low = bytearray(genomes[1][:20_000])
    low[3000:3600] = bytes(low[3000:3600]).lower()
    reads = [[genomes[0]], [genomes[1][:12_000], genomes[2][5_000:20_000]], [genomes[0][100:250]], [b"ACG"], [bytes(low)],
             [genomes[2][:2_000]], [genomes[2][:5_000] * 2 + b"N" * 30 + genomes[0][:4_000]], [genomes[1][200:330], genomes[1][400:520]],
             [genomes[0][7:157].lower()], [b"N" * 3_000]]
    for i in range(60):
        g = genomes[i % 3]
        L = int(rng.choice([40, 150, 700, 2600, 6000]))
        s0 = int(rng.integers(0, len(g) - L))
        reads.append([g[s0:s0 + L]] if i % 3 else [g[s0:s0 + L], g[s0:s0 + min(L, 300)]])
    bases, seq_off, read_seq0 = pack_reads(reads)
    stripes = stripe_indices(hip_ctx, orc, oix, bounds)
    si = StripedIndex(hip_ctx, stripes, n_colors)
    db = torch.from_numpy(bases.copy()).cuda()
    stopped_somewhere = False
    for d, S in ((1, 3), (1, 0), (5, 2)):
        want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
        rep, nk, st = si.readid_count_routed(db, seq_off, read_seq0, d, S)
        assert np.array_equal(st.cpu().numpy(), want[2]) and np.array_equal(nk.cpu().numpy().view(np.uint32), want[1])
        got = rep.cpu().numpy().view(np.uint32)
        bad = np.flatnonzero((got != want[0]).any(axis=1))
        assert len(bad) == 0, (d, S, bad[:5], np.flatnonzero(got[bad[0]] != want[0][bad[0]])[:8])
        assert want[0][0, 0] > 1000 and want[0][0, n_colors - 1] > 1000     # the whole genome hits its colours in both stripes
        stopped_somewhere |= bool(want[0][:, n_colors].any())
    assert stopped_somewhere
    # an empty batch, and misuse
    from colorid_amd._lib import vp
    import ctypes
    lib = si.lib
    so0 = np.zeros(1, np.uint64)
    assert lib.cid_readid_stripe_zero(hip_ctx.h, stripes[0][0].h, None, so0.ctypes.data, 0, so0.ctypes.data, 0, 1, None, None, None) == 0
    nw = ctypes.c_uint64(0)
    assert lib.cid_readid_stripe_mask_words(k, 0, seq_off.ctypes.data, read_seq0.ctypes.data, len(reads), ctypes.byref(nw)) == -1
    z = torch.zeros(8, dtype=torch.int32, device="cuda")
    assert lib.cid_readid_stripe_count(hip_ctx.h, stripes[1][0].h, vp(db.data_ptr()), seq_off.ctypes.data, len(seq_off) - 1, read_seq0.ctypes.data,
                                       len(reads), 1, 0, n_colors, n_colors, 1, vp(z.data_ptr()), vp(z.data_ptr()), vp(z.data_ptr()), vp(z.data_ptr())) == -1
    for hx, _ in stripes:
        hx.close()
