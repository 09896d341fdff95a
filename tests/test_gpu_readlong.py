"""The long-read read_id path (cid_readlong.hip: workgroup-wide LDS hash tables -> first-occurrence bitmap -> ordered lists -> the
search by slices of a read) against the oracle's restatement of read_id_mt_pe.rs:66-165,282-363 / kmer.rs:221-243, bit-exact:
row widths of 8 B ... 1 KiB, reads of 3 kb ... 1.2 Mb, breaks (absent rows) in every slice position, `-B S` from 0 to beyond a read,
strides, pairs, N stretches, repeats, minimizer indices, the resident entry point, and round 1's sorting path as a second witness."""
import numpy as np
import pytest

from test_gpu_readid import pack_reads
from util import ACGT, random_index, to_hip_index

pytestmark = pytest.mark.gpu


def rnd(rng, n):
    return ACGT[rng.integers(0, 4, n)].tobytes()


def compare(oix, hx, reads, d, S, what=""):
    bases, seq_off, read_seq0 = pack_reads(reads)
    want = oix.readid_counts(bases, seq_off, read_seq0, d, S, n_threads=8)
    got = hx.readid_count(bases, seq_off, read_seq0, d, S)
    assert np.array_equal(want[2], got[2]), ("status", what)
    assert np.array_equal(want[1], got[1]), ("n_kmers", what, np.flatnonzero(want[1] != got[1])[:5], want[1][:8], got[1][:8])
    bad = np.flatnonzero((want[0] != got[0]).any(axis=1))
    assert len(bad) == 0, (what, bad[:5], want[0][bad[0]][:12], got[0][bad[0]][:12], want[0][bad[0]][-1], got[0][bad[0]][-1])
    return want


def long_reads(rng, genome):
    """reads that exercise the path: every table size, several buckets, several slices, repeats, N runs, pairs"""
    L = len(genome)
    reads = [[genome[:3_000]], [genome[1_000:5_200]], [genome[2_000:12_000]], [genome[:40_000]], [genome[5_000:5_000 + 70_000]],
             [genome[:20_000], genome[10_000:35_000]],                        # a long pair sharing k-mers across mates
             [genome[:6_000] * 5],                                            # 30 kb with every k-mer five times
             [b"N" * 9_000 + genome[100:9_000]],                              # the first two slices hold no k-mer at all
             [genome[:4_200] + b"N" * 30 + genome[:4_200] + b"N" + genome[4_000:9_000]],
             [b"ACGTTGCA" * 2_000],                                           # 16 kb, eight distinct windows
             [b"A" * 50_000],
             [genome[L - 3_500:]], [b"ACG"], [genome[:100]]]
    return reads


@pytest.mark.parametrize("n_colors,n_hash,k", [(256, 2, 21), (40, 3, 31), (100, 1, 15), (1000, 2, 27), (8192, 2, 21), (300, 4, 32)])
def test_long_reads_every_row_width(orc, hip_ctx, n_colors, n_hash, k):
    rng = np.random.default_rng(n_colors + k)
    m = 200_003 if n_colors < 1000 else 10_007
    # a few absent rows: a read of thousands of k-mers stops somewhere inside — in its first slice or in a later one
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.05, zero_row_frac=0.0002 if n_colors < 1000 else 0.002)
    genome = rnd(rng, 160_000)
    hx = to_hip_index(hip_ctx, oix)
    reads = long_reads(rng, genome)
    stops = 0
    for d, S in ((1, 3), (1, 0), (3, 1), (1, 64), (1, 65), (2, 5000)):
        rep = compare(oix, hx, reads, d, S, (n_colors, d, S))[0]
        stops += int((rep[:, n_colors] > 0).sum())
    assert stops > 0   # the absent-row stop was taken somewhere
    hx.close()


def test_breaks_in_every_slice_position(orc, hip_ctx):
    """one absent row planted at a chosen place of a 30 kb read: the rows stop exactly there, whichever slice holds it"""
    rng = np.random.default_rng(11)
    n_colors, n_hash, k, m = 256, 2, 21, 400_009
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0)
    genome = rnd(rng, 30_000)
    rows = oix.rows()
    keep = rows.copy()
    hx = None
    for pos in (0, 1, 2, 3, 63, 64, 4095, 4096, 4097, 8191, 12_288, 20_000, 29_979):
        rows[:] = keep
        km = genome[pos:pos + k]
        rc = km.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        canon = min(km, rc)
        rows[orc.xxh3(canon, 1) % m, :] = 0   # the row of its second hash
        if hx is not None:
            hx.close()
        hx = to_hip_index(hip_ctx, oix)
        for S in (0, 3):
            rep = compare(oix, hx, [[genome], [genome[:10_000]], [genome[pos:]]], 1, S, (pos, S))[0]
            assert rep[0, n_colors] == 1
    hx.close()


@pytest.mark.parametrize("deal", [1, 0])
def test_megabase_read_and_many_buckets(orc, hip_ctx, deal):
    """74 buckets: their windows dealt to them by the pre-pass (k_long_deal), or every bucket's pass re-reading the read"""
    hip_ctx.tune("readid_long_deal", deal)
    rng = np.random.default_rng(3)
    n_colors, n_hash, k, m = 64, 2, 21, 1_000_003
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.2, zero_row_frac=0.0)
    hx = to_hip_index(hip_ctx, oix)
    g = rnd(rng, 1_200_000)
    reads = [[g], [g[:300_000] + g[:300_000]], [g[:5_000]],
             [g[:100_000] + b"N" * 70 + g[50_000:150_000] + b"N" + g[:60_000], g[140_000:100_000:-1]]]   # N runs and third occurrences among dealt windows
    try:
        rep, nk, st = compare(oix, hx, reads, 1, 3)
    finally:
        hip_ctx.tune("readid_long_deal", 1)
    assert nk[0] > 1_190_000 and nk[1] < 300_100
    hx.close()


def test_minimizer_index_long_reads(orc, hip_ctx):
    rng = np.random.default_rng(21)
    n_colors, n_hash, k, m, msz = 128, 2, 27, 150_001, 15
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0005)
    oix.set_minimizer(msz)
    import colorid_amd
    hx = colorid_amd.Index(hip_ctx, m, n_hash, k, n_colors).set_minimizer(msz)
    hx.put_dense(oix.rows())
    hx.finalize()
    genome = rnd(rng, 90_000)
    for d, S in ((1, 3), (2, 0)):
        compare(oix, hx, long_reads(rng, genome), d, S, ("mini", d, S))
    hx.close()


def test_resident_call_and_the_sorting_path_agree(orc, hip_ctx):
    import torch
    import colorid_amd
    rng = np.random.default_rng(8)
    n_colors, n_hash, k, m = 256, 2, 21, 300_007
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0003)
    hx = to_hip_index(hip_ctx, oix)
    genome = rnd(rng, 120_000)
    reads = long_reads(rng, genome) + [[genome[i:i + 150]] for i in range(0, 6_000, 150)]     # short reads routed to the LDS kernel
    bases, seq_off, read_seq0 = pack_reads(reads)
    want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3, n_threads=8)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    n = len(reads)
    outs = []
    for lds in (1, 0):   # this round's path, then round 1's global sort
        hip_ctx.tune("readid_long_lds", lds)
        rep = torch.full((n, n_colors + 1), 77, dtype=torch.int32, device=dev)
        nk = torch.full((n,), 77, dtype=torch.int32, device=dev)
        st = torch.full((n,), 77, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hx.readid_count_resident(d_bases.data_ptr(), seq_off, read_seq0, 1, 3, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
        hip_ctx.synchronize()
        outs.append((rep.cpu().numpy().view(np.uint32), nk.cpu().numpy().view(np.uint32), st.cpu().numpy()))
    hip_ctx.tune("readid_long_lds", 1)
    for got in outs:
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    hx.close()


def test_lower_case_among_long_reads_falls_back(orc, hip_ctx):
    rng = np.random.default_rng(13)
    n_colors, n_hash, k, m = 256, 2, 21, 100_003
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0)
    hx = to_hip_index(hip_ctx, oix)
    g = rnd(rng, 50_000)
    low = bytearray(g[:30_000])
    low[12_345:12_400] = bytes(low[12_345:12_400]).lower()
    compare(oix, hx, [[g[:20_000]], [bytes(low)], [g[100:260]], [g[:9_000].lower(), g[:9000]]], 1, 3)
    hx.close()


def _dev_call(hip_ctx, hx, reads, d, S, max_bytes, max_win, unaligned=0):
    """cid_readid_count_dev: bases AND offsets on the device (what the FASTQ front end hands over)"""
    import torch
    bases, seq_off, read_seq0 = pack_reads(reads)
    dev = torch.device("cuda", 0)
    n = len(reads)
    buf = torch.zeros(len(bases) + 64, dtype=torch.uint8, device=dev)
    buf[unaligned:unaligned + len(bases)] = torch.from_numpy(bases.copy()).to(dev)
    d_so = torch.from_numpy(seq_off.astype(np.int64)).to(dev)
    d_r0 = torch.from_numpy(read_seq0.astype(np.int64)).to(dev)
    rep = torch.full((n, hx.n_colors + 1), 77, dtype=torch.int32, device=dev)
    nk = torch.full((n,), 77, dtype=torch.int32, device=dev)
    st = torch.full((n,), 77, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    hx.readid_count_dev(buf.data_ptr() + unaligned, d_so.data_ptr(), d_r0.data_ptr(), n, d, S, max_bytes, max_win, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
    hip_ctx.synchronize()
    return rep.cpu().numpy().view(np.uint32), nk.cpu().numpy().view(np.uint32), st.cpu().numpy()


@pytest.mark.parametrize("n_colors,n_hash,k", [(256, 2, 21), (1000, 3, 31), (9000, 2, 21)])
def test_device_offsets_entry_takes_long_reads(orc, hip_ctx, n_colors, n_hash, k):
    """Round 6: the device-pointer entry point routes reads of any length itself — route, work lists and window numbering are made on the
    device from seq_off / read_seq0 in HBM (k_long_route, k_long_plan, k_long_emit); long, short, too-short, empty and paired reads in
    one call, bases at an odd address; a read beyond the maxima the caller stated gets status 3 and an empty row."""
    rng = np.random.default_rng(n_colors + 7)
    m = 200_003 if n_colors < 1000 else 20_011
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.05, zero_row_frac=0.0003)
    hx = to_hip_index(hip_ctx, oix)
    genome = rnd(rng, 160_000)
    shorts = [[genome[i:i + 150]] for i in range(0, 9_000, 150)] + [[genome[i:i + 100], genome[i + 200:i + 330]] for i in range(0, 3_000, 300)]
    reads = long_reads(rng, genome)[:9] + shorts + long_reads(rng, genome)[9:] + [[]] + [[b"", genome[:5_000]]] + [[genome[:7_000], b"AC", genome[100:8_000]]]
    rng.shuffle(reads)
    bases, seq_off, read_seq0 = pack_reads(reads)
    sizes = [sum(len(s) for s in r) for r in reads]
    for d, S, unaligned in ((1, 3, 0), (1, 0, 5), (3, 64, 11)):
        want = oix.readid_counts(bases, seq_off, read_seq0, d, S, n_threads=8)
        got = _dev_call(hip_ctx, hx, reads, d, S, max(sizes), max(sizes), unaligned)
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (d, S)
    # maxima below the two longest reads: those come back with status 3, everything else as before
    cap = sorted(sizes)[-3]
    got = _dev_call(hip_ctx, hx, reads, 1, 3, cap, cap)
    want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3, n_threads=8)
    over = np.array([sz > cap for sz in sizes])
    assert over.sum() >= 1
    assert (got[2][over] == 3).all() and (got[1][over] == 0).all() and (got[0][over] == 0).all()
    assert np.array_equal(got[2][~over], want[2][~over]) and np.array_equal(got[1][~over], want[1][~over]) and np.array_equal(got[0][~over], want[0][~over])
    hx.close()


@pytest.mark.parametrize("fuse", [1, 0])
def test_soft_masked_reads_go_alone(orc, hip_ctx, fuse):
    """A lower-case base keeps its case: the read that holds one takes the byte-string (sorting) path — that read alone
    (round 5: the whole batch).  Long reads with a lower-case stretch in the first, a middle and the last window, a wholly lower-case
    mate, one of several buckets; upper-case neighbours of every class; with the fused kernel and with k_extract_codes + k_long_first_flags."""
    hip_ctx.tune("readid_long_fuse", fuse)
    try:
        rng = np.random.default_rng(131)
        n_colors, n_hash, k, m = 256, 2, 21, 100_003
        oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0004)
        hx = to_hip_index(hip_ctx, oix)
        g = rnd(rng, 90_000)

        def low(b, a, z):
            x = bytearray(b); x[a:z] = bytes(x[a:z]).lower(); return bytes(x)
        reads = [[g[:20_000]], [low(g[:30_000], 12_345, 12_400)], [g[100:260]], [g[:9_000].lower(), g[:9_000]], [low(g[3_000:6_500], 0, 1)],
                 [g[40_000:52_000]], [low(g[:12_000], 11_999, 12_000)], [low(g[:80_000], 70_000, 70_100)], [g[:80_000]], [low(g[500:700], 10, 30)],
                 [g[20_000:24_000], low(g[30_000:33_000], 5, 9)], [b"N" * 3_000 + low(g[:3_000], 100, 200)]]
        for d, S in ((1, 3), (1, 0), (2, 70)):
            compare(oix, hx, reads, d, S, ("soft", fuse, d, S))
        got = _dev_call(hip_ctx, hx, reads, 1, 3, 80_000, 80_000)
        bases, seq_off, read_seq0 = pack_reads(reads)
        want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3, n_threads=8)
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
        hx.close()
    finally:
        hip_ctx.tune("readid_long_fuse", 1)


def test_fused_and_two_kernel_paths_agree_on_every_class(orc, hip_ctx):
    """the classes of the device-made plan: one table in the fused kernel (256 and 1 024 threads), the same reads through
    k_extract_codes + k_long_first_flags (readid_long_fuse = 0), reads of more mates than the fused kernel keeps a table of, strides that
    stretch a read's bases beyond its LDS, minimizer indices; reads of two and three tables in the fused kernel's passes (readid_long_multi)"""
    import colorid_amd
    rng = np.random.default_rng(77)
    n_colors, n_hash, k, m = 200, 2, 25, 150_001
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.1, zero_row_frac=0.0005)
    hx = to_hip_index(hip_ctx, oix)
    g = rnd(rng, 120_000)
    many = [g[i * 1_000:i * 1_000 + 900] for i in range(7)]                 # seven mates: the items route
    reads = [[g[:3_000]], [g[:4_120]], [g[:4_121]], [g[:16_000]], [g[:16_408]], [g[:16_409]], [g[:40_000]], many, [g[:2_000], b"", g[:1_000], b"ACGT", g[5_000:9_000]],
             [g[:6_100]], [g[:20_400]], [g[:20_500]], [g[50_000:50_000 + 1_100]] * 4, [g[:3_000]] * 5]
    # the fused kernel's passes (two and three tables' worth of windows, the last window a pass still takes, pairs, strides that keep the
    # bases within its LDS) beside the buckets of k_long_first_flags
    reads += [[g[:32_792]], [g[:32_793]], [g[:49_176]], [g[:49_177]], [g[:24_000], g[60_000:84_000]], [g[:10_000] * 4], [g[44_999::-1]],
              [g[:30_000] + b"acgt" + g[30_000:35_000]], [b"N" * 20_000 + g[:25_000]]]
    # ... and the shortest reads of the path, a wave each (up to 1 024 windows: k = 25, so 1 048 bases), alone and as pairs
    reads += [[g[:700]], [g[:1_047]], [g[:1_048]], [g[:1_049]], [g[:500], g[300:800]], [g[:300], b"", g[:300], g[100:400]], [g[:512] * 2], [b"N" * 600 + g[:400]],
              [g[:200] + b"acgtacgt" + g[200:700]]]
    for fuse, multi, tiny in ((1, 1, 1), (1, 0, 0), (0, 1, 1)):
        hip_ctx.tune("readid_long_fuse", fuse)
        hip_ctx.tune("readid_long_multi", multi)
        hip_ctx.tune("readid_long_tiny", tiny)
        hip_ctx.tune("readid_long_from", 0)   # (every read on this path, however short)
        try:
            for d, S in ((1, 3), (4, 0), (13, 2)):
                compare(oix, hx, reads, d, S, ("classes", fuse, multi, tiny, d, S))
        finally:
            hip_ctx.tune("readid_long_fuse", 1)
            hip_ctx.tune("readid_long_multi", 1)
            hip_ctx.tune("readid_long_tiny", 1)
            hip_ctx.tune("readid_long_from", -1)
    hx.close()


def _random_long_read(rng, genome, k):
    """a read of 1 kb ... 420 kb (one to three tables in the fused kernel, more: dealt to hash buckets) made of stretches
    of the genome: some repeated, some with N runs, some reverse-complemented"""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    L = int(rng.choice([1_000, 1_500, 3_000, 4_100, 9_000, 16_500, 33_000, 50_000, 70_000, 120_000, 420_000]))
    parts, have = [], 0
    while have < L:
        n = int(min(L - have, rng.integers(200, 20_000)))
        st = int(rng.integers(0, len(genome) - n))
        p = genome[st:st + n]
        kind = int(rng.integers(0, 8))
        if kind == 0:
            p = p.translate(comp)[::-1]                    # the canonical k-mers of the other strand: the same k-mers
        elif kind == 1 and parts:
            p = parts[int(rng.integers(0, len(parts)))][:n]  # an earlier stretch again
        elif kind == 2:
            a = bytearray(p); s0 = int(rng.integers(0, n)); a[s0:s0 + int(rng.integers(1, 3 * k))] = b"N" * len(a[s0:s0 + int(rng.integers(1, 3 * k))]); p = bytes(a)
        elif kind == 3:
            p = (p[:int(rng.integers(1, 30))] * (n // 1 + 1))[:n]   # a short period
        parts.append(p); have += len(p)
    return b"".join(parts)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FUZZ_SEED0", 0)), int(__import__("os").environ.get("FUZZ_SEED0", 0)) + int(__import__("os").environ.get("FUZZ_N", 10))))
def test_long_reads_random_shapes(orc, hip_ctx, seed):
    """random index shapes (k, hashes, colours, absent rows), strides and `-B`, random long reads and pairs of them, short reads between"""
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.choice([11, 15, 21, 25, 31, 32]))
    n_hash = int(rng.integers(1, 5))
    C = int(rng.choice([3, 64, 65, 256, 700, 2000]))
    m = int(rng.choice([20_011, 65_536, 300_007]))
    oix = random_index(orc, rng, m, n_hash, k, C, density=float(rng.choice([0.02, 0.2])), zero_row_frac=float(rng.choice([0.0, 0.0005, 0.01])))
    hx = to_hip_index(hip_ctx, oix)
    genome = rnd(rng, 150_000)
    reads = []
    for _ in range(int(rng.integers(3, 9))):
        r = [_random_long_read(rng, genome, k)]
        if rng.random() < 0.3:
            r.append(_random_long_read(rng, genome, k)[:int(rng.integers(1, 30_000))])
        reads.append(r)
        if rng.random() < 0.5:
            st = int(rng.integers(0, len(genome) - 300))
            reads.append([genome[st:st + int(rng.integers(1, 300))]])
    d = int(rng.choice([1, 1, 2, 5]))
    S = int(rng.choice([0, 1, 3, 3, 50, 64, 65, 4000]))
    compare(oix, hx, reads, d, S, (seed, k, n_hash, C, m, d, S))
    hx.close()
