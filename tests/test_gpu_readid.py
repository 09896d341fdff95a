"""Parity of cid_readid_count (a6/a7/a9/a10) with the oracle's restatement of read_id_mt_pe.rs:300-331:
per-read report rows (n_colors+1 counts), k-mer set sizes and too_short flags, bit-exact."""
import os

import numpy as np
import pytest

from util import random_index, to_hip_index

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REFS = os.path.join(HERE, "golden", "refs")
PHAGES = ["Listeria_phage_B021", "Listeria_phage_B051", "Listeria_phage_B056", "Listeria_phage_B545"]


def pack_reads(reads):
    """reads: list of lists of bytes (1 = SE, 2 = PE) -> (bases, seq_off, read_seq0)"""
    seqs = [s for r in reads for s in r]
    seq_off = np.zeros(len(seqs) + 1, np.uint64)
    seq_off[1:] = np.cumsum([len(s) for s in seqs])
    read_seq0 = np.zeros(len(reads) + 1, np.uint64)
    read_seq0[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(seqs), np.uint8) if seqs else np.zeros(0, np.uint8)
    return bases, seq_off, read_seq0


def sample_reads(orc, rng, genomes, n_reads, read_len, paired, err=0.01, n_rate=0.005, lower_rate=0.02):
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(n_reads):
        g = genomes[rng.integers(len(genomes))]
        L = int(read_len if rng.random() < 0.8 else rng.integers(5, read_len + 1))
        frag = int(max(L, rng.integers(L, 2 * L + 50)))
        if len(g) <= frag:
            continue
        st = int(rng.integers(0, len(g) - frag))
        mates = [g[st:st + L]]
        if paired:
            L2 = int(read_len if rng.random() < 0.85 else rng.integers(1, read_len + 1))
            mates.append(g[st + frag - L2:st + frag].translate(comp)[::-1])
        out = []
        for m in mates:
            a = np.frombuffer(m, np.uint8).copy()
            e = rng.random(len(a)) < err
            a[e] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(e.sum()))]
            a[rng.random(len(a)) < n_rate] = ord("N")
            if rng.random() < lower_rate:
                a = np.frombuffer(a.tobytes().lower(), np.uint8).copy()
            out.append(a.tobytes())
        reads.append(out)
    # hand-made edge cases
    reads.append([b"A" * read_len] + ([b"T" * read_len] if paired else []))            # one distinct k-mer, many windows
    reads.append([b"ACGT" * (read_len // 4)] + ([b"ACGT" * (read_len // 4)] if paired else []))  # 4 distinct windows repeated
    reads.append([b"N" * read_len] + ([b"N" * 7] if paired else []))                   # no valid window
    reads.append([b"ACG"] + ([genomes[0][:read_len]] if paired else []))                # first mate too short
    if paired:
        reads.append([genomes[0][100:100 + read_len], b"ACGTA"])                        # second mate shorter than k (Q8)
        reads.append([genomes[1][500:500 + read_len], genomes[1][500:500 + read_len]])  # identical mates: all duplicates
    return reads


@pytest.fixture(scope="module")
def phage(orc, hip_ctx, tmp_path_factory):
    d = tmp_path_factory.mktemp("phage_gpu")
    tsv = d / "refs.tsv"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    oix = orc.Index.build_single(str(tsv), 750_000, 4, 27)           # test.sh:3 parameters
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    hx = to_hip_index(hip_ctx, oix)
    yield oix, hx, genomes
    hx.close()


def check(oix, hx, reads, d, S):
    bases, seq_off, read_seq0 = pack_reads(reads)
    want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
    got = hx.readid_count(bases, seq_off, read_seq0, d, S)
    assert np.array_equal(want[2], got[2]), "status"
    assert np.array_equal(want[1], got[1]), "n_kmers"
    bad = np.flatnonzero((want[0] != got[0]).any(axis=1))
    assert len(bad) == 0, (bad[:5], want[0][bad[:1]], got[0][bad[:1]])
    # the sparse form of the same call: non-zero entries per read, ascending colour
    rs, col, cnt, nk, st = hx.readid_count_sparse(bases, seq_off, read_seq0, d, S)
    assert np.array_equal(nk, want[1]) and np.array_equal(st, want[2])
    rows, cols = np.nonzero(want[0])
    assert np.array_equal(rs, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=len(want[0])))]).astype(np.uint64))
    assert np.array_equal(col, cols.astype(np.uint32)) and np.array_equal(cnt, want[0][rows, cols])
    return want


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("d,S", [(1, 3), (1, 0), (10, 3), (1, 1), (3, 1000), (7, 0)])
def test_readid_phage(orc, phage, paired, d, S):
    oix, hx, genomes = phage
    rng = np.random.default_rng(100 * d + S + paired)
    reads = sample_reads(orc, rng, genomes, 600, 150, paired)
    rep, nk, st = check(oix, hx, reads, d, S)
    assert st.sum() >= 1 and nk.max() >= 100 // d
    assert rep[:, :4].sum() > 0 and rep[:, 4].sum() > 0      # both real hits and "absent row" stops occur


@pytest.mark.parametrize("n_colors,n_hash,k,m", [(46, 4, 31, 200_003), (256, 2, 21, 1 << 18), (300, 3, 21, 50_021),
                                                 (1024, 4, 31, 20_011), (64, 1, 15, 9_973), (4096, 2, 25, 3_001),
                                                 (10_000, 2, 21, 2_003), (200, 6, 40, 30_011), (33, 5, 33, 9_001)])
def test_readid_layouts(orc, hip_ctx, n_colors, n_hash, k, m):
    rng = np.random.default_rng(n_colors + k)
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.05, zero_row_frac=0.02)
    genomes = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)].tobytes() for _ in range(6)]
    for gi, g in enumerate(genomes):                       # genome gi -> colours gi and n_colors-1-gi
        km = orc.Kmers(k)
        km.kmerize_vector(g, 1)
        for key in km.keys():
            oix.insert(gi, key.tobytes())
            oix.insert(n_colors - 1 - gi, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    for paired, d, S in ((True, 1, 3), (False, 1, 0), (True, 5, 2)):
        reads = sample_reads(orc, rng, genomes, 300, 100 if k < 31 else 150, paired)
        rep, nk, st = check(oix, hx, reads, d, S)
        assert rep[:, :n_colors].sum() > 0
    hx.close()


def test_readid_long_reads_and_empty(orc, phage):
    oix, hx, genomes = phage
    rng = np.random.default_rng(5)
    reads = [[g[s:s + L]] for g in genomes for s, L in ((0, 1000), (2000, 2500), (5000, 27), (7000, 28))]
    check(oix, hx, reads, 1, 3)
    check(oix, hx, reads, 4, 0)
    rep, nk, st = hx.readid_count(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint64))
    assert rep.shape == (0, 5)


@pytest.mark.parametrize("d,S", [(1, 3), (1, 0), (7, 2)])
def test_readid_very_long_reads_sort_path(orc, phage, d, S):
    """Reads whose k-mer set cannot live in one wave's LDS (whole phage genomes, a 150 kb chimera, repeats) take the
    sort-based path; mixed with short reads in the same batch."""
    oix, hx, genomes = phage
    rng = np.random.default_rng(d * 10 + S)
    rnd = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50_000)].tobytes()
    chimera = genomes[0] + rnd[:20_000] + genomes[1][:30_000] + genomes[0][:5_000] + b"N" * 40 + genomes[3]
    reads = [[genomes[2]], [chimera], [genomes[1][:40_000], genomes[1][10_000:45_000]],     # a long "pair" with shared k-mers
             [genomes[0][100:250]], [b"ACG"], [b"A" * 30_000], [genomes[3][:20_000] * 3]]
    check(oix, hx, reads, d, S)
    # lower-case and mixed-case long reads: their case is kept, so the sort-based path keys on byte strings
    mixed = bytearray(genomes[2] + genomes[1])
    for a in range(0, len(mixed), 997):
        mixed[a:a + 300] = bytes(mixed[a:a + 300]).lower()
    check(oix, hx, [[(genomes[2] + genomes[1]).lower()], [bytes(mixed)], [genomes[0][:300]], [bytes(mixed[:50_000]), genomes[2][:9_000].lower()]], d, S)


@pytest.mark.parametrize("k,n_hash", [(35, 2), (64, 3), (128, 1), (33, 4)])
def test_readid_long_reads_k_above_32(orc, hip_ctx, k, n_hash):
    """k > 32 and a read too long for the LDS kernel: byte-string keys through the multi-word sort"""
    rng = np.random.default_rng(k)
    n_colors, m = 40, 60_013
    g = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 70_000)].tobytes() for _ in range(2)]
    oix = orc.Index(m, n_hash, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"acc{c:03d}", 5000)
    for gi, gg in enumerate(g):
        km = orc.Kmers(k)
        km.kmerize_vector(gg[:30_000], 1)
        for key in km.keys():
            oix.insert(3 + 7 * gi, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    low = bytearray(g[1][:60_000])
    low[1000:1500] = bytes(low[1000:1500]).lower()
    reads = [[g[0]], [g[0][10_000:65_000], g[1][:50_000]], [bytes(low)], [g[0][:200]], [g[1][:k - 1]], [g[0][:20_000] * 3 + b"N" + g[1][:30_000]]]
    for d, S in ((1, 3), (1, 0), (5, 2)):
        rep, nk, st = check(oix, hx, reads, d, S)
        assert rep[0, 3] > 1000 and st[4] == 1
    hx.close()


def test_readid_mixed_batch_routes_per_read(orc, phage):
    """One batch with short reads (LDS kernel), 2-6 kb reads (two waves per workgroup still fit / no longer fit) and whole
    genomes (sort-based lists): every read goes to its own path and the rows come back in input order."""
    oix, hx, genomes = phage
    rng = np.random.default_rng(77)
    reads = []
    for i in range(400):
        g = genomes[i % 4]
        L = int(rng.choice([30, 150, 151, 700, 2000, 2600, 3500, 6000, 12_000]))
        st = int(rng.integers(0, len(g) - L))
        r = g[st:st + L]
        if i % 37 == 0:
            r = r.lower()
        reads.append([r] if i % 3 else [r, g[st:st + min(L, 500)]])
    reads += [[genomes[1]], [b"ACG"], [genomes[0][:5000], genomes[2][:9000]], [b"N" * 5000]]
    for d, S in ((1, 3), (1, 0), (4, 2)):
        rep, nk, st = check(oix, hx, reads, d, S)
        assert st[-3] == 1 and rep[-4, 1] > 50 and nk[-4] > 5_000


def test_readid_wide_rows_long_reads(orc, hip_ctx):
    """more than 8192 colours AND reads too long for the LDS kernel: k_readid_list over 1-KiB row steps"""
    rng = np.random.default_rng(99)
    n_colors, n_hash, k, m = 9000, 2, 21, 4_001
    oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.01, zero_row_frac=0.02)
    g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 60_000)].tobytes()
    km = orc.Kmers(k)
    km.kmerize_vector(g[:50_000], 1)
    for key in km.keys():
        oix.insert(7, key.tobytes())
        oix.insert(8999, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    reads = [[g[:45_000]], [g[30_000:60_000], g[:20_000]], [g[100:250]], [b"AC"], [g[5_000:5_400]], [g[:3000]]]   # mixed routing
    for d, S in ((1, 3), (1, 0), (3, 1)):
        rep, nk, st = check(oix, hx, reads, d, S)
        assert rep[0, 7] > 1000 and rep[0, 8999] == rep[0, 7]
    hx.close()


def test_readid_dev_understated_maxima_are_flagged(orc, phage):
    """cid_readid_count_dev sizes the kernel's LDS by the caller's maxima: a read beyond them is left alone with status 3
    (its neighbours' rows stay right) instead of overrunning another wave's LDS."""
    import torch
    oix, hx, genomes = phage
    rng = np.random.default_rng(5)
    reads = [[genomes[i % 4][s:s + 150]] for i, s in enumerate(rng.integers(0, 30000, 64))]
    reads[10] = [genomes[0][1000:1400]]             # 400 bases: beyond the stated 150
    reads[40] = [genomes[1][2000:2151]]             # one base / one window too many
    bases, seq_off, read_seq0 = pack_reads(reads)
    want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3)
    db = torch.from_numpy(bases.copy()).cuda()
    dso = torch.from_numpy(seq_off.astype(np.int64)).cuda()
    dr0 = torch.from_numpy(read_seq0.astype(np.int64)).cuda()
    C = oix.n_colors
    rep = torch.full((len(reads), C + 1), 7, dtype=torch.int32, device="cuda")
    nk = torch.full((len(reads),), 7, dtype=torch.int32, device="cuda")
    st = torch.full((len(reads),), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    hx.readid_count_dev(db.data_ptr(), dso.data_ptr(), dr0.data_ptr(), len(reads), 1, 3, 150, 150 - 27 + 1, rep.data_ptr(), nk.data_ptr(),
                        st.data_ptr())
    hx.ctx.synchronize()
    st, nk, rep = st.cpu().numpy(), nk.cpu().numpy().view(np.uint32), rep.cpu().numpy().view(np.uint32)
    over = np.zeros(len(reads), bool)
    over[[10, 40]] = True
    assert np.array_equal(st[over], [3, 3]) and not nk[over].any() and not rep[over].any()
    assert np.array_equal(st[~over], want[2][~over]) and np.array_equal(nk[~over], want[1][~over])
    assert np.array_equal(rep[~over], want[0][~over])


def test_readid_dense_report_is_sliced(orc, phage):
    """The dense host call works through a batch in slices whose report rows fit a device scratch budget
    (CID_DENSE_REPORT_BYTES; 2 GiB by default): same rows as one launch."""
    import subprocess
    import sys
    oix, hx, genomes = phage
    rng = np.random.default_rng(6)
    reads = sample_reads(orc, rng, genomes, 300, 120, True)
    bases, seq_off, read_seq0 = pack_reads(reads)
    want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3)
    np.savez("/tmp/_slice_case.npz", bases=bases, seq_off=seq_off, read_seq0=read_seq0, rep=want[0], nk=want[1], st=want[2], rows=oix.rows())
    # the budget is read once per process: run the sliced call in a child with 37 rows per slice
    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(HERE)!r})
import colorid_amd
z = np.load('/tmp/_slice_case.npz')
ctx = colorid_amd.Context(0)
hx = colorid_amd.Index(ctx, {oix.m}, {oix.n_hash}, {oix.k}, {oix.n_colors})
hx.put_dense(z['rows']); hx.finalize()
rep, nk, st = hx.readid_count(z['bases'], z['seq_off'], z['read_seq0'], 1, 3)
assert np.array_equal(rep, z['rep']) and np.array_equal(nk, z['nk']) and np.array_equal(st, z['st'])
print('sliced ok')
"""
    env = dict(os.environ, CID_DENSE_REPORT_BYTES=str(37 * (oix.n_colors + 1) * 4))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "sliced ok" in p.stdout, p.stderr[-2000:]


def test_readid_more_than_two_mates(orc, phage):
    """A read is any number of sequences (read_seq0): the window sequence runs over all of them in order — three and four mates,
    short ones in between, chunks of 64 windows straddling every mate boundary."""
    oix, hx, genomes = phage
    rng = np.random.default_rng(11)
    reads = []
    for i in range(120):
        g = genomes[i % 4]
        n_m = int(rng.integers(3, 5))
        mates = []
        for j in range(n_m):
            L = int(rng.choice([5, 26, 27, 28, 60, 91, 150]))
            st = int(rng.integers(0, len(g) - 200))
            mates.append(g[st:st + L])
        if len(mates[0]) < 27:
            mates[0] = g[100:190]                      # keep most reads classifiable (too_short looks at the first mate only)
        reads.append(mates)
    reads.append([genomes[0][:150], b"", genomes[0][150:300], b"ACGT"])
    reads.append([b"ACG", genomes[1][:150], genomes[1][200:350]])       # first mate too short -> too_short whatever follows
    for d, S in ((1, 3), (1, 0), (7, 2)):
        check(oix, hx, reads, d, S)


@pytest.mark.parametrize("packed", [0, 1])
@pytest.mark.parametrize("k,n_colors", [(21, 256), (27, 100), (9, 300), (31, 256), (29, 64), (32, 130)])
def test_readid_set_slot_layouts_are_bit_exact(orc, hip_ctx, packed, k, n_colors):
    """k_readid's per-read k-mer set with 12-byte slots and with one u64 per slot (code << idx_bits | first window index; taken for
    read pairs whose 12-byte table would cost the sixth wave per SIMD; at k = 28..32, where the code leaves no room for the index, one u32 per slot:
    the earliest position of a window holding the k-mer, the code read back from the packed bases on a probe; cid_ctx_tune "readid_packed_table" = 0
    keeps the 12-byte slots): paired 150-bp reads with
    repeats (the same k-mer in both mates: the earlier window must win), N runs, short mates — against the oracle."""
    from colorid_amd._lib import check as cid_check
    rng = np.random.default_rng(k * 100 + n_colors + packed)
    genomes = [bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)) for _ in range(6)]
    oix = orc.Index(50_021, 2, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"a{c}", 500)
    for gi, gen in enumerate(genomes):
        km = orc.Kmers(k)
        km.kmerize_vector(gen, 1)
        for key in km.keys():
            oix.insert(gi, key.tobytes())
            oix.insert(n_colors - 1 - gi, key.tobytes())
    reads = sample_reads(orc, rng, genomes, 400, 150, True)
    g0 = genomes[0]
    reads.append([g0[100:250], g0[100:250]])                       # both mates identical: every k-mer twice
    reads.append([g0[100:250], g0[180:330]])                       # overlapping mates
    reads.append([(g0[300:320] * 8)[:150], (g0[300:320] * 8)[:150]])   # a 20-base period: few distinct k-mers, many windows
    hx = to_hip_index(hip_ctx, oix)
    hip_ctx.tune("readid_packed_table", packed)
    try:
        for d, S in ((1, 3), (1, 0), (2, 5)):
            check(oix, hx, reads, d, S)
    finally:
        hip_ctx.tune("readid_packed_table", 1)
    hx.close()


@pytest.mark.parametrize("k,n_colors,read_len", [(21, 256, 150), (9, 300, 150), (31, 64, 100), (27, 100, 150)])
def test_readid_single_end_edge_reads(orc, hip_ctx, k, n_colors, read_len):
    """Single-end batches through k_readid's aligned 16-byte base loads (reads start at every offset mod 16): neighbours that share
    every k-mer (the sets are per read), reads that are too short / lower-case / all N / of several sequences / one window long,
    a read whose k-mers are in no accession (stops at its first), an odd read count, the batch shifted by one read."""
    rng = np.random.default_rng(k * 1000 + n_colors)
    genomes = [bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)) for _ in range(6)]
    oix = orc.Index(50_021, 2, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"a{c}", 500)
    for gi, gen in enumerate(genomes[:5]):          # genome 5 is in no accession: its reads stop at their first k-mer
        km = orc.Kmers(k)
        km.kmerize_vector(gen, 1)
        for key in km.keys():
            oix.insert(gi, key.tobytes())
            oix.insert(n_colors - 1 - gi, key.tobytes())
    L = read_len
    reads = sample_reads(orc, rng, genomes, 300, L, False, lower_rate=0.05)
    g0, g5 = genomes[0], genomes[5]
    reads += [[g0[100:100 + L]], [g0[100:100 + L]]]                     # the same read twice in a row
    reads += [[g0[200:200 + L]], [g5[200:200 + L]]]                     # a hit beside a miss, both orders
    reads += [[g5[300:300 + L]], [g0[300:300 + L]]]
    reads += [[g0[400:400 + L]], [b"ACG"]]                              # too short second / first
    reads += [[b"AC"], [g0[500:500 + L]]]
    reads += [[g0[600:600 + L].lower()], [g0[600:600 + L]]]             # lower-case first / second
    reads += [[g0[700:700 + L]], [g0[700:700 + L - 1] + b"a"]]
    reads += [[g0[800:800 + L], g0[900:900 + L]], [g0[800:800 + L]]]   # a pair beside a single read
    reads += [[g0[1000:1000 + L]], [b"N" * L]]
    reads += [[g0[1100:1100 + L]], [g0[1100:1100 + k]]]                 # one window
    reads += [[(g0[1200:1217] * 10)[:L]], [(g0[1200:1217] * 10)[:L]]]   # 17-base period: few distinct k-mers, many windows
    reads += [[g0[1300:1300 + L]]]
    if len(reads) % 2 == 0:
        reads += [[g0[1400:1400 + L]]]                                  # odd count: the last read has no partner
    hx = to_hip_index(hip_ctx, oix)
    for d, S in ((1, 3), (1, 0), (2, 5), (1, 200)):
        check(oix, hx, reads, d, S)
    check(oix, hx, reads[1:], 1, 3)                                     # every read at another offset
    hx.close()


def test_readid_grid_cut_does_not_change_rows(orc, phage, hip_ctx):
    """cid_ctx_tune "readid_blocks_per_cu": however a batch is cut into workgroups (one read per wave ... the whole batch in a few
    workgroups), the rows are the same — the offsets a wave fetches 64 reads at a time and the bases it asks for one read ahead must
    follow the cut.  Values outside 1..4096 and unknown names are refused."""
    import colorid_amd
    oix, hx, genomes = phage
    rng = np.random.default_rng(5)
    reads = sample_reads(orc, rng, genomes, 700, 150, False) + sample_reads(orc, rng, genomes, 300, 150, True)
    try:
        for bpc in (1, 3, 64, 4096):
            hip_ctx.tune("readid_blocks_per_cu", bpc)
            check(oix, hx, reads, 1, 3)
        for bad in (0, 4097, -1):
            with pytest.raises(colorid_amd.CidError):
                hip_ctx.tune("readid_blocks_per_cu", bad)
        with pytest.raises(colorid_amd.CidError):
            hip_ctx.tune("no_such_switch", 1)
    finally:
        hip_ctx.tune("readid_blocks_per_cu", 64)


def test_readid_more_than_64_reads_per_wave(orc, hip_ctx):
    """A wave fetches its reads' offsets 64 reads at a time (one lane each) and the next read's bases one read ahead: 1 800-base reads
    (n = 2, k = 21: 66 KB of LDS per wave) leave two waves per workgroup, and 40 000 of them cut into one workgroup per CU give each
    wave 78 reads — the second fetch, and the read-ahead across it.  Rows equal those of the finest cut (one fetch per wave), and the
    oracle's on a sample."""
    rng = np.random.default_rng(12)
    k, n_colors = 21, 70
    genomes = [bytes(rng.choice(list(b"ACGT"), size=30_000).astype(np.uint8)) for _ in range(3)]
    oix = orc.Index(200_003, 2, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"a{c}", 500)
    for gi, gen in enumerate(genomes[:2]):
        km = orc.Kmers(k)
        km.kmerize_vector(gen, 1)
        for key in km.keys():
            oix.insert(gi, key.tobytes())
            oix.insert(n_colors - 1 - gi, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    g = b"".join(genomes)
    n = 40_000
    starts = rng.integers(0, len(g) - 1800, n)
    lens = np.where(rng.random(n) < 0.7, 1800, rng.integers(k, 1800, n))
    reads = [[g[int(s):int(s) + int(L)]] for s, L in zip(starts, lens)]
    bases, seq_off, read_seq0 = pack_reads(reads)
    try:
        hip_ctx.tune("readid_blocks_per_cu", 4096)
        fine = hx.readid_count(bases, seq_off, read_seq0, 1, 3)
        hip_ctx.tune("readid_blocks_per_cu", 1)
        coarse = hx.readid_count(bases, seq_off, read_seq0, 1, 3)
    finally:
        hip_ctx.tune("readid_blocks_per_cu", 64)
    for a, b in zip(fine, coarse):
        assert np.array_equal(a, b)
    sample = sorted(rng.choice(n, 600, replace=False).tolist())
    sb, so, sr = pack_reads([reads[i] for i in sample])
    want = oix.readid_counts(sb, so, sr, 1, 3)
    assert np.array_equal(want[0], coarse[0][sample]) and np.array_equal(want[1], coarse[1][sample]) and np.array_equal(want[2], coarse[2][sample])
    assert int(want[1].max()) > 1500          # (the reads were not cut short on the way)
    hx.close()
