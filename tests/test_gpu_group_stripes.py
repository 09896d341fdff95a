"""Colour stripes over the ranks of a cid_group (cid_group_stripes_*): rank r holds the colours [base_r, base_{r+1})
of every row, every rank sees the whole query, one exchange per call (packed per-k-mer facts summed; zero-row masks ANDed).  The
results must equal the oracle's on the unsplit index — proportional search (host k-mers and device-resident sets), perfect search,
read_id — for 1..4 ranks sharing the one GPU (peer copies + add / AND kernels; RCCL would refuse repeated device ids), stripes
narrower and wider than 8192 colours, colour counts that do not fill the last word, and a .bxi's own records as the input."""
import os
import struct

import numpy as np
import pytest

from test_gpu_readid import pack_reads, sample_reads
from util import plant, random_index, random_kmers

pytestmark = pytest.mark.gpu


def _striped(g, oix, via_records=False):
    st = g.stripes(oix.m, oix.n_hash, oix.k, oix.n_colors)
    rows = oix.rows()                                  # dense m x W32; the file / put_rows carry the non-zero rows only
    w32 = (oix.n_colors + 31) // 32
    ids = np.nonzero(rows.any(axis=1))[0].astype(np.uint64)
    words = np.ascontiguousarray(rows[ids.astype(np.int64)], np.uint32)
    if via_records:   # bincode (usize, BitVec<u32>) records as a .bxi holds them
        rec = b"".join(struct.pack("<QQ", int(i), w32) + w.astype("<u4").tobytes() + struct.pack("<Q", oix.n_colors) for i, w in zip(ids, words))
        st.put_records(rec, len(ids))
    else:
        st.put_rows(ids, words)
    return st.finalize()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0], [0, 0, 0, 0]])
@pytest.mark.parametrize("n_colors,k,via_records", [(512, 31, False), (300, 27, True), (1100, 21, False), (20000, 31, True), (256, 40, False)])
def test_striped_group_search_equals_oracle(orc, devices, n_colors, k, via_records):
    import colorid_amd
    if (n_colors + 63) // 64 < len(devices):
        pytest.skip("fewer 64-colour words than ranks")
    rng = np.random.default_rng(n_colors * 3 + len(devices))
    m = 30_011 if n_colors < 2000 else 3001
    oix = random_index(orc, rng, m, 3, k, n_colors, density=0.12 if n_colors < 2000 else 0.004, zero_row_frac=0.05)
    kmers = random_kmers(rng, 9001, k)
    plant(oix, rng, kmers, frac=0.7)
    freq = rng.integers(1, 30, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    g = colorid_amd.Group(devices)
    st = _striped(g, oix, via_records)
    assert st.base[0] == 0 and st.base[-1] == n_colors and all(b % 64 == 0 for b in st.base[:-1])
    got = st.search_count(kmers, freq)
    for w, x in zip(want, got):
        assert np.array_equal(w, x)
    for nk in (1, 0):
        w = oix.search_count(kmers[:nk], freq[:nk].astype(np.uint64))
        x = st.search_count(kmers[:nk], freq[:nk])
        assert all(np.array_equal(a, b) for a, b in zip(w, x))
    g.close()
    # perfect search: a subset planted in the first and the last colour (two different stripes), then sets with an absent row
    sub = kmers[:500].copy()
    for km in sub:
        oix.insert(0, km.tobytes())
        oix.insert(n_colors - 1, km.tobytes())
    g = colorid_amd.Group(devices)
    st = _striped(g, oix, via_records)
    hit = False
    for sel in (sub, kmers[:3000], kmers[:1]):
        pw, pm = oix.search_perfect(sel)
        gw, gm = st.search_perfect(sel)
        assert gm == pm and np.array_equal(gw, pw)
        hit = hit or (not pm and pw.any())
    assert hit
    # device-resident k-mer set on rank 0 (2-bit codes; byte strings for k > 32)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=2500).astype(np.uint8)) for _ in range(4)] + [sub[:200].tobytes()]
    ks = colorid_amd.KmerSet(g.ctxs[0], k)
    ks.add_seqs(seqs, 0)
    ks.finalize()
    km, cnt = ks.download()
    w = oix.search_count(km, cnt.astype(np.uint64))
    x = st.search_count_set(ks)
    assert all(np.array_equal(a, b) for a, b in zip(w, x))
    hits, nu, sf, md = st.search_count_set_report(ks)          # the report finished on rank 0's device (mode per colour)
    assert np.array_equal(hits, w[0]) and np.array_equal(nu, w[1]) and np.array_equal(sf, w[2])
    assert np.array_equal(md, orc.unique_modes(w[3], cnt.astype(np.uint64), n_colors))
    pw, pm = oix.search_perfect(km)
    gw, gm = st.search_perfect_set(ks)
    assert gm == pm and np.array_equal(gw, pw)
    g.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
@pytest.mark.parametrize("n_colors", [384, 200])
def test_striped_group_readid_equals_oracle(orc, devices, n_colors):
    import colorid_amd
    rng = np.random.default_rng(n_colors + len(devices))
    k, m = 21, 60_013
    genomes = [bytes(rng.choice(list(b"ACGT"), size=6000).astype(np.uint8)) for _ in range(12)]
    oix = orc.Index(m, 2, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"acc{c}", 1000)
    kms = orc.Kmers(k)
    for gi, gen in enumerate(genomes):
        kms = orc.Kmers(k)
        kms.kmerize_vector(gen, 1)
        for key in kms.keys():
            for c in (gi, gi + 100, n_colors - 1 - gi):     # colours in different stripes share k-mers
                oix.insert(c, key.tobytes())
    g = colorid_amd.Group(devices)
    st = _striped(g, oix)
    for paired, n_reads in ((True, 301), (False, 200), (False, 0)):
        reads = sample_reads(orc, rng, genomes, n_reads, 150, paired) if n_reads else []
        bases, seq_off, read_seq0 = pack_reads(reads)
        for d, S in ((1, 3), (1, 0), (7, 2)):
            want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
            rs, col, cnt, nk, stt = st.readid_count_sparse(bases, seq_off, read_seq0, d, S)
            assert np.array_equal(nk, want[1]) and np.array_equal(stt, want[2])
            rows, cols = np.nonzero(want[0])
            assert np.array_equal(rs, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=len(want[0])))]).astype(np.uint64))
            assert np.array_equal(col, cols.astype(np.uint32)) and np.array_equal(cnt, want[0][rows, cols])
            if n_reads:
                assert (cols == n_colors).any() and (cols < n_colors).any()      # no-hits entries and real colours both occur
    g.close()


@pytest.mark.parametrize("devices,n_colors", [([0, 0], 384), ([0, 0, 0], 17_000)])
def test_striped_group_readid_long_reads(orc, devices, n_colors):
    """cid_group_stripes_readid_count_sparse with reads too long for a wave's LDS (sort-based path), mixed with short ones; the
    17 000-colour case gives two ranks a stripe of more than 8192 colours (wide-row kernels) — against the oracle on the unsplit index."""
    import colorid_amd
    rng = np.random.default_rng(n_colors)
    k, m = 21, 60_013
    genomes = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000)].tobytes() for _ in range(3)]
    oix = orc.Index(m, 2, k, n_colors)
    for c in range(n_colors):
        oix.set_color(c, f"acc{c}", 1000)
    for gi, gen in enumerate(genomes):
        kms = orc.Kmers(k)
        kms.kmerize_vector(gen[:15_000], 1)
        for key in kms.keys():
            for c in (gi, n_colors // 2 + gi, n_colors - 1 - gi):
                oix.insert(c, key.tobytes())
    g = colorid_amd.Group(devices)
    st = _striped(g, oix)
    reads = [[genomes[0]], [genomes[1][:9_000], genomes[2][2_000:12_000]], [genomes[0][50:200]], [b"AC"], [genomes[2][:3_000].lower()],
             [genomes[1][:4_000] + b"N" * 25 + genomes[0][:6_000]]]
    reads += sample_reads(orc, rng, genomes, 40, 150, False)
    bases, seq_off, read_seq0 = pack_reads(reads)
    for d, S in ((1, 3), (1, 0), (6, 2)):
        want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
        rs, col, cnt, nk, stt = st.readid_count_sparse(bases, seq_off, read_seq0, d, S)
        assert np.array_equal(nk, want[1]) and np.array_equal(stt, want[2])
        rows, cols = np.nonzero(want[0])
        assert np.array_equal(rs, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=len(want[0])))]).astype(np.uint64))
        assert np.array_equal(col, cols.astype(np.uint32)) and np.array_equal(cnt, want[0][rows, cols])
        assert want[0][0, 0] > 1000 and (cols == n_colors).any()
    g.close()


def test_striped_group_refuses_misuse(orc):
    import ctypes as C

    import colorid_amd
    from colorid_amd._lib import vp
    g = colorid_amd.Group([0, 0, 0])
    arr = (vp * 3)()
    assert g.lib.cid_group_stripes_create(g.h, 1000, 2, 21, 128, 0, arr) == -1 and b"fewer than the 3 ranks" in g.lib.cid_last_error()
    st = g.stripes(1000, 2, 21, 640)
    hits = np.zeros(640, np.uint64)
    km = np.frombuffer(b"A" * 21, np.uint8)
    rc = g.lib.cid_group_stripes_search_count(g.h, st.arr, km.ctypes.data_as(vp), None, 1, hits.ctypes.data_as(vp), None, None, None)
    assert rc < 0 and b"not finalized" in g.lib.cid_last_error()
    st.finalize()
    words = np.zeros(20, np.uint32)
    miss = C.c_int(0)
    assert g.lib.cid_group_stripes_search_perfect(g.h, st.arr, km.ctypes.data_as(vp), 0, words.ctypes.data_as(vp), C.byref(miss)) == -1
    # a malformed record (word count of another file) is reported, not written
    rec = struct.pack("<QQ", 5, 3) + b"\0" * 12 + struct.pack("<Q", 96)
    st2 = g.stripes(1000, 2, 21, 640)
    buf = np.frombuffer(rec, np.uint8)
    assert g.lib.cid_group_stripes_put_records(g.h, st2.arr, buf.ctypes.data_as(vp), 1) < 0 and b"malformed" in g.lib.cid_last_error()
    g.close()


def test_cli_placement_striped_equals_single_gpu(orc, tmp_path):
    """`colorid search|read_id --devices 0,0[,0] --placement striped` (the .bxi loaded as colour stripes, one per rank) prints what
    `--device 0` prints: 200 synthetic accessions (4 words of 64 colours), proportional / gene / perfect searches and read_id."""
    import subprocess

    from test_gpu_cli import BANNER, BIN
    from util import synth_fastq_records, write_fastq_gz

    def cli(*args, env=None):
        p = subprocess.run([BIN, *args], capture_output=True, text=True, env=dict(os.environ, COLORID_QUIET="1", **(env or {})))
        assert p.returncode == 0, p.stderr[-3000:]
        assert p.stdout.startswith(BANNER)
        return p.stdout[len(BANNER):], p.stderr

    rng = np.random.default_rng(77)
    genomes = []
    lines = []
    for i in range(200):
        g = bytes(rng.choice(list(b"ACGT"), size=4000).astype(np.uint8))
        if i % 10 == 3:                              # related accessions: shared k-mers, fewer unique hits
            g = genomes[i - 1][:2500] + g[2500:]
        genomes.append(g)
        fa = tmp_path / f"acc{i:03d}.fasta"
        fa.write_text(f">acc{i:03d}\n{g.decode()}\n")
        lines.append(f"acc{i:03d}\t{fa}\n")
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(lines))
    pre = str(tmp_path / "syn")
    cli("build", "-s", "400000", "-n", "3", "-k", "25", "-b", pre, "-r", str(tsv))
    r1 = synth_fastq_records(np.random.default_rng(5), genomes[:40], 3000, 120, mate=0)
    r2 = synth_fastq_records(np.random.default_rng(5), genomes[:40], 3000, 120, mate=1)
    f1, f2 = str(tmp_path / "r_1.fastq.gz"), str(tmp_path / "r_2.fastq.gz")
    write_fastq_gz(f1, r1)
    write_fastq_gz(f2, r2)
    fasta = str(tmp_path / "acc007.fasta")
    cases = {
        "search_pe": ("search", "-b", pre + ".bxi", "-q", f1, "-r", f2, "-f", "1", "-p", "0.01"),
        "search_gene": ("search", "-b", pre + ".bxi", "-q", f1, "-g", "-f", "0", "-p", "0.01"),
        "search_fasta": ("search", "-b", pre + ".bxi", "-q", fasta, "-p", "0.01"),
        "perfect": ("search", "-b", pre + ".bxi", "-q", fasta, "-s"),
        "perfect_mf": ("search", "-b", pre + ".bxi", "-q", fasta, "-s", "-m"),
    }
    for name, args in cases.items():
        base, _ = cli(*args, "--device", "0")
        assert base.strip(), name
        for devs in ("0,0", "0,0,0", "0"):
            out, err = cli(*args, "--devices", devs, "--placement", "striped")
            assert sorted(out.splitlines()) == sorted(base.splitlines()), (name, devs)
            assert "one colour stripe of the index each" in err
        # the RCCL reduction of the per-k-mer facts (ncclAllReduce, u32, K) with one rank: what N distinct GPUs take
        out, err = cli(*args, "--devices", "0", "--placement", "striped", env={"COLORID_REDUCE": "rccl"})
        assert sorted(out.splitlines()) == sorted(base.splitlines()), (name, "rccl")
        assert "with RCCL all-reduce" in err
    for tag, q in (("se", (f1,)), ("pe", (f1, f2))):
        cli("read_id", "-b", pre + ".bxi", "-q", *q, "-n", str(tmp_path / f"one_{tag}"), "-c", "700")
        cli("read_id", "-b", pre + ".bxi", "-q", *q, "-n", str(tmp_path / f"str_{tag}"), "-c", "700", "--devices", "0,0,0", "--placement", "striped")
        assert open(tmp_path / f"one_{tag}_reads.txt").read() == open(tmp_path / f"str_{tag}_reads.txt").read()
        assert open(tmp_path / f"one_{tag}_counts.txt").read() == open(tmp_path / f"str_{tag}_counts.txt").read()
        # the zero-row masks through RCCL (ncclAllGather + local AND on every rank) with one rank: what N distinct GPUs take
        _, err = cli("read_id", "-b", pre + ".bxi", "-q", *q, "-n", str(tmp_path / f"rccl_{tag}"), "-c", "700", "--devices", "0", "--placement", "striped",
                     env={"COLORID_REDUCE": "rccl"})
        assert "with RCCL all-reduce" in err
        assert open(tmp_path / f"one_{tag}_reads.txt").read() == open(tmp_path / f"rccl_{tag}_reads.txt").read()
    # more ranks than 64-colour words: refused with the reason
    p = subprocess.run([BIN, *cases["perfect"], "--devices", "0,0,0,0,0", "--placement", "striped"], capture_output=True, text=True)
    assert p.returncode != 0 and "fewer than the 5 ranks" in p.stderr
