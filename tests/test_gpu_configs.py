"""BASELINE.json's two remaining configurations AT THEIR STATED WORKLOAD (the other configs' shapes live in test_gpu_fullsize.py):

configs[0]  `build` + `search -s` at k = 31, -s 50,000,000, -n 4 on 46 accessions of ~2.9 Mbp, query = one genome (K ~ 2.9 M distinct
            canonical 31-mers): /root/reference/src/perfect_search.rs:6-60 (batch_search), :62-120 (batch_search_mf), through the C++
            CLI and through cid_search_perfect_set, against the oracle run on the index the CLI wrote — plus a doctored index with
            one of the query's rows removed ("No perfect hits!", perfect_search.rs:31-39).
configs[3]  one GPU's share of "100 M reads over 8 GPUs": 12.5 M x 150 bp reads = 1.5 G k-mer windows counted by cid_kmerset in
            >= 8 incremental merges (/root/reference/src/kmer.rs:461-510 the fastq map, :826-837 clean_map's input) and searched
            against m = 50 M x 1024 colours (/root/reference/src/batch_search_pe.rs:24-105).
configs[2]  `build -k 21 -s 30000000 -n 2` on 256 synthetic genomes of 5 Mbp + `read_id` on 1 M synthetic reads (the stated fastq,
            test_data/SRR548019.fastq.gz, is absent from the reference tree), single-end and paired, as a single-stream .fastq.gz and
            as block gzip: /root/reference/src/build.rs:33-130, src/read_id_mt_pe.rs:66-165 (counts), :187-251 (kmer_poll_plus),
            :835-951 / :701-832 (the streams), src/reports.rs:98-120 (_counts.txt) — _reads.txt rows of a 3,000-read sample against
            the oracle reading the .bxi the CLI wrote.
"""
import ctypes as C
import json
import os
import struct
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACGT = np.frombuffer(b"ACGT", np.uint8)


def _record_timings(name, payload):
    """per-phase wall times of the full-size tests, kept when the run has a gpurun_out/ (copied to profiles/ from there)"""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            json.dump(payload, f, indent=1)


def _parse_bxi(path):
    """-> (m, n, k, names[colour], row_ids u64[n_rows], words u32[n_rows, W32], header_bytes, records_bytes_view, tail_bytes)"""
    buf = np.fromfile(path, np.uint8)
    m, n, k, nc = struct.unpack_from("<QQQQ", buf, 0)
    pos = 32
    names = {}
    for _ in range(nc):
        cid, ln = struct.unpack_from("<QQ", buf, pos)
        names[cid] = bytes(buf[pos + 16:pos + 16 + ln]).decode()
        pos += 16 + ln
    (n_rows,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    w32 = (nc + 31) // 32
    rec = 24 + 4 * w32
    recs = buf[pos:pos + n_rows * rec].reshape(n_rows, rec)
    row_ids = recs[:, :8].copy().view("<u8").reshape(-1)
    assert (recs[:, 8:16].copy().view("<u8").reshape(-1) == w32).all()
    words = recs[:, 16:16 + 4 * w32].copy().view("<u4").reshape(n_rows, w32)
    return m, n, k, [names[c] for c in range(nc)], row_ids, words, buf[:pos - 8], recs, buf[pos + n_rows * rec:]


def test_config_a_build_and_perfect_search_at_full_size(orc, hip_ctx, tmp_path):
    import colorid_amd
    from test_gpu_cli import BANNER, BIN
    import subprocess

    def cli(*args):
        p = subprocess.run([BIN, *args], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        assert p.stdout.startswith(BANNER)
        return p.stdout[len(BANNER):], p.stderr

    t = {"t0": time.perf_counter()}

    def lap(name):
        now = time.perf_counter()
        t[name] = round((now - t["t0"]) * 1e3)
        t["t0"] = now

    n_acc, m, n, k, Lg = 46, 50_000_000, 4, 31, 2_900_000
    rng = np.random.default_rng(2024)
    genomes = [ACGT[rng.integers(0, 4, Lg)].tobytes() for _ in range(n_acc)]
    genomes[5] = genomes[17]                                   # the query genome sits in two colours (as EGD-e does in the reference's list)
    near = np.frombuffer(genomes[17], np.uint8).copy()         # and a close relative in a third: most k-mers shared, no perfect hit
    sub = rng.random(Lg) < 0.001
    near[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    genomes[30] = near.tobytes()
    lines = []
    for i, g in enumerate(genomes):
        fa = tmp_path / f"acc{i:02d}.fasta"
        fa.write_bytes(b">acc%02d synthetic\n" % i + g + b"\n")
        lines.append(f"acc{i:02d}\t{fa}\n")
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(reversed(lines)))                  # colours are ranks of the sorted names, whatever the list order (build.rs:102-113)
    lap("make_genomes_ms")
    pre = str(tmp_path / "cfg0")
    cli("build", "-k", str(k), "-s", str(m), "-n", str(n), "-b", pre, "-r", str(tsv))
    bxi = pre + ".bxi"
    lap("cli_build_ms")
    # the query: the genome as one record (-s), and cut into two records (-s -m)
    q = tmp_path / "query.fasta"
    q.write_bytes(b">chromosome\n" + genomes[17] + b"\n")
    qm = tmp_path / "query_mf.fasta"
    qm.write_bytes(b">part one\n" + genomes[17][:1_700_000] + b"\n>part two\n" + genomes[17][1_700_000 - 30:] + b"\n>short\nACGT\n")
    out_s, err_s = cli("search", "-b", bxi, "-q", str(q), "-s")
    lap("cli_search_s_ms")
    out_m, err_m = cli("search", "-b", bxi, "-q", str(qm), "-s", "-m")
    lap("cli_search_s_m_ms")
    # the oracle on the index the CLI wrote (every row of the file, not rows read back from the GPU)
    fm, fn, fk, names, row_ids, words, head, recs, tail = _parse_bxi(bxi)
    assert (fm, fn, fk) == (m, n, k) and names == [f"acc{i:02d}" for i in range(n_acc)]
    assert len(row_ids) > 0.99 * m and (np.diff(row_ids.astype(np.int64)) > 0).all()
    oix = orc.Index(m, n, k, n_acc)
    oix.rows()[row_ids.astype(np.int64)] = words
    for c, nm in enumerate(names):
        oix.set_color(c, nm, 1)
    km = orc.Kmers(k)
    km.kmerize_vector(genomes[17], 1)
    K = len(km)
    assert 2_850_000 < K <= Lg - k + 1
    keys = km.keys()
    pw, pm = oix.search_perfect(keys)
    lap("oracle_ms")
    assert not pm
    hit_colours = [c for c in range(n_acc) if pw[c // 32] >> (c % 32) & 1]
    assert hit_colours == [5, 17]                              # (no Bloom false positive survives 11.6 M rows)
    assert out_s.splitlines() == [f"{q}\t{names[c]}\t{K}\t1.00" for c in hit_colours]
    assert f"{K} kmers in query" in err_s and "2 hits" in err_s
    # -s -m: one search per record, k-mers by kmerize_string (kmer.rs:271-299), label = the header line minus '>'
    labels, seqs = orc.read_fasta_mf(str(qm))
    want = []
    for lab, s in zip(labels, seqs):
        kr = orc.Kmers(k)
        if kr.kmerize_string(s) != 0:
            want.append(f"Warning! no kmers in query '{lab.decode()}'; maybe your kmer length is larger than your query length?")
            continue
        w, miss = oix.search_perfect(kr.keys())
        if not miss:
            want += [f"{lab.decode()}\t{names[c]}\t{len(kr)}\t1.00" for c in range(n_acc) if w[c // 32] >> (c % 32) & 1]
    assert out_m.splitlines() == want and len(want) == 5
    lap("oracle_mf_ms")
    # the same through the ABI: the file's records -> cid_index_put_records, the query counted on the GPU, cid_search_perfect_set
    hx = colorid_amd.Index(hip_ctx, m, n, k, n_acc)
    step = 8_000_000
    for r0 in range(0, len(recs), step):
        hx.put_records(recs[r0:r0 + step].tobytes())
    hx.finalize()
    ks = colorid_amd.KmerSet(hip_ctx, k)
    ks.add_seqs([genomes[17]], 0)
    assert ks.finalize() == K
    gw, gm = ks.search_perfect(hx)
    assert not gm and np.array_equal(gw, pw)
    hw, hm = hx.search_perfect(keys)                           # and the host-k-mer form (cid_search_perfect) on the oracle's own keys
    assert not hm and np.array_equal(hw, pw)
    hits, nu, sf, uc = ks.search_count(hx)                     # proportional search of the same query: both colours see every k-mer
    assert hits[5] == K and hits[17] == K and hits[30] > 0.9 * K and hits[30] < K and nu.sum() == (uc != 0xFFFFFFFF).sum()
    lap("abi_ms")
    # a doctored index: ONE row of one query k-mer removed -> a row is absent -> no rows, "No perfect hits!" (perfect_search.rs:31-39)
    victim = orc.xxh3(keys[K // 2].tobytes(), 2) % m
    at = int(np.searchsorted(row_ids, victim))
    assert row_ids[at] == victim
    oix.rows()[victim] = 0
    dw, dm = oix.search_perfect(keys)
    assert dm and not dw.any()
    hx2 = colorid_amd.Index(hip_ctx, m, n, k, n_acc)
    for r0 in range(0, len(recs), step):
        lo, hi = r0, min(len(recs), r0 + step)
        blk = recs[lo:hi] if not (lo <= at < hi) else np.concatenate([recs[lo:at], recs[at + 1:hi]])
        hx2.put_records(blk.tobytes())
    hx2.finalize()
    gw2, gm2 = ks.search_perfect(hx2)
    assert gm2 and not gw2.any()
    other = np.r_[0:100_000, K // 2 + 1:K // 2 + 100_001]      # k-mers around the removed row: the oracle and the GPU agree on them too
    ow, om = oix.search_perfect(keys[other])
    xw, xm = hx2.search_perfect(keys[other])
    assert om == xm and np.array_equal(ow, xw)
    hx2.close()
    doctored = str(tmp_path / "doctored.bxi")
    with open(doctored, "wb") as f:
        f.write(head.tobytes())
        f.write(struct.pack("<Q", len(recs) - 1))
        f.write(recs[:at].tobytes())
        f.write(recs[at + 1:].tobytes())
        f.write(tail.tobytes())
    out_d, err_d = cli("search", "-b", doctored, "-q", str(q), "-s")
    assert out_d == "" and "No perfect hits!" in err_d
    lap("doctored_ms")
    ks.close()
    hx.close()
    t.pop("t0")
    _record_timings("r03_config_a_full.json", {"config": "configs[0]: build + search -s, k=31 m=50M n=4, 46 accessions x 2.9 Mbp, query K=%d" % K,
                                                "bxi_bytes": os.path.getsize(bxi), "phases_ms": t})


def _fastq_blob(ids_width, reads, quals):
    """fixed-width FASTQ text: '@r%0*d\n' + bases + '\n+\n' + quals + '\n' per read, built column-wise"""
    n, L = reads.shape
    w = 2 + ids_width + 1 + L + 3 + L + 1
    rec = np.empty((n, w), np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    num = np.arange(n, dtype=np.int64)
    for j in range(ids_width):
        rec[:, 2 + ids_width - 1 - j] = 48 + (num // 10 ** j) % 10
    c = 2 + ids_width
    rec[:, c] = 10
    rec[:, c + 1:c + 1 + L] = reads
    rec[:, c + 1 + L:c + 4 + L] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, c + 4 + L:c + 4 + 2 * L] = quals
    rec[:, w - 1] = 10
    return rec.reshape(-1).tobytes()


def _write_bgzf_blocks(path, blob, block=65280):
    import zlib
    with open(path, "wb") as f:
        for i in range(0, len(blob), block):
            c = blob[i:i + block]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
            f.write(body + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_config_c_build_and_read_id_at_full_size(orc, tmp_path):
    import subprocess
    import zlib

    from test_gpu_cli import BANNER, BIN, counts_file

    def cli(*args, env=None):
        p = subprocess.run([BIN, *args], capture_output=True, text=True, env=dict(os.environ, COLORID_TIMING="1", **(env or {})))
        assert p.returncode == 0, p.stderr[-3000:]
        assert p.stdout.startswith(BANNER)
        return p.stdout[len(BANNER):], p.stderr

    t = {"t0": time.perf_counter()}

    def lap(name):
        now = time.perf_counter()
        t[name] = round((now - t["t0"]) * 1e3)
        t["t0"] = now

    n_acc, m, n, k, Lg, R, L = 256, 30_000_000, 2, 21, 5_000_000, 1_000_000, 150
    rng = np.random.default_rng(2026)
    meta = ACGT[rng.integers(0, 4, n_acc * Lg, dtype=np.uint8)]
    lines = []
    for i in range(n_acc):
        fa = tmp_path / f"g{i:03d}.fasta"
        with open(fa, "wb") as f:
            f.write(b">genome%03d\n" % i)
            f.write(meta[i * Lg:(i + 1) * Lg].tobytes())
            f.write(b"\n")
        lines.append(f"genome{i:03d}\t{fa}\n")
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(lines))
    lap("make_genomes_ms")
    pre = str(tmp_path / "cfg2")
    _, err_b = cli("build", "-k", str(k), "-s", str(m), "-n", str(n), "-b", pre, "-r", str(tsv))
    bxi = pre + ".bxi"
    lap("cli_build_ms")
    # reads: fragments of 300-500 bp, mate 1 from the fragment's start, mate 2 the reverse complement of its end; 1 % substitutions,
    # N at 0.3 % of the bases, phred < 15 at 4 % (masked to N by -Q 15), one read in 64 drawn from no genome at all
    gsel = rng.integers(0, n_acc, R)
    frag = rng.integers(300, 500, R)
    st = gsel.astype(np.int64) * Lg + rng.integers(0, Lg - 500, R)
    ar = np.arange(L, dtype=np.int64)
    r1 = meta[st[:, None] + ar[None, :]]
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    r2 = comp[meta[(st + frag)[:, None] - 1 - ar[None, :]]]
    alien = np.flatnonzero(rng.random(R) < 1 / 64)
    r1[alien] = ACGT[rng.integers(0, 4, (len(alien), L), dtype=np.uint8)]
    r2[alien] = ACGT[rng.integers(0, 4, (len(alien), L), dtype=np.uint8)]
    quals = []
    for rd in (r1, r2):
        e = rng.random(rd.shape) < 0.01
        rd[e] = ACGT[rng.integers(0, 4, int(e.sum()), dtype=np.uint8)]
        rd[rng.random(rd.shape) < 0.003] = ord("N")
        q = np.full(rd.shape, ord("I"), np.uint8)
        lq = rng.random(rd.shape) < 0.04
        q[lq] = rng.integers(33, 48, int(lq.sum()), dtype=np.uint8)
        quals.append(q)
    del meta
    blobs = [_fastq_blob(7, r1, quals[0]), _fastq_blob(7, r2, quals[1])]
    files = {}
    for mate, blob in enumerate(blobs):
        plain = str(tmp_path / f"reads_{mate + 1}.fastq.gz")
        co = zlib.compressobj(1, zlib.DEFLATED, 31)           # one gzip stream, as `gzip -1` writes it
        with open(plain, "wb") as f:
            f.write(co.compress(blob)); f.write(co.flush())
        bg = str(tmp_path / f"reads_{mate + 1}.bgzf.fastq.gz")
        _write_bgzf_blocks(bg, blob)
        files[("plain", mate)], files[("bgzf", mate)] = plain, bg
    del blobs
    lap("make_reads_ms")
    outs, errs = {}, {}
    for kind in ("plain", "bgzf"):
        for pe in (False, True):
            prefix = str(tmp_path / f"rid_{kind}_{int(pe)}")
            q = [files[(kind, 0)]] + ([files[(kind, 1)]] if pe else [])
            _, e = cli("read_id", "-b", bxi, "-q", *q, "-n", prefix)
            outs[(kind, pe)] = (open(prefix + "_reads.txt").read().splitlines(), open(prefix + "_counts.txt").read().splitlines())
            errs[(kind, pe)] = [ln.split("\r")[-1] for ln in e.splitlines() if "timing:" in ln or "Classified" in ln]
            lap(f"cli_read_id_{kind}_{'pe' if pe else 'se'}_ms")
    for pe in (False, True):
        assert outs[("plain", pe)] == outs[("bgzf", pe)]                      # the two front ends write the same files
        rows = outs[("plain", pe)][0]
        assert len(rows) == R
        assert sorted(outs[("plain", pe)][1]) == counts_file(rows)            # reports.rs:98-120 over all million rows
    # the oracle on the index file the CLI wrote, for a sample spread over the input (batch starts, middles and the tail)
    oix = orc.Index.read(bxi)
    assert (oix.m, oix.n_hash, oix.k, oix.n_colors) == (m, n, k, n_acc)
    lap("oracle_reads_bxi_ms")
    sample = np.r_[0:1000, 49_990:50_010, 499_000:500_000, R - 980:R]
    labels = set()
    for pe in (False, True):
        masked = [[orc.qual_mask(r1[i].tobytes(), quals[0][i].tobytes(), 15)] + ([orc.qual_mask(r2[i].tobytes(), quals[1][i].tobytes(), 15)] if pe else [])
                  for i in sample]
        seqs = [s for rd in masked for s in rd]
        seq_off = np.zeros(len(seqs) + 1, np.uint64)
        seq_off[1:] = np.cumsum([len(s) for s in seqs])
        read0 = np.arange(len(masked) + 1, dtype=np.uint64) * (2 if pe else 1)
        rep, nk, st_ = oix.readid_counts(np.frombuffer(b"".join(seqs), np.uint8), seq_off, read0, 1, 3, n_threads=8)
        got = outs[("plain", pe)][0]
        for j, i in enumerate(sample):
            rid = "@r%07d" % i
            if st_[j] == 1:
                want = f"{rid}\ttoo_short\t0\t0\taccept\t0"
            else:
                lab, cnt, kl, verdict, ntop = oix.kmer_poll_plus(rep[j].astype(np.uint64), int(nk[j]), 1e-3)
                want = f"{rid}\t{lab}\t{cnt}\t{kl}\t{verdict}\t{ntop}"
                labels.add((lab, verdict))
            assert got[i] == want, (pe, i)
    lap("oracle_sample_ms")
    # the classification means something: most reads go to the genome they were cut from, the reads from nowhere do not
    se_rows = outs[("plain", False)][0]
    aliens = set(alien.tolist())
    right = sum(1 for i in sample if i not in aliens and se_rows[i].split("\t")[1] == "genome%03d" % gsel[i])
    assert right > 0.8 * len(sample)
    turned_away = sum(1 for i in alien[:400] if se_rows[i].split("\t")[4] == "reject" or se_rows[i].split("\t")[1] in ("no_hits", "too_short"))
    assert turned_away > 0.9 * 400
    assert len(labels) > 100
    t.pop("t0")
    _record_timings("r04_config_c_full.json", {"config": "configs[2]: build -k 21 -s 30000000 -n 2 on 256 x 5 Mbp + read_id on 1 M reads (SE, PE; single-stream gzip, block gzip)",
                                                "bxi_bytes": os.path.getsize(bxi), "sample_reads_checked": int(len(sample)) * 2, "phases_ms": t,
                                                "build_stderr": [ln for ln in err_b.splitlines() if "timing:" in ln], "read_id_stderr": {f"{k_}_{'pe' if p_ else 'se'}": v for (k_, p_), v in errs.items()}})


def _canon_codes(torch, reads, k):
    """canonical 2-bit window codes (A<C<G<T, base 0 most significant) of reads given as codes 0..3: int64 [n, L-k+1] — a torch
    restatement of kmer.rs:493-499 for upper-case ACGT reads (min of the window and its reverse complement)"""
    n, L = reads.shape
    nw = L - k + 1
    fwd = torch.zeros((n, nw), device=reads.device, dtype=torch.int64)
    rc = torch.zeros((n, nw), device=reads.device, dtype=torch.int64)
    for j in range(k):
        c = reads[:, j:j + nw].to(torch.int64)
        fwd = (fwd << 2) | c
        rc = rc | ((3 - c) << (2 * j))
    return torch.minimum(fwd, rc)


def _codes_to_ascii(torch, codes, k):
    shifts = torch.arange(2 * (k - 1), -1, -2, device=codes.device, dtype=torch.int64)
    lut = torch.tensor(list(b"ACGT"), device=codes.device, dtype=torch.uint8)
    return lut[((codes[:, None] >> shifts[None, :]) & 3)].contiguous()


def test_config_d_one_gpus_share_of_the_query(orc, hip_ctx):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import colorid_amd
    from colorid_amd._lib import check, vp
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~120 GiB of free HBM")
    lib = hip_ctx.lib
    Cn, n, k, m = 1024, 4, 31, 50_000_000
    R, L, CH = 12_500_000, 150, 8
    per = R // CH
    nw = L - k + 1
    t = {}
    t0 = time.perf_counter()

    def lap(name, t_start):
        hip_ctx.synchronize()
        torch.cuda.synchronize()
        t[name] = round((time.perf_counter() - t_start) * 1e3, 1)
        return time.perf_counter()

    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    G = 2_000_000_000                                          # a 2 Gbp "metagenome": the share covers it 0.94x
    meta = torch.randint(0, 4, (G,), device=dev, dtype=torch.uint8, generator=g)
    ar = torch.arange(L, device=dev, dtype=torch.int64)
    chunks = []
    for c in range(CH):
        st = torch.randint(0, G - L, (per,), device=dev, dtype=torch.int64, generator=g)
        rd = meta[st[:, None] + ar[None, :]].contiguous()
        flip = torch.rand(per, device=dev, generator=g) < 0.5    # half the reads come from the other strand
        rd[flip] = (3 - rd[flip]).flip(1)
        err = torch.rand((per, L), device=dev, generator=g) < 0.01
        rd[err] = torch.randint(0, 4, (int(err.sum().item()),), device=dev, dtype=torch.uint8, generator=g)
        chunks.append(rd.contiguous())
        del st, flip, err
    del meta
    lut = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)
    host = [lut[rd.to(torch.int64)].cpu().numpy() for rd in chunks]          # the reads as the host hands them over: ASCII, 150 bytes each
    seq_off = (np.arange(per + 1, dtype=np.uint64) * L)
    t1 = lap("make_reads_ms", t0)

    def count(order, compact_windows):
        hip_ctx.tune("kmerset_compact_at", compact_windows)   # (read when the set is made)
        try:
            ks = colorid_amd.KmerSet(hip_ctx, k)
        finally:
            hip_ctx.tune("kmerset_compact_at", 0)
        for c in order:
            check(lib.cid_kmerset_add_seqs(ks.h, vp(host[c].ctypes.data), vp(seq_off.ctypes.data), per, 1))
        nd = ks.finalize()
        dc, dn, nn = vp(), vp(), C.c_uint64(0)
        check(lib.cid_kmerset_device_arrays(ks.h, C.byref(dc), C.byref(dn), C.byref(nn)))
        assert nn.value == nd
        return ks, nd, dc.value, dn.value

    # 12.5 M reads in 8 calls of 187.5 M windows; the window buffer is merged into the set whenever it passes 160 M codes: 8 incremental merges
    ks, K, p_codes, p_counts = count(range(CH), 160_000_000)
    t2 = lap("kmerset_count_ms", t1)
    assert 900_000_000 < K < R * nw
    codes = torch.empty(K, dtype=torch.int64, device=dev)
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    bench.hip_memcpy(codes.data_ptr(), p_codes, K * 8, 3)
    bench.hip_memcpy(counts.data_ptr(), p_counts, K * 4, 3)
    assert bool((codes[1:] > codes[:-1]).all())                                  # distinct, ascending
    assert int(counts.to(torch.int64).sum().item()) == R * nw                    # every window of every read is in the set exactly once
    vals, cnts = ks.histogram()                                                  # what auto_cutoff reads (kmer.rs:866-942)
    assert int(cnts.sum()) == K and int((vals.astype(np.int64) * cnts.astype(np.int64)).sum()) == R * nw
    assert vals[0] == 1 and 0.4 * K < cnts[0] < 0.95 * K and len(vals) > 4      # 0.94x coverage: mostly singletons, a tail of repeats
    # order / merge-schedule independence: the chunks in another order, merged on another schedule -> the identical arrays
    ks2, K2, p2_codes, p2_counts = count([5, 2, 7, 0, 3, 6, 1, 4], 410_000_000)
    t3 = lap("kmerset_recount_ms", t2)
    assert K2 == K
    c2 = torch.empty(K, dtype=torch.int64, device=dev)
    n2 = torch.empty(K, dtype=torch.int32, device=dev)
    bench.hip_memcpy(c2.data_ptr(), p2_codes, K * 8, 3)
    bench.hip_memcpy(n2.data_ptr(), p2_counts, K * 4, 3)
    assert torch.equal(c2, codes) and torch.equal(n2, counts)
    del c2, n2
    ks2.close()
    # the oracle's fastq map of the first 1,000 reads == the torch restatement on them; the restatement over ALL reads == the GPU set
    S_READS = 1000
    okm = orc.Kmers(k)
    qual = b"I" * L
    for r in range(S_READS):
        okm.kmerize_fq_read(host[0][r].tobytes(), qual, 15)
    okeys, ocnt = okm.keys(), okm.counts()
    assert len(okm) >= 100_000
    w = (1 << (2 * np.arange(k - 1, -1, -1, dtype=np.int64))).astype(np.int64)
    ocodes = (((okeys >> 1) & 3) ^ (((okeys >> 1) & 3) >> 1)).astype(np.int64) @ w    # A,C,G,T -> 0..3
    order = np.argsort(ocodes)
    sample = torch.from_numpy(ocodes[order]).to(dev)
    S = sample.numel()
    first = _canon_codes(torch, chunks[0][:S_READS], k).reshape(-1)
    u, uc_first = torch.unique(first, return_counts=True)
    assert torch.equal(u, sample) and np.array_equal(uc_first.cpu().numpy(), ocnt[order].astype(np.int64))
    full = torch.zeros(S, dtype=torch.int64, device=dev)
    for rd in chunks:
        for r0 in range(0, per, 400_000):
            cc = _canon_codes(torch, rd[r0:r0 + 400_000], k).reshape(-1)
            idx = torch.searchsorted(sample, cc).clamp_(max=S - 1)
            hit = sample[idx] == cc
            full += torch.bincount(idx[hit], minlength=S)
            del cc, idx, hit
    pos = torch.searchsorted(codes, sample)
    assert bool((codes[pos] == sample).all())
    assert torch.equal(counts[pos].to(torch.int64), full) and bool((full >= uc_first).all()) and int((full > uc_first).sum().item()) > 1000
    t4 = lap("oracle_and_restatement_ms", t3)
    # the index: configs[3]'s 50 M x 1024 colours (6.4 GB), background + the k-mers of the first 100,000 reads planted
    hx = colorid_amd.Index(hip_ctx, m, n, k, Cn)
    ptr, rs = hx.device_matrix()
    bench.fill_background_fast(dev, ptr, m, rs, Cn, 0.2134, seed=11)
    pl = _canon_codes(torch, chunks[0][:100_000], k).reshape(-1)
    pl_col = (torch.arange(100_000, device=dev, dtype=torch.int32) % Cn)[:, None].expand(-1, nw).reshape(-1).contiguous()
    pl_ascii = _codes_to_ascii(torch, pl, k)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(pl_ascii.data_ptr(), pl_col.data_ptr(), pl.numel())
    hip_ctx.synchronize()
    hx.finalize()
    del pl, pl_col, pl_ascii
    t5 = lap("index_ms", t4)

    def search(c_t, f_t):
        nk = c_t.numel()
        out = torch.zeros(3 * Cn, dtype=torch.int64, device=dev)
        uc = torch.empty(nk, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        check(lib.cid_search_count_codes_dev(hip_ctx.h, hx.h, vp(c_t.data_ptr()), vp(f_t.data_ptr()), nk, vp(out.data_ptr()),
                                             vp(out.data_ptr() + 8 * Cn), vp(out.data_ptr() + 16 * Cn), vp(uc.data_ptr())))
        hip_ctx.synchronize()
        return out, uc

    whole, uc = search(codes, counts)
    t6 = lap("search_whole_set_ms", t5)
    # the same share counted FOR this index (cid_kmerset_set_target_index: what `colorid search` does) — 8 merges on the (first-row key,
    # code) pair: the same k-mers and multiplicities in another order, the same counters from the search, (key, code) order
    hip_ctx.tune("kmerset_compact_at", 160_000_000)
    try:
        kt = colorid_amd.KmerSet(hip_ctx, k)
    finally:
        hip_ctx.tune("kmerset_compact_at", 0)
    kt.set_target_index(hx)
    for c in range(CH):
        check(lib.cid_kmerset_add_seqs(kt.h, vp(host[c].ctypes.data), vp(seq_off.ctypes.data), per, 1))
    assert kt.finalize() == K
    dc, dn, nn = vp(), vp(), C.c_uint64(0)
    check(lib.cid_kmerset_device_arrays(kt.h, C.byref(dc), C.byref(dn), C.byref(nn)))
    t_codes = torch.empty(K, dtype=torch.int64, device=dev)
    t_counts = torch.empty(K, dtype=torch.int32, device=dev)
    bench.hip_memcpy(t_codes.data_ptr(), dc.value, K * 8, 3)
    bench.hip_memcpy(t_counts.data_ptr(), dn.value, K * 4, 3)
    t6 = lap("kmerset_count_for_index_ms", t6)
    whole_t, uc_t = search(t_codes, t_counts)
    t6 = lap("search_set_built_for_index_ms", t6)
    assert torch.equal(whole_t, whole)
    by_code = torch.argsort(t_codes)
    assert torch.equal(t_codes[by_code], codes) and torch.equal(t_counts[by_code], counts) and torch.equal(uc_t[by_code], uc)
    del by_code
    at = int(K // 2)
    win = _codes_to_ascii(torch, t_codes[at:at + 20_000], k).cpu().numpy()
    keyed = [(((orc.xxh3(win[j].tobytes(), 0) % m) * (0xFFFFFFFF00000000 // m)) >> 32, win[j].tobytes()) for j in range(len(win))]
    assert keyed == sorted(keyed)
    del t_codes, t_counts, whole_t, uc_t
    kt.close()
    t6 = lap("set_built_for_index_checks_ms", t6)
    again, uc_again = search(codes, counts)
    assert torch.equal(whole, again) and torch.equal(uc, uc_again)
    cut = K // 3 + 11
    a, uca = search(codes[:cut].clone(), counts[:cut].clone())
    b, ucb = search(codes[cut:].clone(), counts[cut:].clone())
    assert torch.equal(a + b, whole) and torch.equal(torch.cat([uca, ucb]), uc)
    del a, b, uca, ucb, again, uc_again
    P = 200_000_000
    perm = torch.randperm(P, device=dev, generator=g)
    base, ucb0 = search(codes[:P].clone(), counts[:P].clone())
    p_out, ucp = search(codes[:P][perm].contiguous(), counts[:P][perm].contiguous())
    assert torch.equal(p_out, base) and torch.equal(ucp, ucb0[perm])
    del perm, p_out, ucp, base, ucb0
    hits, nu, sf = whole[:Cn], whole[Cn:2 * Cn], whole[2 * Cn:]
    assert int(nu.sum().item()) == int((uc != -1).sum().item()) and int(hits.sum().item()) > K
    assert int(sf.sum().item()) == int(counts.to(torch.int64)[uc != -1].sum().item())
    t7 = lap("search_properties_ms", t6)
    # the oracle on the sample (its index holds the rows the sample touches), with the multiplicities of the WHOLE share
    hk = _codes_to_ascii(torch, sample, k).cpu().numpy()
    hf = full.cpu().numpy()
    ridx = np.unique(np.array([orc.xxh3(hk[j].tobytes(), s) % m for j in range(S) for s in range(n)], np.uint64))
    oix = orc.Index(m, n, k, Cn)
    oix.rows()[ridx.astype(np.int64)] = hx.get_rows(ridx)
    want = oix.search_count(hk, hf.astype(np.uint64))
    got = hx.search_count(hk, hf.astype(np.uint32))
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    assert np.array_equal(got[3].view(np.int32), uc[pos].cpu().numpy())          # the big launch says the same about these k-mers
    assert (got[3] != 0xFFFFFFFF).sum() > 1000 and want[0].sum() > S
    lap("oracle_search_sample_ms", t7)
    t["total_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    _record_timings("r04_config_d_share.json", {
        "config": "configs[3], one GPU's share: 12.5 M x 150 bp reads (%d windows) -> cid_kmerset in %d add_seqs calls, merged every 160 M windows; "
                  "m=50M n=4 k=31 C=1024" % (R * nw, CH),
        "distinct_kmers": K, "singletons": int(cnts[0]), "phases_ms": t,
        "kmerset_count_reads_per_s": R / (t["kmerset_count_ms"] * 1e-3), "search_kmers_per_s": K / (t["search_whole_set_ms"] * 1e-3)})
    del oix, codes, counts, whole, uc, chunks
    ks.close()
    hx.close()
    torch.cuda.empty_cache()
