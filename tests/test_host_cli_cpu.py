"""Host-side logic of the C++ `colorid` binary that runs without a GPU: FASTA/FASTQ(.gz) parsing, quality masking,
canonical k-mer counting (FASTA / fastq SE / fastq PE / multi-FASTA modes), clean_map + auto_cutoff and `info`,
each against the oracle's restatement of the same reference function."""
import os
import subprocess

import numpy as np
import pytest

from util import synth_fastq_records, write_fastq_gz

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.environ.get("COLORID_BIN", os.path.join(ROOT, "colorid_amd", "bin", "colorid"))   # COLORID_BIN: e.g. a sanitizer build
REFS = os.path.join(HERE, "golden", "refs")
BANNER = "\n ************** initializing logger *****************\n\n"   # src/main.rs:18


def run(*args):
    p = subprocess.run([BIN, *args], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith(BANNER)
    return p.stdout[len(BANNER):], p.stderr


def parse_kmers(out):
    d, cutoff = {}, None
    for line in out.splitlines():
        if line.startswith("#auto_cutoff"):
            cutoff = int(line.split("\t")[1])
            continue
        k, c = line.split("\t")
        d[k.encode()] = int(c)
    return d, cutoff


@pytest.fixture(scope="module")
def genomes(orc):
    return [b"".join(orc.read_fasta(os.path.join(REFS, n))) for n in sorted(os.listdir(REFS))]


def test_binary_built():
    assert os.path.exists(BIN), "build with __graft_entry__.build()"


@pytest.mark.parametrize("name", sorted(os.listdir(REFS)))
@pytest.mark.parametrize("k", [21, 27, 31])
def test_fasta_kmers(orc, name, k):
    path = os.path.join(REFS, name)
    km = orc.Kmers(k)
    for s in orc.read_fasta(path):
        km.kmerize_vector(s, 1)
    got, _ = parse_kmers(run("debug-kmers", "-q", path, "-k", str(k))[0])
    assert got == km.as_dict()


def test_fasta_quirks(orc, tmp_path):
    # header = any line containing '>' (kmer.rs:26); CRLF stripped by lines(); lower case upper-cased after the compare;
    # last line without newline; N windows dropped
    p = tmp_path / "q.fasta"
    p.write_bytes(b">r1 desc\r\nACGTNACGTTGCA\r\nacgtacgtaa\r\nseq with > inside is a header\nGGGGCCCCAT\n>r3\nTTTTT\n>r4\nACGTACGTAC")
    for k in (4, 5, 9):
        km = orc.Kmers(k)
        for s in orc.read_fasta(str(p)):
            km.kmerize_vector(s, 1)
        got, _ = parse_kmers(run("debug-kmers", "-q", str(p), "-k", str(k))[0])
        assert got == km.as_dict() and len(got) > 0
    labels, seqs = orc.read_fasta_mf(str(p))
    out = run("debug-kmers", "-q", str(p), "-k", "6", "--mode", "mf")[0].splitlines()
    want = []
    for lab, s in zip(labels, seqs):
        km = orc.Kmers(6)
        want.append(f">{lab.decode()}\t{len(km) if km.kmerize_string(s) == 0 else -1}")
    assert out == want and any(w.endswith("-1") for w in want)


@pytest.mark.parametrize("q", [0, 15, 30])
def test_fastq_se_and_pe_kmers(orc, genomes, tmp_path, q):
    rng = np.random.default_rng(q)
    r1 = synth_fastq_records(rng, genomes, 400, 120, mate=0)
    rng = np.random.default_rng(q)
    r2 = synth_fastq_records(rng, genomes, 380, 120, mate=1)   # file 2 shorter: the reference stops there
    f1, f2 = str(tmp_path / "a_1.fastq.gz"), str(tmp_path / "a_2.fastq.gz")
    write_fastq_gz(f1, r1, multi_member=True)
    write_fastq_gz(f2, r2)
    want = orc.kmers_from_fq_qual(f1, 27, q).as_dict()
    got, _ = parse_kmers(run("debug-kmers", "-q", f1, "-k", "27", "--mode", "fq", "-Q", str(q))[0])
    assert got == want and len(want) > 1000
    want = orc.kmers_fq_pe_qual(f1, f2, 27, q).as_dict()
    got, _ = parse_kmers(run("debug-kmers", "-q", f1, f2, "-k", "27", "--mode", "fqpe", "-Q", str(q))[0])
    assert got == want


def test_clean_map_and_auto_cutoff(orc, genomes, tmp_path):
    rng = np.random.default_rng(3)
    recs = synth_fastq_records(rng, [genomes[0][:3000]], 3000, 100, err=0.02)   # ~100x coverage of a 3 kb region
    f1 = str(tmp_path / "cov.fastq.gz")
    write_fastq_gz(f1, recs)
    km = orc.kmers_from_fq_qual(f1, 21, 15)
    cutoff = km.auto_cutoff()
    assert cutoff >= 1
    got, got_cut = parse_kmers(run("debug-kmers", "-q", f1, "-k", "21", "--mode", "fq", "-f", "-1")[0])
    assert got_cut == cutoff and got == km.clean_map(cutoff).as_dict()
    got, _ = parse_kmers(run("debug-kmers", "-q", f1, "-k", "21", "--mode", "fq", "-f", "3")[0])
    assert got == km.clean_map(3).as_dict()


def test_info_matches_reference_format(orc, tmp_path):
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(f"{n[:-6]}\t{os.path.join(REFS, n)}\n" for n in sorted(os.listdir(REFS))))
    oix = orc.Index.build_single(str(tsv), 750000, 4, 27)
    bxi = str(tmp_path / "phage.bxi")
    oix.save(bxi)
    out, err = run("info", "-b", bxi)
    lines = out.splitlines()
    assert lines[:5] == ["BIGSI parameters:", "Bloomfilter-size: 750000", "Number of hashes: 4", "K-mer size: 27",
                         "Number of accessions in index: 4"]     # src/main.rs:673-677
    for c, (name, n) in enumerate(zip(oix.colors(), oix.n_ref_kmers())):
        assert lines[5 + c] == f"{name} {n} {orc.false_prob(750000, 4, n):.3f}"
    assert "Loading index" in err and "Index loaded in" in err


def test_cli_errors():
    p = subprocess.run([BIN, "search", "-b", "/nonexistent.bxi", "-q", "x.fasta"], capture_output=True, text=True)
    assert p.returncode != 0 and ("GPU" in p.stderr or "index" in p.stderr)
    p = subprocess.run([BIN, "search", "-q", "x.fasta"], capture_output=True, text=True)
    assert p.returncode != 0 and "required" in p.stderr
    # batch_id (main.rs:329-418): -b, -q and -T are required; a sheet that is not there ends the run before any GPU work
    p = subprocess.run([BIN, "batch_id", "-b", "/nonexistent.bxi", "-q", "sheet.tsv"], capture_output=True, text=True)
    assert p.returncode != 0 and "--tag" in p.stderr
    p = subprocess.run([BIN, "batch_id", "-b", "/nonexistent.bxi", "-q", "/nonexistent_sheet.tsv", "-T", "t"], capture_output=True, text=True)
    assert p.returncode != 0
    p = subprocess.run([BIN, "read_filter"], capture_output=True, text=True)
    assert p.returncode != 0 and "outside the accelerated query path" in p.stderr


def test_bxi_loader_rejects_malformed_files(orc, tmp_path):
    """The reference panics ("can't deserialize") on a malformed index; the C++ loader must fail loudly too, never read past
    the file or accept rows of the wrong width (parsed through `info`, which walks the whole file without a GPU)."""
    import struct
    oix = orc.Index(1000, 2, 21, 40)
    for c in range(40):
        oix.set_color(c, f"acc{c:02d}", 100 + c)
    rows = oix.rows()
    rows[5, 0] = 3
    rows[900, 1] = 0x80
    good = str(tmp_path / "good.bxi")
    oix.save(good)
    out, _ = run("info", "-b", good)
    assert "Number of accessions in index: 40" in out
    raw = open(good, "rb").read()

    def expect_fail(blob, name):
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        r = subprocess.run([BIN, "info", "-b", p], capture_output=True, text=True)
        assert r.returncode != 0 and "deserialize" in r.stderr, (name, r.stderr)

    expect_fail(raw[:len(raw) // 2], "truncated.bxi")
    expect_fail(raw[:40], "header_only.bxi")
    # first row record starts after the colours block: find it and corrupt its word count (2 -> 3)
    off = 32 + sum(16 + len(f"acc{c:02d}") for c in range(40)) + 8
    assert struct.unpack_from("<2Q", raw, off) == (5, 2)
    expect_fail(raw[:off + 8] + struct.pack("<Q", 3) + raw[off + 16:], "bad_words.bxi")
    expect_fail(raw[:off + 16 + 8] + struct.pack("<Q", 41) + raw[off + 32:], "bad_nbits.bxi")
    expect_fail(struct.pack("<4Q", 1000, 2, 21, 0) + raw[32:], "no_colours.bxi")


def test_line_reader_edge_cases(orc, genomes, tmp_path):
    """The threaded gz/plain reader keeps BufRead::lines() semantics: CRLF, a last line without a newline, records whose
    sequence line is longer than the reader's 4 MiB blocks, plain (uncompressed) input, an empty file."""
    import gzip
    g = genomes[0]
    big = (g * 300)[:9_000_000]                                  # one 9 Mbp sequence line: spans three decode blocks
    recs = [(b"r0", g[:200], b"I" * 200), (b"big", big, b"I" * len(big)), (b"r2", g[300:420], b"I" * 120)]

    def blob(eol, last_newline=True):
        s = b"".join(b"@" + i + eol + q + eol + b"+" + eol + w + eol for i, q, w in recs)
        return s if last_newline else s[:-len(eol)]

    want = None
    for tag, eol, last, gz in (("lf", b"\n", True, True), ("crlf", b"\r\n", True, True), ("nolast", b"\n", False, True), ("plain", b"\n", True, False),
                               ("crlf_plain_nolast", b"\r\n", False, False)):
        path = str(tmp_path / (tag + (".fastq.gz" if gz else ".fastq")))
        data = blob(eol, last)
        if gz:
            with gzip.open(path, "wb", compresslevel=1) as f:
                f.write(data)
        else:
            open(path, "wb").write(data)
        got, _ = parse_kmers(run("debug-kmers", "-q", path, "-k", "27", "--mode", "fq", "-Q", "15")[0])
        if want is None:
            km = orc.Kmers(27)
            for _, seq, qual in recs:
                km.kmerize_fq_read(seq, qual, 15)
            want = km.as_dict()
            assert len(want) > 10_000
        assert got == want, tag
    empty = str(tmp_path / "empty.fastq.gz")
    with gzip.open(empty, "wb") as f:
        pass
    got, _ = parse_kmers(run("debug-kmers", "-q", empty, "-k", "27", "--mode", "fq", "-Q", "15")[0])
    assert got == {}


def _line_loop_reads(texts, q):
    """The reference's line loops restated (read_id_mt_pe.rs:862-895 single-end, :927-975 paired; seq.rs:36-56 qual_mask): lines as
    BufRead::lines() yields them, a record pushed at every fourth line, for pairs until the shorter file ends."""
    def lines(t):
        ls = t.split(b"\n")
        if ls and ls[-1] == b"":
            ls.pop()
        return [x[:-1] if x.endswith(b"\r") else x for x in ls]

    def mask(seq, qual):
        if q == 0:
            return seq
        assert len(seq) >= len(qual)
        return bytes(b"N"[0] if w < q + 33 else s for s, w in zip(seq, qual))
    ls = [lines(t) for t in texts]
    out = []
    n = min(len(x) for x in ls)
    for r in range(n // 4):
        out.append((ls[0][4 * r], [mask(x[4 * r + 1], x[4 * r + 3]) for x in ls]))
    return out


@pytest.mark.parametrize("threads", ["1", "4"])
def test_record_pipeline_equals_the_line_loops(tmp_path, threads):
    """`read_id` / `search` cut the decoded FASTQ text into whole records and pack them on several threads (RecordChunker +
    pack_records): the reads, their order, their ids and the quality masking must be those of the reference's line-by-line loops —
    over block boundaries (4 MiB decode blocks, 777-byte BGZF members), CRLF, records longer than a block and longer than the
    64 KiB carry headroom, a last line without its newline, trailing lines that complete no record, files of different length."""
    import gzip
    from test_linereader_cpu import write_bgzf
    rng = np.random.default_rng(int(threads))
    env = dict(os.environ, COLORID_PARSE_THREADS=threads)

    def fastq(n, eol=b"\n", long_at=(), seed=0):
        r = np.random.default_rng(seed)
        out = []
        for i in range(n):
            L = int(r.integers(1, 300)) if i not in long_at else int(r.integers(70_000, 5_000_000))
            s = bytes(r.choice(list(b"ACGTNacgt"), size=L).astype(np.uint8))
            w = bytes(r.integers(33, 75, size=L).astype(np.uint8))
            out.append(b"@r%d/x y" % i + eol + s + eol + b"+" + eol + w + eol)
        return b"".join(out)

    def got_reads(paths, q):
        p = subprocess.run([BIN, "debug-records", "-q", *[str(x) for x in paths], "-Q", str(q)], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[-500:]
        out = p.stdout[len(BANNER):]
        res = []
        for line in out.split(b"\n")[:-1]:
            f = line.split(b"\t")
            res.append((f[0], f[1:]))
        return res

    cases = {
        "big": fastq(40_000, seed=1),                                    # ~ 12 MB: several 4 MiB blocks
        "crlf": fastq(3_000, eol=b"\r\n", seed=2),
        "long": fastq(60, long_at=(3, 17, 18, 59), seed=3),              # records longer than the headroom and than a block
        "nolast": fastq(500, seed=4)[:-1],                               # the last quality line has no newline
        "trailing": fastq(500, seed=5) + b"@partial\nACGT\n",            # two lines that complete no record
        "empty": b"",
        "one": fastq(1, seed=6),
    }
    for tag, text in cases.items():
        for q in (15, 0):
            want = _line_loop_reads([text], q)
            for kind in ("plain", "gz", "bgzf"):
                if kind != "gz" and tag in ("long",) and q == 0:
                    continue
                path = tmp_path / f"{tag}_{kind}.fastq{'' if kind == 'plain' else '.gz'}"
                if kind == "plain":
                    path.write_bytes(text)
                elif kind == "gz":
                    with gzip.open(path, "wb", compresslevel=1) as f:
                        f.write(text)
                else:
                    write_bgzf(path, text, block=777 if len(text) < 1_000_000 else 65280, level=1)
                got = got_reads([path], q)
                assert len(got) == len(want), (tag, kind, q, len(got), len(want))
                bad = [i for i in range(len(want)) if got[i] != want[i]]
                assert not bad, (tag, kind, q, bad[:3])
    # pairs: files of different length (and a mate file that ends inside a record): the walk ends with the shorter one
    a, b = fastq(30_000, seed=7), fastq(30_000, seed=8)
    cut = b[: len(b) * 2 // 3]
    for second in (b, cut, b""):
        pa, pb = tmp_path / "p1.fastq.gz", tmp_path / "p2.fastq.gz"
        with gzip.open(pa, "wb", compresslevel=1) as f:
            f.write(a)
        write_bgzf(pb, second, level=1)
        want = _line_loop_reads([a, second], 15)
        got = got_reads([pa, pb], 15)
        assert len(got) == len(want) and all(g == w for g, w in zip(got, want)), (len(second), len(got), len(want))
    assert len(_line_loop_reads([a, cut], 15)) < 30_000
