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
