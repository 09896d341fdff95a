"""The C++ `colorid` CLI (the reference's drop-in surface) end to end on the GPU, against output composed from the
oracle's restatement of the same reference functions: build -> .bxi bytes, search -s / -s -m / default / -g on FASTA and
fastq.gz (SE, PE), read_id on fastq.gz SE/PE and FASTA.  Row ORDER is unspecified in the reference (RandomState
HashMap), so multi-row outputs are compared as sorted lists."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_readid import pack_reads
from util import synth_fastq_records, write_fastq_gz

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.environ.get("COLORID_BIN", os.path.join(ROOT, "colorid_amd", "bin", "colorid"))   # COLORID_BIN: e.g. a sanitizer build
REFS = os.path.join(HERE, "golden", "refs")
PHAGES = ["Listeria_phage_B021", "Listeria_phage_B051", "Listeria_phage_B056", "Listeria_phage_B545"]
BANNER = "\n ************** initializing logger *****************\n\n"


def run(*args, cwd=None, host_kmers=False):
    env = dict(os.environ)
    if host_kmers:
        env["COLORID_HOST_KMERS"] = "1"      # k-mer map on the host instead of cid_kmerset
    p = subprocess.run([BIN, *args], capture_output=True, text=True, cwd=cwd, env=env)
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith(BANNER)
    return p.stdout[len(BANNER):], p.stderr


@pytest.fixture(scope="module")
def env(orc, tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    tsv = d / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in reversed(PHAGES)))
    # test.sh:3: build -s 750000 -n 4 -k 27
    out, err = run("build", "-s", "750000", "-n", "4", "-k", "27", "-b", str(d / "phage"), "-r", str(tsv))
    assert "Saving BIGSI to file." in out and "Adding Listeria_phage_B021 to index (1/4)" in err
    oix = orc.Index.build_single(str(tsv), 750000, 4, 27)
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    return d, str(d / "phage.bxi"), oix, genomes


def test_build_writes_identical_bxi(env, tmp_path):
    d, bxi, oix, _ = env
    ref = str(tmp_path / "oracle.bxi")
    oix.save(ref)
    assert open(bxi, "rb").read() == open(ref, "rb").read()          # k-mer maps counted on the GPU (default)
    run("build", "-s", "750000", "-n", "4", "-k", "27", "-b", str(tmp_path / "hostmap"), "-r", str(d / "ref_file.txt"), host_kmers=True)
    assert open(tmp_path / "hostmap.bxi", "rb").read() == open(ref, "rb").read()


def test_build_from_fastq_accessions(orc, env, tmp_path):
    """build.rs:54-84: paired and single fastq.gz accessions with -f (explicit) and the auto cutoff, next to a FASTA one."""
    d, _, _, genomes = env
    rng = np.random.default_rng(3)
    r1 = synth_fastq_records(rng, [genomes[0][:4000]], 2500, 120, mate=0, lower_rate=0.0)
    rng = np.random.default_rng(3)
    r2 = synth_fastq_records(rng, [genomes[0][:4000]], 2500, 120, mate=1, lower_rate=0.0)
    f1, f2, f3 = str(tmp_path / "a_1.fastq.gz"), str(tmp_path / "a_2.fastq.gz"), str(tmp_path / "b.fastq.gz")
    write_fastq_gz(f1, r1); write_fastq_gz(f2, r2); write_fastq_gz(f3, r1[:1500])
    tsv = tmp_path / "refs.tsv"
    tsv.write_text(f"pe_sample\t{f1}\t{f2}\nse_sample\t{f3}\nphage\t{os.path.join(REFS, PHAGES[1] + '.fasta')}\n")
    for flt in ([], ["-f", "2"]):
        pre = str(tmp_path / ("idx" + "".join(flt)))
        run("build", "-s", "200003", "-n", "3", "-k", "21", "-b", pre, "-r", str(tsv), *flt)
        oix = orc.Index.build_single(str(tsv), 200003, 3, 21, 15, int(flt[1]) if flt else -1)
        ref = pre + "_oracle.bxi"
        oix.save(ref)
        assert open(pre + ".bxi", "rb").read() == open(ref, "rb").read()
        nref = dict(zip(oix.colors(), oix.n_ref_kmers()))
        assert nref["pe_sample"] > 1000 and nref["se_sample"] > 1000
        assert (nref["phage"] == 0) == bool(flt)      # an explicit -f 2 empties a single-copy FASTA accession (build.rs:89-91)


@pytest.mark.parametrize("host_kmers", [False, True])
def test_search_perfect(orc, env, host_kmers):
    d, bxi, oix, _ = env
    q = os.path.join(REFS, "Listeria_phage_B056.fasta")
    out, err = run("search", "-b", bxi, "-q", q, "-s", host_kmers=host_kmers)
    km = orc.Kmers(27)
    for s in orc.read_fasta(q):
        km.kmerize_vector(s, 1)
    words, missing = oix.search_perfect(km.keys())
    assert not missing
    want = [f"{q}\t{oix.colors()[c]}\t{len(km)}\t1.00" for c in range(4) if words[0] >> c & 1]
    assert out.splitlines() == want and want == [f"{q}\tListeria_phage_B056\t32634\t1.00"]
    assert "32634 kmers in query" in err and "1 hits" in err
    # a query that is not in the index: absent rows -> "No perfect hits!" and no rows (perfect_search.rs:38-39)
    rnd = d / "random.fasta"
    rng = np.random.default_rng(1)
    rnd.write_bytes(b">r\n" + np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 500)].tobytes() + b"\n")
    out, err = run("search", "-b", bxi, "-q", str(rnd), "-s", host_kmers=host_kmers)
    assert out == "" and "No perfect hits!" in err


def test_search_perfect_multifasta(orc, env):
    d, bxi, oix, _ = env
    q = os.path.join(REFS, "Listeria_phage_B021.fasta")
    out, err = run("search", "-b", bxi, "-q", q, "-s", "-m")
    labels, seqs = orc.read_fasta_mf(q)
    want = []
    for lab, s in zip(labels, seqs):
        km = orc.Kmers(27)
        if km.kmerize_string(s) != 0:
            want.append(f"Warning! no kmers in query '{lab.decode()}'; maybe your kmer length is larger than your query length?")
            continue
        words, missing = oix.search_perfect(km.keys())
        if missing:
            continue
        want += [f"{lab.decode()}\t{oix.colors()[c]}\t{len(km)}\t1.00" for c in range(4) if words[0] >> c & 1]
    assert out.splitlines() == want and len(want) >= len(labels)


def expected_report(orc, oix, query, km, cov, gene):
    hits, nu, sf, uc = oix.search_count(km.keys(), km.counts())
    if gene:
        return oix.generate_report_gene(query, hits, len(km), cov).splitlines()
    modes = orc.unique_modes(uc, km.counts(), oix.n_colors)
    return oix.generate_report(query, hits, nu, sf, modes, len(km), cov).splitlines()


@pytest.mark.parametrize("host_kmers", [False, True])
@pytest.mark.parametrize("gene", [False, True])
def test_search_fasta(orc, env, gene, host_kmers):
    d, bxi, oix, genomes = env
    # a chimeric query: B056 + part of B545 + random sequence, so several accessions pass -p 0.05
    q = d / "chimera.fasta"
    rng = np.random.default_rng(2)
    q.write_bytes(b">a\n" + genomes[2][:9000] + b"\n>b\n" + genomes[3][2000:9000] + b"\n>c\n" +
                  np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)].tobytes() + b"\n")
    km = orc.Kmers(27)
    for s in orc.read_fasta(str(q)):
        km.kmerize_vector(s, 1)
    km = km.clean_map(0)   # gene: cutoff 0 (:112-113); default: auto_cutoff == 0 for an assembled query
    args = ["search", "-b", bxi, "-q", str(q), "-p", "0.05"] + (["-g"] if gene else [])
    out, err = run(*args, host_kmers=host_kmers)
    want = expected_report(orc, oix, str(q), km, 0.05, gene)
    assert sorted(out.splitlines()) == sorted(want) and len(want) >= 2
    assert f"{len(km)} k-mers in query" in err


@pytest.mark.parametrize("host_kmers", [False, True])
@pytest.mark.parametrize("mode", ["se_f1", "pe_f0", "se_gene", "pe_auto"])
def test_search_fastq(orc, env, mode, host_kmers):
    d, bxi, oix, genomes = env
    lower = 0.03 if mode == "se_f1" else 0.0     # lower-case reads force the host k-mer map (case is preserved, Q2)
    rng = np.random.default_rng(7)
    r1 = synth_fastq_records(rng, genomes[2:3] + genomes[0:1], 2500, 150, mate=0, lower_rate=lower)
    rng = np.random.default_rng(7)
    r2 = synth_fastq_records(rng, genomes[2:3] + genomes[0:1], 2500, 150, mate=1, lower_rate=lower)
    f1, f2 = str(d / f"{mode}_1.fastq.gz"), str(d / f"{mode}_2.fastq.gz")
    write_fastq_gz(f1, r1)
    write_fastq_gz(f2, r2)
    pe = mode.startswith("pe")
    km = orc.kmers_fq_pe_qual(f1, f2, 27, 15) if pe else orc.kmers_from_fq_qual(f1, 27, 15)
    if mode == "pe_auto":
        cutoff = km.auto_cutoff()
        flt = []
    else:
        cutoff = 1 if mode == "se_f1" else 0
        flt = ["-f", str(cutoff)]
    km = km.clean_map(cutoff)
    gene = mode == "se_gene"
    args = ["search", "-b", bxi, "-q", f1] + (["-r", f2] if pe else []) + flt + (["-g"] if gene else []) + ["-p", "0.02"]
    out, err = run(*args, host_kmers=host_kmers)
    want = expected_report(orc, oix, f1, km, 0.02, gene)
    assert sorted(out.splitlines()) == sorted(want) and len(want) >= 1
    assert f"{len(km)} k-mers in query" in err and "Search: " in err
    assert ("k-mer map on the host" in err) == (host_kmers or lower > 0)


def expected_readid(orc, oix, ids, reads, d, S, fp_correct=1e-3):
    bases, seq_off, read_seq0 = pack_reads(reads)
    rep, nk, st = oix.readid_counts(bases, seq_off, read_seq0, d, S)
    lines = []
    for i in range(len(reads)):
        if st[i] == 1:
            lines.append(f"{ids[i]}\ttoo_short\t0\t0\taccept\t0")
            continue
        lab, cnt, kl, verdict, ntop = oix.kmer_poll_plus(rep[i].astype(np.uint64), int(nk[i]), fp_correct)
        lines.append(f"{ids[i]}\t{lab}\t{cnt}\t{kl}\t{verdict}\t{ntop}")
    return lines


def counts_file(lines):
    c = {}
    for l in lines:
        v = l.split("\t")
        key = v[1] if v[4] == "accept" else "reject"
        c[key] = c.get(key, 0) + 1
    return sorted(f"{k}\t{n}" for k, n in c.items())


@pytest.mark.parametrize("pe,dflag,bflag,batch", [(False, 1, 3, 50000), (True, 1, 3, 50000), (False, 10, 3, 64), (True, 3, 0, 100),
                                                   (False, 1, 1, 7)])
def test_read_id_fastq(orc, env, pe, dflag, bflag, batch):
    d, bxi, oix, genomes = env
    rng = np.random.default_rng(11 + dflag)
    r1 = synth_fastq_records(rng, genomes, 700, 150, mate=0)
    rng = np.random.default_rng(11 + dflag)
    r2 = synth_fastq_records(rng, genomes, 700, 150, mate=1)
    tag = f"rid_{int(pe)}_{dflag}_{bflag}_{batch}"
    f1, f2 = str(d / f"{tag}_1.fastq.gz"), str(d / f"{tag}_2.fastq.gz")
    write_fastq_gz(f1, r1, multi_member=True)
    write_fastq_gz(f2, r2)
    prefix = str(d / tag)
    args = ["read_id", "-b", bxi, "-q", f1] + ([f2] if pe else []) + ["-n", prefix, "-d", str(dflag), "-B", str(bflag), "-c", str(batch)]
    out, err = run(*args)
    ids = ["@" + r[0].decode() for r in r1]
    masked = [[orc.qual_mask(a[1], a[2], 15)] + ([orc.qual_mask(b[1], b[2], 15)] if pe else []) for a, b in zip(r1, r2)]
    want = expected_readid(orc, oix, ids, masked, dflag, bflag)
    got = open(prefix + "_reads.txt").read().splitlines()
    assert got == want
    assert sorted(open(prefix + "_counts.txt").read().splitlines()) == counts_file(want)
    assert len({l.split("\t")[1] for l in want}) >= 4        # several labels incl. no_hits / phage names occur
    assert f"Classified {len(want)} read" in err


@pytest.mark.parametrize("pe", [False, True])
def test_read_id_threaded_host_pipeline_writes_the_same_rows(orc, env, pe):
    """Batches big enough for every host pool to split its work — several inflating threads (BGZF input), record-packing threads, a
    poll sliced over threads (>= 8192 reads per batch), batches recycled between the stages — against the same run on one thread of
    each, and against the oracle on a sample of the rows."""
    from test_linereader_cpu import write_bgzf
    d, bxi, oix, genomes = env
    n = 40_000
    files = []
    recs = []
    for mate in range(2 if pe else 1):
        rng = np.random.default_rng(77)
        r = synth_fastq_records(rng, genomes, n, 150, mate=mate)
        recs.append(r)
        text = b"".join(b"@" + a[0] + b"\n" + a[1] + b"\n+\n" + a[2] + b"\n" for a in r)
        f = str(d / f"thr_{int(pe)}_{mate}.fastq.gz")
        write_bgzf(f, text, level=1)
        files.append(f)
    outs = []
    for tag, envv in (("one", {"COLORID_GZ_THREADS": "1", "COLORID_PARSE_THREADS": "1", "COLORID_POLL_THREADS": "1"}),
                      ("many", {"COLORID_GZ_THREADS": "5", "COLORID_PARSE_THREADS": "4", "COLORID_POLL_THREADS": "4"}),
                      ("zlib", {"COLORID_LIBDEFLATE": "0", "COLORID_NO_AVX2": "1", "CID_PIN_STAGING": "0"}),          # every fallback at once
                      ("gpu", {"COLORID_GPU_INFLATE": "1", "COLORID_GPU_INFLATE_MB": "1"})):                          # members inflated on the GPU, several batches
        prefix = str(d / f"thr_{int(pe)}_{tag}")
        p = subprocess.run([BIN, "read_id", "-b", bxi, "-q", *files, "-n", prefix, "-c", "20000"], capture_output=True, text=True, env=dict(os.environ, **envv))
        assert p.returncode == 0, p.stderr[-500:]
        outs.append((open(prefix + "_reads.txt").read(), open(prefix + "_counts.txt").read()))
    assert outs[0] == outs[1] == outs[2] == outs[3]
    rows = outs[0][0].splitlines()
    assert len(rows) == n
    sample = list(range(0, n, 97))
    ids = ["@" + recs[0][i][0].decode() for i in sample]
    masked = [[orc.qual_mask(r[i][1], r[i][2], 15) for r in recs] for i in sample]
    want = expected_readid(orc, oix, ids, masked, 1, 3)
    assert [rows[i] for i in sample] == want


def test_read_id_fasta(orc, env):
    d, bxi, oix, genomes = env
    # stream_fasta keeps line breaks inside the sequence (read_id_mt_pe.rs:480-494): 70-column records
    rng = np.random.default_rng(5)
    recs = []
    for i in range(60):
        g = genomes[i % 4]
        st = int(rng.integers(0, len(g) - 400))
        s = g[st:st + int(rng.integers(30, 400))]
        recs.append((f">contig{i} len={len(s)}".encode(), b"".join(s[j:j + 70] + b"\n" for j in range(0, len(s), 70))))
    q = d / "contigs.fasta"
    q.write_bytes(b"".join(h + b"\n" + body for h, body in recs))
    prefix = str(d / "rid_fasta")
    run("read_id", "-b", bxi, "-q", str(q), "-n", prefix, "-B", "0")
    ids = [h.decode() for h, _ in recs]
    want = expected_readid(orc, oix, ids, [[body] for _, body in recs], 1, 0)
    assert open(prefix + "_reads.txt").read().splitlines() == want


def _stream_fasta_records(text: bytes):
    """the reference's line loop (read_id_mt_pe.rs:450-569) restated: the first line is the first id (its last byte dropped); a later line
    holding a '>' ANYWHERE closes the record before it if that record has any sequence yet (else the line is dropped, the id stays); every
    other line joins the sequence WITH its line end; the end of the file closes the last record whatever it holds"""
    lines = text.splitlines(keepends=True)
    recs, cur_id, sub = [], b"", b""
    for i, line in enumerate(lines):
        if i == 0:
            cur_id = line[:-1]
        elif b">" in line:
            if sub:
                recs.append((cur_id, sub))
                cur_id, sub = line[:-1], b""
        else:
            sub += line
    recs.append((cur_id, sub))
    return recs


@pytest.mark.parametrize("batch", ["3", "50000"])
def test_read_id_fasta_line_loop_edges(orc, env, batch):
    """files the loop treats in its own way: headers in a row, a '>' inside a sequence line, a last line without its line end, CRLF, a
    first line that is no header, a header at the very end; batches of three reads (records closed at a batch's edge)"""
    d, bxi, oix, genomes = env
    g = genomes[0]
    cases = {
        "plain": b">a\n" + g[:300] + b"\n>b\n" + g[300:500] + b"\n" + g[500:640] + b"\n>c x\n" + g[1000:1200] + b"\n",
        "headers_in_a_row": b">a\n>b\n>c\n" + g[:200] + b"\n>d\n>e\n" + g[200:420] + b"\n",
        "gt_inside": b">a\n" + g[:100] + b"\n" + g[100:150] + b">" + g[150:200] + b"\n" + g[200:330] + b"\n>z\n" + g[400:600] + b"\n",
        "no_last_newline": b">a\n" + g[:250] + b"\n>b\n" + g[300:533],
        "crlf": b">a\r\n" + g[:120] + b"\r\n" + g[120:260] + b"\r\n>b\r\n" + g[300:480] + b"\r\n",
        "no_header_first": g[:90] + b"\n" + g[90:300] + b"\n>b\n" + g[300:500] + b"\n",
        "header_last": b">a\n" + g[:200] + b"\n>b\n",
        "seven": b"".join(b">r%d\n" % i + g[i * 100:i * 100 + 150 + i] + b"\n" for i in range(7)),
    }
    for name, text in cases.items():
        q = d / f"edge_{name}.fasta"
        q.write_bytes(text)
        prefix = str(d / f"rid_edge_{name}_{batch}")
        run("read_id", "-b", bxi, "-q", str(q), "-n", prefix, "-B", "0", "-c", batch)
        recs = _stream_fasta_records(text)
        want = expected_readid(orc, oix, [i.decode() for i, _ in recs], [[s] for _, s in recs], 1, 0)
        assert open(prefix + "_reads.txt", newline="").read().split("\n")[:-1] == want, name   # (an id keeps its '\r': rows end with '\n' only)


def test_batch_id_sample_sheet(orc, env, tmp_path):
    """batch_id (main.rs:869-888, read_id_batch.rs:7-181): a sheet `name \\t reads1 [\\t reads2]`, the index loaded once, every sample
    through read_id's streamers under the prefix NAME_TAG: single-stream gzip single-end, block-gzip pairs (the device FASTQ front
    end), a FASTA sample, a name given twice (the later line wins, build.rs:15-31) — each sample's two files against the oracle and
    against what `read_id -n` writes for the same files."""
    from test_linereader_cpu import write_bgzf
    d, bxi, oix, genomes = env
    rng = np.random.default_rng(401)
    se = synth_fastq_records(rng, genomes, 900, 150, mate=0)
    rng = np.random.default_rng(402)
    p1 = synth_fastq_records(rng, genomes, 1200, 150, mate=0)
    rng = np.random.default_rng(402)
    p2 = synth_fastq_records(rng, genomes, 1200, 150, mate=1)
    f_se = str(tmp_path / "s_se.fastq.gz")
    write_fastq_gz(f_se, se, multi_member=True)
    f_p = []
    for mate, recs in enumerate((p1, p2)):
        f = str(tmp_path / f"s_pe_{mate + 1}.fastq.gz")
        write_bgzf(f, b"".join(b"@" + a[0] + b"\n" + a[1] + b"\n+\n" + a[2] + b"\n" for a in recs), level=6)
        f_p.append(f)
    fa = []
    for i in range(40):
        g = genomes[i % 4]
        st = int(rng.integers(0, len(g) - 400))
        fa.append((f">ctg{i}".encode(), g[st:st + int(rng.integers(30, 400))] + b"\n"))
    f_fa = str(tmp_path / "s_contigs.fasta")
    open(f_fa, "wb").write(b"".join(h + b"\n" + body for h, body in fa))
    sheet = tmp_path / "samples.tsv"
    sheet.write_text(f"zeta\t{f_fa}\nalpha\t{f_fa}\nmid\t{f_p[0]}\t{f_p[1]}\nalpha\t{f_se}\n")     # alpha: the later line replaces the earlier
    out, err = run("batch_id", "-b", bxi, "-q", str(sheet), "-T", "run7", "-c", "500", cwd=str(tmp_path))
    assert [l for l in err.splitlines() if l.startswith("Classifying ")] == ["Classifying alpha", "Classifying mid", "Classifying zeta"]
    assert err.count("Index loaded in") == 1
    want = {
        "alpha": expected_readid(orc, oix, ["@" + r[0].decode() for r in se], [[orc.qual_mask(a[1], a[2], 15)] for a in se], 1, 3),
        "mid": expected_readid(orc, oix, ["@" + r[0].decode() for r in p1],
                               [[orc.qual_mask(a[1], a[2], 15), orc.qual_mask(b[1], b[2], 15)] for a, b in zip(p1, p2)], 1, 3),
        "zeta": expected_readid(orc, oix, [h.decode() for h, _ in fa], [[body] for _, body in fa], 1, 3),
    }
    queries = {"alpha": [f_se], "mid": f_p, "zeta": [f_fa]}
    for name, rows in want.items():
        got = open(tmp_path / f"{name}_run7_reads.txt").read()
        assert got.splitlines() == rows
        counts = open(tmp_path / f"{name}_run7_counts.txt").read()
        assert sorted(counts.splitlines()) == counts_file(rows)
        single = str(tmp_path / f"single_{name}")
        run("read_id", "-b", bxi, "-q", *queries[name], "-n", single, "-c", "500")
        assert open(single + "_reads.txt").read() == got and open(single + "_counts.txt").read() == counts
    # a sheet needs -T, and a sample whose file is missing ends the run with an error (the reference panics on the open)
    p = subprocess.run([BIN, "batch_id", "-b", bxi, "-q", str(sheet)], capture_output=True, text=True)
    assert p.returncode != 0 and "--tag" in p.stderr
    bad = tmp_path / "bad.tsv"
    bad.write_text(f"gone\t{tmp_path}/nothing_here.fastq.gz\n")
    p = subprocess.run([BIN, "batch_id", "-b", bxi, "-q", str(bad), "-T", "x"], capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode != 0


def test_k_above_32_uses_host_map(orc, env, tmp_path):
    """k > 32 cannot be packed in 64 bits: build / search / read_id use byte-string k-mers — counted on the GPU (sorted on a
    4-bit-per-base image), or in the host map with COLORID_HOST_KMERS=1; read_id takes the LDS byte path."""
    d, _, _, genomes = env
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES[:3]))
    pre = str(tmp_path / "k35")
    run("build", "-s", "300007", "-n", "3", "-k", "35", "-b", pre, "-r", str(tsv))
    oix = orc.Index.build_single(str(tsv), 300007, 3, 35)
    ref = pre + "_oracle.bxi"
    oix.save(ref)
    assert open(pre + ".bxi", "rb").read() == open(ref, "rb").read()
    q = os.path.join(REFS, PHAGES[1] + ".fasta")
    out, err = run("search", "-b", pre + ".bxi", "-q", q, "-g", "-p", "0.01")
    km = orc.Kmers(35)
    for s in orc.read_fasta(q):
        km.kmerize_vector(s, 1)
    assert sorted(out.splitlines()) == sorted(expected_report(orc, oix, q, km.clean_map(0), 0.01, True))
    assert "k-mer map on the host" not in err
    out2, err2 = run("search", "-b", pre + ".bxi", "-q", q, "-g", "-p", "0.01", host_kmers=True)
    assert sorted(out2.splitlines()) == sorted(out.splitlines()) and "k-mer map on the host" in err2
    out3, _ = run("search", "-b", pre + ".bxi", "-q", q, "-p", "0.01")                     # default report: per-k-mer unique colours
    out4, _ = run("search", "-b", pre + ".bxi", "-q", q, "-p", "0.01", host_kmers=True)
    assert sorted(out3.splitlines()) == sorted(out4.splitlines()) == sorted(expected_report(orc, oix, q, km.clean_map(0), 0.01, False))
    run("build", "-s", "300007", "-n", "3", "-k", "35", "-b", pre + "_host", "-r", str(tsv), host_kmers=True)
    assert open(pre + "_host.bxi", "rb").read() == open(ref, "rb").read()
    rng = np.random.default_rng(4)
    r1 = synth_fastq_records(rng, genomes[:3], 300, 150, mate=0)
    f1 = str(tmp_path / "r.fastq.gz")
    write_fastq_gz(f1, r1)
    run("read_id", "-b", pre + ".bxi", "-q", f1, "-n", str(tmp_path / "rid"))
    want = expected_readid(orc, oix, ["@" + r[0].decode() for r in r1], [[orc.qual_mask(a[1], a[2], 15)] for a in r1], 1, 3)
    assert open(tmp_path / "rid_reads.txt").read().splitlines() == want


def test_minimizer_index_cli(orc, env, tmp_path):
    """`build -m -v 15` -> .mxi (BigsyMapMiniNew), `info`, `read_id`, and `search` refusing it (src/main.rs:485-522, 569-573, 645-648)."""
    d, _, _, genomes = env
    tsv = str(d / "ref_file.txt")
    pre = str(tmp_path / "mini")
    out, err = run("build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", tsv, "-m", "-v", "15")
    assert "Build with minimizers, minimizer size: 15" in out
    oix = orc.Index.build_single_mini(tsv, 750000, 4, 27, 15)
    ref = str(tmp_path / "oracle.mxi")
    oix.save(ref)
    assert open(pre + ".mxi", "rb").read() == open(ref, "rb").read()
    run("build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre + "_h", "-r", tsv, "-m", host_kmers=True)   # -v defaults to 15
    assert open(pre + "_h.mxi", "rb").read() == open(ref, "rb").read()
    out, _ = run("info", "-b", pre + ".mxi")
    assert out.splitlines()[:7] == ["BIGSI parameters:", "Bloomfilter-size: 750000", "Number of hashes: 4", "K-mer size: 27",
                                    " minimizer size: 15", "", "Number of accessions in index: 4"]
    out, err = run("search", "-b", pre + ".mxi", "-q", os.path.join(REFS, PHAGES[0] + ".fasta"), "-s")
    assert out == "" and "An index with minimizers (.mxi) is used, but not available for this function" in err
    for pe in (False, True):
        rng = np.random.default_rng(21)
        r1 = synth_fastq_records(rng, genomes, 500, 150, mate=0)
        rng = np.random.default_rng(21)
        r2 = synth_fastq_records(rng, genomes, 500, 150, mate=1)
        f1, f2 = str(tmp_path / f"m{int(pe)}_1.fastq.gz"), str(tmp_path / f"m{int(pe)}_2.fastq.gz")
        write_fastq_gz(f1, r1)
        write_fastq_gz(f2, r2)
        prefix = str(tmp_path / f"rid{int(pe)}")
        run("read_id", "-b", pre + ".mxi", "-q", f1, *([f2] if pe else []), "-n", prefix)
        ids = ["@" + r[0].decode() for r in r1]
        masked = [[orc.qual_mask(a[1], a[2], 15)] + ([orc.qual_mask(b[1], b[2], 15)] if pe else []) for a, b in zip(r1, r2)]
        want = expected_readid(orc, oix, ids, masked, 1, 3)
        assert open(prefix + "_reads.txt").read().splitlines() == want
        assert sum("Listeria_phage" in l for l in want) > 300
