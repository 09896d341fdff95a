"""cid_fastq — the FASTQ front end of read_id on the device (inflate -> lines -> records -> qual_mask -> packed reads -> k_readid) —
against the host's way of doing the same: records cut out of the text by a restatement of the reference's line loops
(read_id_mt_pe.rs:862-879 single-end, :927-975 pairs: four lines per record, lines() strips "\\n" and "\\r\\n", an unterminated last line
counts, leftovers are dropped, the walk ends with the shorter file), quality-masked by the oracle's qual_mask (seq.rs:36-56), packed and
classified through cid_readid_count_sparse — and against the oracle's own per-read counts.  The text reaches the device cut at arbitrary
points (push_text) and as block-gzip members whose boundaries fall inside records (push_bgzf)."""
import numpy as np
import pytest

from test_gpu_inflate import bgzf_member
from test_gpu_readid import pack_reads
from util import random_index, synth_fastq_records, to_hip_index

pytestmark = pytest.mark.gpu


def fastq_text(records, eol=b"\n", last_newline=True, plus=b"+"):
    out = bytearray()
    for i, (rid, seq, qual) in enumerate(records):
        out += b"@" + rid + eol + seq + eol + plus + eol + qual
        if i + 1 < len(records) or last_newline:
            out += eol
    return bytes(out)


def line_loop_records(text):
    """the reference's reader: lines (\\n or \\r\\n stripped; an unterminated last line too), four per record, leftovers dropped"""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines]
    return [(lines[i], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


def expected(orc, hx, texts, q, d, S):
    recs = [line_loop_records(t) for t in texts]
    n = min(len(r) for r in recs)
    reads = [[orc.qual_mask(r[i][1], r[i][2], q) for r in recs] for i in range(n)]
    ids = [recs[0][i][0] for i in range(n)]
    bases, so, r0 = pack_reads(reads)
    rs, col, cnt, nk, st = hx.readid_count_sparse(bases, so, r0, d, S)
    return ids, (nk, st, rs, col, cnt), (bases, so, r0)


def collect(fr, hx, d, S, acc):
    ids, nk, st, rs, col, cnt = fr.classify(hx, d, S)
    acc["ids"] += ids
    acc["nk"].append(nk); acc["st"].append(st)
    for r in range(len(ids)):
        acc["rows"].append(list(zip(col[int(rs[r]):int(rs[r + 1])].tolist(), cnt[int(rs[r]):int(rs[r + 1])].tolist())))


def check_equal(acc, want_ids, want):
    nk, st, rs, col, cnt = want
    assert acc["ids"] == want_ids
    assert np.array_equal(np.concatenate(acc["nk"]) if acc["nk"] else np.zeros(0, np.uint32), nk)
    assert np.array_equal(np.concatenate(acc["st"]) if acc["st"] else np.zeros(0, np.uint8), st)
    rows = [list(zip(col[int(rs[r]):int(rs[r + 1])].tolist(), cnt[int(rs[r]):int(rs[r + 1])].tolist())) for r in range(len(want_ids))]
    assert acc["rows"] == rows


@pytest.fixture(scope="module")
def world(orc, hip_ctx):
    rng = np.random.default_rng(11)
    genomes = [bytes(rng.choice(list(b"ACGT"), size=6000).astype(np.uint8)) for _ in range(5)]
    oix = random_index(orc, rng, 40_009, 3, 21, 130, density=0.02, zero_row_frac=0.2)
    for gi, g in enumerate(genomes):
        km = orc.Kmers(21)
        km.kmerize_vector(g, 1)
        for key in km.keys():
            oix.insert(gi * 7, key.tobytes())
    hx = to_hip_index(hip_ctx, oix)
    yield oix, hx, genomes
    hx.close()


@pytest.mark.parametrize("seed", range(8))
def test_line_ends_at_every_alignment(orc, hip_ctx, world, seed):
    """the front end's own line-end kernels (k_nl_count / k_nl_positions: 16-byte pieces, 1 KiB rounds, 8 KiB chunks): names of random
    length put the newlines at every offset inside a piece and the texts' ends at every distance from a round's and a chunk's end;
    empty lines, a text of a single byte, texts cut so that a step sees 0 ... 3 dangling lines."""
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(900 + seed)
    recs = []
    for i in range(int(rng.integers(1, 400))):
        name = b"r%d" % i + b"n" * int(rng.integers(0, 40))
        L = int(rng.choice([0, 1, 15, 16, 17, 21, 150, 1000]))
        g = genomes[int(rng.integers(len(genomes)))]
        recs.append((name, g[:L], b"I" * L))
    text = fastq_text(recs, b"\n", bool(seed % 2))
    want_ids, want, _ = expected(orc, hx, [text], 0, 1, 3)
    for pieces in (1, 2, 13):
        fr = colorid_amd.FastqReader(hip_ctx, 1, 0)
        cuts = sorted(rng.integers(0, len(text) + 1, pieces - 1).tolist()) if pieces > 1 else []
        acc = {"ids": [], "nk": [], "st": [], "rows": []}
        prev = 0
        for cut in cuts + [len(text)]:
            fr.push_text(0, text[prev:cut], last=(cut == len(text)))
            prev = cut
            collect(fr, hx, 1, 3, acc)                                # a step after every push: carries of every length
        check_equal(acc, want_ids, want)
        fr.close()


@pytest.mark.parametrize("eol,last_newline", [(b"\n", True), (b"\r\n", True), (b"\n", False), (b"\r\n", False)])
@pytest.mark.parametrize("q", [0, 15])
def test_text_pushed_in_pieces_single_end(orc, hip_ctx, world, eol, last_newline, q):
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(len(eol) * 10 + q + int(last_newline))
    recs = synth_fastq_records(rng, genomes, 900, 150, lower_rate=0.0)
    recs.insert(5, (b"empty read", b"", b""))
    recs.insert(9, (b"read with\ttab and a very long name " + b"x" * 300, genomes[0][:150], b"I" * 150))
    text = fastq_text(recs, eol, last_newline) + (b"" if last_newline else b"")
    if last_newline:
        text += b"@dangling header" + eol + b"ACGT" + eol           # two lines that complete no record: dropped
    want_ids, want, packed = expected(orc, hx, [text], q, 1, 3)
    assert len(want_ids) == len(recs)
    # the oracle itself on the host-packed reads
    orep = oix.readid_counts(*packed, 1, 3)
    grep = hx.readid_count(*packed, 1, 3)
    assert all(np.array_equal(a, b) for a, b in zip(grep, orep))
    for pieces in (1, 7, 60):
        fr = colorid_amd.FastqReader(hip_ctx, 1, q)
        cuts = sorted(rng.integers(0, len(text), pieces - 1).tolist()) if pieces > 1 else []
        acc = {"ids": [], "nk": [], "st": [], "rows": []}
        prev = 0
        for j, cut in enumerate(cuts + [len(text)]):
            fr.push_text(0, text[prev:cut], last=(cut == len(text)))
            prev = cut
            if j % 3 == 2 or cut == len(text):
                collect(fr, hx, 1, 3, acc)
        check_equal(acc, want_ids, want)
        ids, nk, st, rs, col, cnt = fr.classify(hx, 1, 3)           # nothing is left
        assert ids == [] and len(nk) == 0
        fr.close()


def test_step_in_two_halves_keeps_the_results_of_the_step_before(orc, hip_ctx, world):
    """cid_fastq_classify_begin / _end: the results of step i are fetched AFTER step i + 1 has begun (its classifier in flight, more text
    pushed meanwhile) and still are step i's; fetched twice they are the same; _end without _begin and two _begins in a row are errors."""
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(77)
    recs = synth_fastq_records(rng, genomes, 1500, 150, lower_rate=0.0)
    text = fastq_text(recs)
    want_ids, want, _ = expected(orc, hx, [text], 15, 1, 3)
    fr = colorid_amd.FastqReader(hip_ctx, 1, 15)
    with pytest.raises(colorid_amd.CidError):
        fr.classify_end()
    cuts = sorted(rng.integers(0, len(text), 5).tolist()) + [len(text)]
    acc = {"ids": [], "nk": [], "st": [], "rows": []}

    def take(sizes):
        ids, nk, st, rs, col, cnt = fr.fetch(sizes)
        again = fr.fetch(sizes)
        assert again[0] == ids and all(np.array_equal(a, b) for a, b in zip(again[1:], (nk, st, rs, col, cnt)))
        acc["ids"] += ids
        acc["nk"].append(nk); acc["st"].append(st)
        for r in range(len(ids)):
            acc["rows"].append(list(zip(col[int(rs[r]):int(rs[r + 1])].tolist(), cnt[int(rs[r]):int(rs[r + 1])].tolist())))

    fr.push_text(0, text[:cuts[0]])
    pending = None
    for j in range(len(cuts)):
        fr.classify_begin(hx, 1, 3)
        if j == 0:
            with pytest.raises(colorid_amd.CidError):
                fr.classify_begin(hx, 1, 3)
        if j + 1 < len(cuts):
            fr.push_text(0, text[cuts[j]:cuts[j + 1]], last=(j + 2 == len(cuts)))     # the next piece goes up while the step is in flight
        if pending is not None:
            take(pending)                                                                # the step BEFORE, fetched beside this one's classifier
        pending = fr.classify_end()
    take(pending)
    check_equal(acc, want_ids, want)
    fr.close()


@pytest.mark.parametrize("q,d,S", [(15, 1, 3), (0, 2, 0), (20, 1, 5)])
def test_pairs_from_two_files(orc, hip_ctx, world, q, d, S):
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(q + d)
    r1 = synth_fastq_records(np.random.default_rng(5 + q), genomes, 700, 140, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(5 + q), genomes, 700, 140, mate=1, lower_rate=0.0)
    t1 = fastq_text(r1, b"\n", True)
    t2 = fastq_text(r2[:650], b"\r\n", False)                        # the second file is shorter: the walk ends with it
    want_ids, want, packed = expected(orc, hx, [t1, t2], q, d, S)
    assert len(want_ids) == 650
    orep = oix.readid_counts(*packed, d, S)
    grep = hx.readid_count(*packed, d, S)
    assert all(np.array_equal(a, b) for a, b in zip(grep, orep))
    fr = colorid_amd.FastqReader(hip_ctx, 2, q)
    acc = {"ids": [], "nk": [], "st": [], "rows": []}
    c1 = sorted(rng.integers(0, len(t1), 9).tolist()) + [len(t1)]
    c2 = sorted(rng.integers(0, len(t2), 4).tolist()) + [len(t2)]
    p1 = p2 = 0
    i1 = i2 = 0
    while i1 < len(c1) or i2 < len(c2):                              # the two files advance at different paces
        if i1 < len(c1):
            fr.push_text(0, t1[p1:c1[i1]], last=(i1 == len(c1) - 1)); p1 = c1[i1]; i1 += 1
        if i2 < len(c2) and (i1 % 2 == 0 or i1 >= len(c1)):
            fr.push_text(1, t2[p2:c2[i2]], last=(i2 == len(c2) - 1)); p2 = c2[i2]; i2 += 1
        collect(fr, hx, d, S, acc)
    check_equal(acc, want_ids, want)
    fr.close()


def test_block_gzip_members_cut_records(orc, hip_ctx, world):
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(3)
    recs = synth_fastq_records(rng, genomes, 3000, 150, lower_rate=0.0)
    text = fastq_text(recs)
    want_ids, want, _ = expected(orc, hx, [text], 15, 1, 3)
    # members of irregular sizes (1 byte .. 64 KiB), an empty one, every compression level: record and line boundaries fall anywhere
    members, lens, pos = [], [], 0
    while pos < len(text):
        n = int(rng.choice([1, 7, 300, 5000, 30000, 65536]))
        chunk = text[pos:pos + n]
        members.append(bgzf_member(chunk, level=int(rng.integers(0, 10)), extra_subfield=bool(rng.integers(0, 2))))
        lens.append(len(chunk))
        pos += n
        if rng.random() < 0.05:
            members.append(bgzf_member(b"")); lens.append(0)
    members.append(bgzf_member(b"")); lens.append(0)                 # the BGZF end marker
    fr = colorid_amd.FastqReader(hip_ctx, 1, 15)
    acc = {"ids": [], "nk": [], "st": [], "rows": []}
    i = 0
    while i < len(members):
        j = min(len(members), i + int(rng.integers(1, 12)))
        fr.push_bgzf(0, members[i:j], lens[i:j], last=(j == len(members)))
        collect(fr, hx, 1, 3, acc)
        i = j
    check_equal(acc, want_ids, want)
    fr.close()
    # the way the CLI drives it: stretch i + 1 is pushed (and inflates on the reader's own stream) before stretch i is classified
    fr = colorid_amd.FastqReader(hip_ctx, 1, 15)
    acc = {"ids": [], "nk": [], "st": [], "rows": []}
    cuts = list(range(0, len(members), 9)) + [len(members)]
    waiting = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        fr.push_bgzf(0, members[a:b], lens[a:b], last=(b == len(members)))
        waiting += 1
        if waiting == 3:
            ids, nk, st, rs, col, cnt = fr.classify(hx, 1, 3, max_pushes=1)
            acc["ids"] += ids; acc["nk"].append(nk); acc["st"].append(st)
            acc["rows"] += [list(zip(col[int(rs[r]):int(rs[r + 1])].tolist(), cnt[int(rs[r]):int(rs[r + 1])].tolist())) for r in range(len(ids))]
            waiting -= 1
    while waiting:
        ids, nk, st, rs, col, cnt = fr.classify(hx, 1, 3, max_pushes=1)
        acc["ids"] += ids; acc["nk"].append(nk); acc["st"].append(st)
        acc["rows"] += [list(zip(col[int(rs[r]):int(rs[r + 1])].tolist(), cnt[int(rs[r]):int(rs[r + 1])].tolist())) for r in range(len(ids))]
        waiting -= 1
    check_equal(acc, want_ids, want)
    with pytest.raises(colorid_amd.CidError):
        fr.push_bgzf(0, members[:1], lens[:1])                       # the file was closed by its last push
    fr.close()
    # a corrupt member is named; a quality line longer than its sequence is the reference's panic
    fr = colorid_amd.FastqReader(hip_ctx, 1, 15)
    bad = bytearray(members[2]); bad[-6] ^= 1
    fr.push_bgzf(0, members[:2] + [bytes(bad)], lens[:3])
    with pytest.raises(colorid_amd.CidError) as ei:
        fr.classify(hx)
    assert ei.value.code == -1 and "member 2" in str(ei.value) and "CRC-32" in str(ei.value)
    fr.close()
    fr = colorid_amd.FastqReader(hip_ctx, 1, 15)
    fr.push_text(0, b"@r\nACGT\n+\nIIIIII\n", last=True)
    with pytest.raises(colorid_amd.CidError) as ei:
        fr.classify(hx)
    assert ei.value.code == -1 and "next nt" in str(ei.value)
    fr.close()
    fr = colorid_amd.FastqReader(hip_ctx, 1, 0)                      # without masking the same record is fine: the sequence as it is
    fr.push_text(0, b"@r\nACGT\n+\nIIIIII\n", last=True)
    ids, nk, st, *_ = fr.classify(hx)
    assert ids == [b"@r"] and st[0] == 1                             # 4 bases < k: too_short
    fr.close()


def _long_records(rng, genomes, soft_mask=True):
    """records of 150 b ... 60 kb in one file: short reads between long ones, an empty read, a read of several hash buckets, and (soft_mask)
    two reads with a lower-case stretch — their case is kept, they alone take the byte-string path"""
    big = b"".join(genomes) * 3                                       # 90 kb with repeats
    recs = []
    for i, L in enumerate([150, 3_000, 150, 12_000, 900, 40_000, 150, 1_100, 60_000, 0, 9_999, 150, 20_000, 2_500, 150]):
        st = int(rng.integers(0, len(big) - L)) if L else 0
        seq = bytearray(big[st:st + L])
        for _ in range(L // 2_000):
            seq[int(rng.integers(0, L))] = ord("N")
        if soft_mask and i in (3, 7):
            a = int(rng.integers(0, L - 60)); seq[a:a + 50] = bytes(seq[a:a + 50]).lower()
        qual = bytes(rng.choice(list(b"#5I"), size=L, p=[0.02, 0.08, 0.9]).astype(np.uint8)) if L else b""
        recs.append((b"read%d len=%d" % (i, L), bytes(seq), qual))
    return recs


@pytest.mark.parametrize("q", [0, 15])
@pytest.mark.parametrize("paired", [False, True])
def test_long_reads_through_the_device_front_end(orc, hip_ctx, world, q, paired):
    """Round 6: records of any length through cid_fastq_classify — until round 5 a read that did not fit a wave's LDS was refused
    (CID_ERR_UNSUPPORTED) and the CLI re-ran the whole input on the host.  Text cut anywhere, block-gzip members cutting records, pairs."""
    import colorid_amd
    oix, hx, genomes = world
    rng = np.random.default_rng(50 + q + int(paired))
    texts = [fastq_text(_long_records(rng, genomes))]
    if paired:
        texts.append(fastq_text(_long_records(rng, genomes, soft_mask=False)[:13], b"\r\n"))
    want_ids, want, packed = expected(orc, hx, texts, q, 1, 3)
    assert len(want_ids) == (13 if paired else 15)
    orep = oix.readid_counts(*packed, 1, 3, n_threads=8)               # the oracle itself on the host-packed reads
    grep = hx.readid_count(*packed, 1, 3)
    assert all(np.array_equal(a, b) for a, b in zip(grep, orep))
    nf = len(texts)
    for pieces in (1, 9):
        fr = colorid_amd.FastqReader(hip_ctx, nf, q)
        acc = {"ids": [], "nk": [], "st": [], "rows": []}
        cuts = [sorted(rng.integers(0, len(t), pieces - 1).tolist()) + [len(t)] for t in texts]
        prev = [0] * nf
        for j in range(pieces):
            for f in range(nf):
                fr.push_text(f, texts[f][prev[f]:cuts[f][j]], last=(j == pieces - 1)); prev[f] = cuts[f][j]
            collect(fr, hx, 1, 3, acc)
        check_equal(acc, want_ids, want)
        fr.close()
    # as block-gzip members
    fr = colorid_amd.FastqReader(hip_ctx, nf, q)
    acc = {"ids": [], "nk": [], "st": [], "rows": []}
    for f in range(nf):
        members, lens, pos = [], [], 0
        while pos < len(texts[f]):
            n = int(rng.choice([3_000, 30_000, 65_280]))
            members.append(bgzf_member(texts[f][pos:pos + n], level=int(rng.integers(1, 7)))); lens.append(len(texts[f][pos:pos + n])); pos += n
        members.append(bgzf_member(b"")); lens.append(0)
        half = len(members) // 2
        fr.push_bgzf(f, members[:half], lens[:half])
        fr.push_bgzf(f, members[half:], lens[half:], last=True)
    collect(fr, hx, 1, 3, acc)
    check_equal(acc, want_ids, want)
    fr.close()


@pytest.mark.parametrize("kind", ["bgzf", "fasta"])
def test_cli_read_id_long_reads_no_restart(orc, tmp_path, kind):
    """`colorid read_id` on long reads — a block-gzip FASTQ of 150 b ... 60 kb records (two of them soft-masked) through the device front
    end, without giving way to the host front end (round 5 drained, truncated and re-ran the whole input), and a plain FASTA of long
    records through the host reader: both equal the host front end's / the oracle's rows."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "phage")
    p = subprocess.run([BIN, "build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(15)
    recs = _long_records(rng, [g[:30_000] for g in genomes[:3]]) * 3
    recs = [(b"%s copy%d" % (r[0], i), r[1], r[2]) for i, r in enumerate(recs)]
    if kind == "bgzf":
        f1 = str(tmp_path / "long.fastq.gz")
        _write_bgzf(f1, fastq_text(recs), rng)
    else:
        f1 = str(tmp_path / "long.fasta")
        with open(f1, "wb") as f:
            for rid, seq, _ in recs:
                f.write(b">" + rid + b"\n" + seq + b"\n")
    outs = {}
    for tag, env in (("host", {"COLORID_DEVICE_FASTQ": "0"}), ("dev", {"COLORID_DEVICE_FASTQ_MB": "1"}), ("dev_default", {})):
        name = str(tmp_path / tag)
        p = subprocess.run([BIN, "read_id", "-b", pre + ".bxi", "-q", f1, "-n", name, "-Q", "0"], capture_output=True, text=True,
                           env=dict(os.environ, COLORID_TIMING="1", **env))
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        outs[tag] = (open(name + "_reads.txt").read(), open(name + "_counts.txt").read(), p.stderr)
    assert outs["host"][0].count("\n") == len(recs)
    for tag in ("dev", "dev_default"):
        assert outs[tag][0] == outs["host"][0] and outs[tag][1] == outs["host"][1], tag
        assert "host front end" not in outs[tag][2], outs[tag][2][-1500:]          # neither "using the" nor "starting over with the"
        if kind == "bgzf":
            assert "device front end" in outs[tag][2]
    assert "accept" in outs["host"][0]


def _write_bgzf(path, text, rng):
    """block-gzip file: members of irregular sizes (records and lines cut anywhere) + the end marker"""
    with open(path, "wb") as f:
        pos = 0
        while pos < len(text):
            n = int(rng.choice([200, 5000, 40000, 65280]))
            f.write(bgzf_member(text[pos:pos + n], level=int(rng.integers(1, 7))))
            pos += n
        f.write(bgzf_member(b""))


@pytest.mark.parametrize("paired", [False, True])
def test_cli_read_id_device_front_end_equals_host_front_end(orc, tmp_path, paired):
    """`colorid read_id` on block-gzip input: the device front end (default on one GPU) writes the same _reads.txt / _counts.txt as the host
    front end (COLORID_DEVICE_FASTQ=0), single-end and paired, several stretches per file (COLORID_DEVICE_FASTQ_MB=1), -Q and -d."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "phage")
    p = subprocess.run([BIN, "build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(8)
    r1 = synth_fastq_records(np.random.default_rng(21), genomes, 12000, 150, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(21), genomes, 12000, 150, mate=1, lower_rate=0.0)
    f1, f2 = str(tmp_path / "r_1.fastq.gz"), str(tmp_path / "r_2.fastq.gz")
    _write_bgzf(f1, fastq_text(r1), rng)
    _write_bgzf(f2, fastq_text(r2[:11500], b"\r\n", False), rng)
    q = [f1, f2] if paired else [f1]
    outs = {}
    for tag, env in (("host", {"COLORID_DEVICE_FASTQ": "0"}), ("dev", {"COLORID_DEVICE_FASTQ": "1"}), ("dev_small", {"COLORID_DEVICE_FASTQ": "1", "COLORID_DEVICE_FASTQ_MB": "1"}),
                     ("dev_ahead", {"COLORID_DEVICE_FASTQ": "1", "COLORID_DEVICE_FASTQ_MB": "1", "COLORID_DEVICE_FASTQ_AHEAD": "3"}), ("dev_default", {}),
                     ("dev_gpu_inflate", {"COLORID_DEVICE_FASTQ": "1", "COLORID_DEVICE_FASTQ_HOST_SHARE": "0"}), ("dev_host_inflate", {"COLORID_DEVICE_FASTQ": "1", "COLORID_DEVICE_FASTQ_HOST_SHARE": "1"}),
                     # the stretches' bytes page-locked, their copies running beside the loop (CID_FASTQ_KEEP on cid_fastq_push_bgzf), buffers in turn
                     ("dev_pinned", {"COLORID_DEVICE_FASTQ_PINNED": "1", "COLORID_DEVICE_FASTQ_MB": "1", "COLORID_DEVICE_FASTQ_HOST_SHARE": "0"}),
                     ("dev_pinned_ahead", {"COLORID_DEVICE_FASTQ_PINNED": "1", "COLORID_DEVICE_FASTQ_MB": "1", "COLORID_DEVICE_FASTQ_AHEAD": "3"}),
                     ("dev_behind", {"CID_FASTQ_INFLATE_BESIDE": "0", "COLORID_DEVICE_FASTQ_MB": "1", "COLORID_DEVICE_FASTQ_HOST_SHARE": "0"})):   # the inflate on the classifier's stream
        for extra in ([], ["-Q", "0", "-d", "3", "-B", "0"]):
            name = str(tmp_path / f"{tag}{len(extra)}")
            p = subprocess.run([BIN, "read_id", "-b", pre + ".bxi", "-q", *q, "-n", name, *extra], capture_output=True, text=True,
                               env=dict(os.environ, COLORID_TIMING="1", **env))
            assert p.returncode == 0, p.stderr[-2000:]
            outs[(tag, len(extra))] = (open(name + "_reads.txt").read(), open(name + "_counts.txt").read(), p.stderr)
    for extra in (0, 6):
        host = outs[("host", extra)]
        assert host[0].count("\n") == (11500 if paired else 12000)
        for tag in ("dev", "dev_small", "dev_ahead", "dev_default", "dev_gpu_inflate", "dev_host_inflate", "dev_pinned", "dev_pinned_ahead", "dev_behind"):
            assert outs[(tag, extra)][0] == host[0] and outs[(tag, extra)][1] == host[1], (tag, extra)
            assert "device front end" in outs[(tag, extra)][2], tag      # (the timing line of the path that ran)
    assert "accept" in outs[("dev", 0)][0]


@pytest.mark.parametrize("paired", [False, True])
def test_cli_read_id_device_front_end_gives_way_mid_run(orc, tmp_path, paired):
    """A stretch the device front end refuses (a read too long for the LDS kernels, a step over the dense-row limit) is not a hard
    failure: in the first step nothing has been written and the host front end takes over; in a LATER step the rows written so far are
    discarded and the whole input runs through the host front end — same _reads.txt / _counts.txt as COLORID_DEVICE_FASTQ=0.  The
    refusal is injected (CID_FASTQ_REFUSE_AT_STEP) at steps 0, 1 and 3 of a run of several stretches; R2 is shorter than R1."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "phage")
    p = subprocess.run([BIN, "build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(9)
    r1 = synth_fastq_records(np.random.default_rng(22), genomes, 16000, 150, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(22), genomes, 16000, 150, mate=1, lower_rate=0.0)
    f1, f2 = str(tmp_path / "r_1.fastq.gz"), str(tmp_path / "r_2.fastq.gz")
    _write_bgzf(f1, fastq_text(r1), rng)
    _write_bgzf(f2, fastq_text(r2[:9000]), rng)          # a truncated R2: R1's surplus has no mates
    q = [f1, f2] if paired else [f1]
    outs = {}
    for tag, env in (("host", {"COLORID_DEVICE_FASTQ": "0"}), ("dev", {"COLORID_DEVICE_FASTQ_MB": "1"}),
                     ("fail0", {"COLORID_DEVICE_FASTQ_MB": "1", "CID_FASTQ_REFUSE_AT_STEP": "0"}),
                     ("fail1", {"COLORID_DEVICE_FASTQ_MB": "1", "CID_FASTQ_REFUSE_AT_STEP": "1"}),
                     ("fail3", {"COLORID_DEVICE_FASTQ_MB": "1", "CID_FASTQ_REFUSE_AT_STEP": "3", "COLORID_DEVICE_FASTQ_AHEAD": "2"})):
        name = str(tmp_path / tag)
        p = subprocess.run([BIN, "read_id", "-b", pre + ".bxi", "-q", *q, "-n", name], capture_output=True, text=True, env=dict(os.environ, COLORID_TIMING="1", **env))
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        outs[tag] = (open(name + "_reads.txt").read(), open(name + "_counts.txt").read(), p.stderr)
    host = outs["host"]
    assert host[0].count("\n") == (9000 if paired else 16000)
    for tag in ("dev", "fail0", "fail1", "fail3"):
        assert outs[tag][0] == host[0] and outs[tag][1] == host[1], tag
    assert "using the host front end" in outs["fail0"][2]
    assert "starting over with the host front end" in outs["fail1"][2] and "starting over with the host front end" in outs["fail3"][2]


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("q", [0, 15])
def test_count_kmers_equals_the_oracles_fastq_maps(orc, hip_ctx, world, paired, q):
    """cid_fastq_count_kmers — `search`'s fastq k-mer maps (kmer.rs:461-510 single-end, :581-655 pairs) from text on the device: records
    cut anywhere by pushes and by block-gzip members, \\r\\n in the second file, which is shorter (the walk ends with it), quality
    masking, reads with N, an empty read; the set equals the oracle's map filled read by read (kmerize_fq_read).  A lower-case base is
    refused (its case would have to be kept) and so is a read longer than a segment."""
    import colorid_amd
    _, _, genomes = world
    k = 21
    rng = np.random.default_rng(40 + q + int(paired))
    r1 = synth_fastq_records(np.random.default_rng(9 + q), genomes, 1500, 150, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(9 + q), genomes, 1500, 150, mate=1, lower_rate=0.0)
    r1.insert(3, (b"empty", b"", b""))
    r2.insert(3, (b"empty", b"", b""))
    texts = [fastq_text(r1)] + ([fastq_text(r2[:1400], b"\r\n", False)] if paired else [])
    recs = [line_loop_records(t) for t in texts]
    n = min(len(r) for r in recs)
    want = orc.Kmers(k)
    for i in range(n):
        for r in recs:
            want.kmerize_fq_read(r[i][1], r[i][2], q)
    want = want.as_dict()
    assert len(want) > 20_000
    for how in ("text", "bgzf"):
        fr = colorid_amd.FastqReader(hip_ctx, len(texts), q)
        ks = colorid_amd.KmerSet(hip_ctx, k)
        taken = 0
        if how == "text":
            cuts = [sorted(rng.integers(0, len(t), 6).tolist()) + [len(t)] for t in texts]
            pos = [0] * len(texts)
            for step in range(7):
                for f, t in enumerate(texts):
                    fr.push_text(f, t[pos[f]:cuts[f][step]], last=(step == 6)); pos[f] = cuts[f][step]
                if step % 2 == 1 or step == 6:
                    taken += fr.count_kmers(ks)
        else:
            for f, t in enumerate(texts):
                members, lens, p = [], [], 0
                while p < len(t):
                    m = int(rng.choice([300, 5000, 30000, 65536]))
                    members.append(bgzf_member(t[p:p + m], level=int(rng.integers(1, 10)))); lens.append(len(t[p:p + m])); p += m
                half = len(members) // 2
                fr.push_bgzf(f, members[:half], lens[:half])
                fr.push_bgzf(f, members[half:], lens[half:], last=True)
            taken += fr.count_kmers(ks, max_pushes=1)
            taken += fr.count_kmers(ks)
        assert taken == n
        ks.finalize()
        got = ks.as_dict()
        assert got == want, (how, len(got), len(want))
        fr.close()
    fr = colorid_amd.FastqReader(hip_ctx, 1, q)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    fr.push_text(0, b"@r\n" + genomes[0][:100] + b"acgt" + genomes[0][104:150] + b"\n+\n" + b"I" * 150 + b"\n", last=True)
    with pytest.raises(colorid_amd.CidError) as ei:
        fr.count_kmers(ks)
    assert ei.value.code == -4 and "lower-case" in str(ei.value)
    fr.close()
    fr = colorid_amd.FastqReader(hip_ctx, 1, 0)
    ks = colorid_amd.KmerSet(hip_ctx, k)
    long_read = (genomes[1] * 3)[:9000]
    fr.push_text(0, b"@long\n" + long_read + b"\n+\n" + b"I" * 9000 + b"\n", last=True)
    with pytest.raises(colorid_amd.CidError) as ei:
        fr.count_kmers(ks)
    assert ei.value.code == -4
    fr.close()


@pytest.mark.parametrize("paired", [False, True])
def test_cli_search_counts_block_gzip_queries_on_the_device(orc, tmp_path, paired):
    """`colorid search` with a block-gzip fastq query: the k-mer map comes from the device front end (cid_fastq_count_kmers) and the
    report equals the host front end's (COLORID_DEVICE_FASTQ=0), default report and -g, with and without the host inflating a share."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "phage")
    p = subprocess.run([BIN, "build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(18)
    r1 = synth_fastq_records(np.random.default_rng(31), genomes[:2], 9000, 150, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(31), genomes[:2], 9000, 150, mate=1, lower_rate=0.0)
    f1, f2 = str(tmp_path / "q_1.fastq.gz"), str(tmp_path / "q_2.fastq.gz")
    _write_bgzf(f1, fastq_text(r1), rng)
    _write_bgzf(f2, fastq_text(r2[:8800], b"\r\n", False), rng)
    q = [f1, f2] if paired else [f1]
    outs = {}
    for tag, env in (("host", {"COLORID_DEVICE_FASTQ": "0"}), ("dev", {}), ("dev_small", {"COLORID_DEVICE_FASTQ_MB": "1"}),
                     ("dev_gpu_inflate", {"COLORID_DEVICE_FASTQ_HOST_SHARE": "0"}), ("dev_pinned", {"COLORID_DEVICE_FASTQ_PINNED": "1", "COLORID_DEVICE_FASTQ_MB": "1"})):
        for extra in (["-f", "1", "-p", "0.01"], ["-g", "-f", "0", "-p", "0.3"], ["-Q", "0", "-f", "2", "-p", "0.01"]):
            p = subprocess.run([BIN, "search", "-b", pre + ".bxi", "-q", q[0], *(["-r", q[1]] if paired else []), *extra], capture_output=True, text=True,
                               env=dict(os.environ, COLORID_TIMING="1", **env))
            assert p.returncode == 0, p.stderr[-2000:]
            assert ("through the device front end" in p.stderr) == (tag != "host"), (tag, p.stderr[-1500:])
            outs[(tag, tuple(extra))] = sorted(p.stdout.splitlines())
    for key, rows in outs.items():
        assert rows == outs[("host", key[1])], key
    assert any("Listeria_phage_B021" in r for r in outs[("dev", ("-f", "1", "-p", "0.01"))])


def test_cli_build_counts_block_gzip_accessions_on_the_device(orc, tmp_path):
    """`colorid build` with block-gzip fastq accessions (paired, single, next to a FASTA one; auto cutoff and -f 2): the accessions'
    k-mer maps come from the device front end and the .bxi is byte-identical to the oracle's and to the host front end's."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    genome = b"".join(orc.read_fasta(os.path.join(REFS, PHAGES[0] + ".fasta")))[:4000]
    rng = np.random.default_rng(77)
    r1 = synth_fastq_records(np.random.default_rng(3), [genome], 2500, 120, mate=0, lower_rate=0.0)
    r2 = synth_fastq_records(np.random.default_rng(3), [genome], 2500, 120, mate=1, lower_rate=0.0)
    f1, f2, f3 = str(tmp_path / "a_1.fastq.gz"), str(tmp_path / "a_2.fastq.gz"), str(tmp_path / "b.fastq.gz")
    _write_bgzf(f1, fastq_text(r1), rng); _write_bgzf(f2, fastq_text(r2), rng); _write_bgzf(f3, fastq_text(r1[:1500]), rng)
    tsv = tmp_path / "refs.tsv"
    tsv.write_text(f"pe_sample\t{f1}\t{f2}\nse_sample\t{f3}\nphage\t{os.path.join(REFS, PHAGES[1] + '.fasta')}\n")
    for flt in ([], ["-f", "2"]):
        oix = orc.Index.build_single(str(tsv), 200003, 3, 21, 15, int(flt[1]) if flt else -1)
        ref = str(tmp_path / ("oracle" + "".join(flt) + ".bxi"))
        oix.save(ref)
        for tag, env in (("dev", {}), ("host", {"COLORID_DEVICE_FASTQ": "0"})):
            pre = str(tmp_path / (tag + "".join(flt)))
            p = subprocess.run([BIN, "build", "-s", "200003", "-n", "3", "-k", "21", "-b", pre, "-r", str(tsv), *flt], capture_output=True, text=True,
                               env=dict(os.environ, COLORID_TIMING="1", **env))
            assert p.returncode == 0, p.stderr[-2000:]
            assert ("through the device front end" in p.stderr) == (tag == "dev"), p.stderr[-1500:]
            assert open(pre + ".bxi", "rb").read() == open(ref, "rb").read(), (tag, flt)


@pytest.mark.parametrize("paired", [False, True])
def test_cli_read_id_minimizer_index_block_gzip(orc, tmp_path, paired):
    """A minimizer index (.mxi) and block-gzip reads, some of them lower-case (the byte-string kernel takes those): the device front end
    and the host front end write the same rows."""
    import os
    import subprocess

    from test_gpu_cli import BIN, PHAGES, REFS
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "mini")
    p = subprocess.run([BIN, "build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv), "-m", "-v", "15"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(91)
    r1 = synth_fastq_records(np.random.default_rng(61), genomes, 4000, 150, mate=0, lower_rate=0.03)
    r2 = synth_fastq_records(np.random.default_rng(61), genomes, 4000, 150, mate=1, lower_rate=0.03)
    f1, f2 = str(tmp_path / "m_1.fastq.gz"), str(tmp_path / "m_2.fastq.gz")
    _write_bgzf(f1, fastq_text(r1), rng)
    _write_bgzf(f2, fastq_text(r2), rng)
    q = [f1, f2] if paired else [f1]
    outs = {}
    for tag, env in (("host", {"COLORID_DEVICE_FASTQ": "0"}), ("dev", {"COLORID_DEVICE_FASTQ_MB": "1"})):
        name = str(tmp_path / tag)
        p = subprocess.run([BIN, "read_id", "-b", pre + ".mxi", "-q", *q, "-n", name], capture_output=True, text=True, env=dict(os.environ, COLORID_TIMING="1", **env))
        assert p.returncode == 0, p.stderr[-2000:]
        assert ("device front end" in p.stderr) == (tag == "dev")
        outs[tag] = (open(name + "_reads.txt").read(), open(name + "_counts.txt").read())
    assert outs["dev"] == outs["host"]
    assert outs["host"][0].count("\n") == 4000 and outs["host"][0].count("Listeria_phage") > 2500
