"""Hash variants.  The reference hashes with crate `xxh3 ^0.1.1` (Cargo.toml:9; src/simple_bloom.rs:19-38), which cannot run
here, and the .bxi format carries no hash id.  Two variants exist behind `hash_variant` / `--hash`: the published XXH3 (v0.8,
pinned to known answers) and the v0.7.1/v0.7.2 draft (a candidate, restated from memory).  These tests hold
  * HIP == oracle under the candidate variant on every path that hashes (search, perfect search, read_id, Bloom insert);
  * `colorid hashcheck` — the tool that decides, from an index and the sequences it was built from, which variant the index
    was built with — on indices written by this program AND by the oracle's writer, under either variant."""
import os
import subprocess

import numpy as np
import pytest

from test_gpu_cli import BANNER, BIN, PHAGES, REFS
from test_gpu_readid import pack_reads, sample_reads
from util import plant, random_index, random_kmers

pytestmark = pytest.mark.gpu


def _hip_index(ctx, oix, variant):
    import colorid_amd
    hx = colorid_amd.Index(ctx, oix.m, oix.n_hash, oix.k, oix.n_colors, hash_variant=variant)
    hx.put_dense(oix.rows())
    return hx.finalize()


@pytest.mark.parametrize("k", [3, 8, 12, 16, 17, 21, 27, 31, 32, 33, 64, 100, 128])
def test_search_parity_under_the_v07_candidate(orc, hip_ctx, k):
    rng = np.random.default_rng(k)
    with orc.hash_variant(1):
        oix = random_index(orc, rng, 20_011, 4, k, 200, density=0.25, zero_row_frac=0.1)
        kmers = random_kmers(rng, 3000, k)
        plant(oix, rng, kmers, frac=0.6)
        freq = rng.integers(1, 9, size=len(kmers)).astype(np.uint32)
        want = oix.search_count(kmers, freq.astype(np.uint64))
        pw, pm = oix.search_perfect(kmers[:50])
    hx = _hip_index(hip_ctx, oix, 1)
    got = hx.search_count(kmers, freq)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    gw, gm = hx.search_perfect(kmers[:50])
    assert gm == pm and np.array_equal(gw, pw)
    # the same rows read under the other variant give another answer (the planted k-mers are no longer found) ...
    hx.set_hash_variant(0)
    other = hx.search_count(kmers, freq)
    assert not np.array_equal(other[0], want[0])
    # ... and switching back restores it
    hx.set_hash_variant(1)
    assert np.array_equal(hx.search_count(kmers, freq)[0], want[0])
    hx.close()


def test_readid_and_insert_under_the_v07_candidate(orc, hip_ctx, tmp_path):
    import colorid_amd
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(2)
    reads = sample_reads(orc, rng, genomes, 300, 150, True)
    bases, seq_off, read_seq0 = pack_reads(reads)
    with orc.hash_variant(1):
        oix = orc.Index.build_single(str(tsv), 750_000, 4, 27)
        want = oix.readid_counts(bases, seq_off, read_seq0, 1, 3)
    with orc.hash_variant(0):
        oix0 = orc.Index.build_single(str(tsv), 750_000, 4, 27)
    assert not np.array_equal(oix.rows(), oix0.rows())
    # Bloom inserts on the device under variant 1 == the oracle's build under variant 1
    hx = colorid_amd.Index(hip_ctx, 750_000, 4, 27, 4, hash_variant=1)
    for c, g in enumerate(PHAGES):
        ks = colorid_amd.KmerSet(hip_ctx, 27)
        ks.add_seqs(orc.read_fasta(os.path.join(REFS, g + ".fasta")), 0)
        ks.finalize()
        from colorid_amd._lib import check
        check(hip_ctx.lib.cid_index_insert_kmerset(hx.h, ks.h, c))
        ks.close()
    hx.finalize()
    nz = np.flatnonzero(oix.rows().any(axis=1)).astype(np.uint64)
    assert np.array_equal(hx.get_rows(nz), oix.rows()[nz])
    assert np.array_equal(hx.get_rows(np.arange(0, 750_000, 97, dtype=np.uint64)), oix.rows()[::97])
    got = hx.readid_count(bases, seq_off, read_seq0, 1, 3)
    assert all(np.array_equal(w, g) for w, g in zip(want, got))
    hx.close()


def _cli(*args, ok=(0,)):
    p = subprocess.run([BIN, *args], capture_output=True, text=True)
    assert p.returncode in ok, (p.returncode, p.stderr[-2000:])
    assert p.stdout.startswith(BANNER)
    return p.stdout[len(BANNER):], p.stderr, p.returncode


def test_hashcheck_identifies_the_variant(orc, tmp_path):
    """`colorid hashcheck -b X.bxi -r ref_file`: under the right variant every accession's own k-mers are all present (a Bloom
    filter has no false negatives), under a wrong one only about (row density)^n of them."""
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    for name, variant in (("xxh3_v08", 0), ("xxh3_v07", 1)):
        other = "xxh3_v07" if variant == 0 else "xxh3_v08"
        # (a) an index written by the C++ CLI, (b) one written by the oracle's independent writer
        pre = str(tmp_path / f"cli_{name}")
        _cli("build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv), "--hash", name)
        with orc.hash_variant(variant):
            orc.Index.build_single(str(tsv), 750000, 4, 27).save(str(tmp_path / f"orc_{name}.bxi"))
        assert open(pre + ".bxi", "rb").read() == open(tmp_path / f"orc_{name}.bxi", "rb").read()
        out, _, rc = _cli("hashcheck", "-b", pre + ".bxi", "-r", str(tsv))
        rows = [l.split("\t") for l in out.splitlines()]
        assert rows[0] == ["variant", "accession", "kmers", "present", "fraction"]
        frac = {(r[0], r[1]): (int(r[2]), int(r[3]), float(r[4])) for r in rows[1:] if len(r) == 5 and r[0] != "verdict"}
        assert len(frac) == 2 * len(PHAGES)
        for acc in PHAGES:
            nk, present, f = frac[(name, acc)]
            assert present == nk and f == 1.0 and nk > 20_000
            assert frac[(other, acc)][2] < 0.05                              # ~ density^4 for a wrong variant
        verdict = [r for r in rows if r[0] == "verdict"][0]
        assert verdict[1] == name and rc == 0
        # searching with the wrong --hash finds nothing; with the right one, the genome itself
        q = os.path.join(REFS, PHAGES[2] + ".fasta")
        good, _, _ = _cli("search", "-b", pre + ".bxi", "-q", q, "-s", "--hash", name)
        bad, err, _ = _cli("search", "-b", pre + ".bxi", "-q", q, "-s", "--hash", other)
        assert PHAGES[2] in good and bad.strip() == ""
    # an index that no variant reproduces (rows of a different k): verdict none, exit code 3
    pre = str(tmp_path / "k25")
    _cli("build", "-s", "750000", "-n", "4", "-k", "25", "-b", pre, "-r", str(tsv))
    raw = bytearray(open(pre + ".bxi", "rb").read())
    raw[16:24] = (27).to_bytes(8, "little")                                  # claim k = 27 for rows hashed from 25-mers
    open(pre + "_lie.bxi", "wb").write(raw)
    out, _, rc = _cli("hashcheck", "-b", pre + "_lie.bxi", "-r", str(tsv), ok=(3,))
    assert "verdict\tnone" in out and rc == 3
