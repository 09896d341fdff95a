"""Several ranks of the HIP path (cid_group): the query is sharded over N contexts, each with its own replica of
the index, the per-accession counters are reduced, and the result must equal the single-rank result and the oracle — for the
proportional search (host k-mers and device-resident k-mer sets), the perfect search and read_id.  One GPU is available, so
the ranks share device 0 (`[0, 0]`, `[0, 0, 0]`: RCCL refuses two ranks on one GPU, the counters are then summed through the
host); the RCCL plumbing itself (dlopen, ncclCommInitAll, grouped ncclAllReduce on the ranks' streams) runs with one rank under
COLORID_REDUCE=rccl.  The same through the CLI: --devices 0,0 vs --device 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_cli import BANNER, BIN, PHAGES, REFS
from test_gpu_readid import pack_reads, sample_reads
from util import plant, random_index, random_kmers, synth_fastq_records, write_fastq_gz

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group_index(g, oix, variant=0):
    import colorid_amd
    hx = colorid_amd.Index(g.ctxs[0], oix.m, oix.n_hash, oix.k, oix.n_colors, hash_variant=variant)
    hx.put_dense(oix.rows())
    hx.finalize()
    g.replicate(hx)
    return hx


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0]])
@pytest.mark.parametrize("n_colors,k", [(256, 31), (46, 27), (1024, 21), (9000, 31), (200, 40)])
def test_group_search_equals_single_rank_and_oracle(orc, hip_ctx, devices, n_colors, k, monkeypatch, tune):
    import colorid_amd
    monkeypatch.setenv("CID_KMERSET_TARGET_SMALL", "1")   # (a set for an index below 2^20 rows keeps code order by default; the group's contexts are made below)
    tune("CID_KMERSET_TARGET_SMALL", 1)                   # ... and the session's context
    rng = np.random.default_rng(n_colors + len(devices))
    m = 40_009 if n_colors < 2000 else 4001
    oix = random_index(orc, rng, m, 4, k, n_colors, density=0.15 if n_colors < 2000 else 0.01, zero_row_frac=0.05)
    kmers = random_kmers(rng, 10_007, k)                       # not a multiple of anything
    plant(oix, rng, kmers, frac=0.7)
    freq = rng.integers(1, 30, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    g = colorid_amd.Group(devices)
    assert not g.uses_rccl
    _group_index(g, oix)
    got = g.search_count(kmers, freq)
    for w, x in zip(want, got):
        assert np.array_equal(w, x)
    # fewer k-mers than ranks, and none
    for nk in (1, 2, 0):
        w = oix.search_count(kmers[:nk], freq[:nk].astype(np.uint64))
        x = g.search_count(kmers[:nk], freq[:nk])
        assert all(np.array_equal(a, b) for a, b in zip(w, x))
    # perfect search: a subset planted in two colours, then one with an absent row somewhere in some shard
    sub = kmers[:600].copy()
    for km in sub:
        oix.insert(1, km.tobytes())
        oix.insert(n_colors - 1, km.tobytes())
    g.close()
    g = colorid_amd.Group(devices)
    hx = _group_index(g, oix)
    for sel in (sub, kmers[:3000], kmers[:1]):
        pw, pm = oix.search_perfect(sel)
        gw, gm = g.search_perfect(sel)
        assert gm == pm and np.array_equal(gw, pw)
    # device-resident k-mer sets (2-bit codes, or byte strings for k > 32): counted on rank 0, sliced to the ranks device-to-device
    seqs = [bytes(rng.choice(list(b"ACGT"), size=3000).astype(np.uint8)) for _ in range(5)] + [sub[:200].tobytes()]
    for targeted in (False, True):      # in code order, and built FOR rank 0's replica (cid_kmerset_set_target_index): the ranks take slices of either
        ks = colorid_amd.KmerSet(g.ctxs[0], k)
        if targeted:
            ks.set_target_index(hx)
        ks.add_seqs(seqs, 0)
        ks.finalize()
        single = ks.search_count(hx)
        multi = g.search_count_set(ks)
        assert all(np.array_equal(a, b) for a, b in zip(single, multi))
        sw, sm = ks.search_perfect(hx)
        mw, mm = g.search_perfect_set(ks)
        assert sm == mm and np.array_equal(sw, mw)
        km, cnt = ks.download()
        w = oix.search_count(km, cnt.astype(np.uint64))
        assert all(np.array_equal(a, b) for a, b in zip(w, multi))
        ks.close()
    g.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_group_readid_rows_in_input_order(orc, devices, tmp_path):
    import colorid_amd
    tsv = tmp_path / "refs.tsv"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    oix = orc.Index.build_single(str(tsv), 750_000, 4, 27)
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(len(devices))
    g = colorid_amd.Group(devices)
    _group_index(g, oix)
    for paired, n_reads in ((True, 501), (False, 3), (False, 0)):
        reads = sample_reads(orc, rng, genomes, n_reads, 150, paired) if n_reads else []
        bases, seq_off, read_seq0 = pack_reads(reads)
        for d, S in ((1, 3), (10, 0)):
            want = oix.readid_counts(bases, seq_off, read_seq0, d, S)
            rs, col, cnt, nk, st = g.readid_count_sparse(bases, seq_off, read_seq0, d, S)
            assert np.array_equal(nk, want[1]) and np.array_equal(st, want[2])
            rows, cols = np.nonzero(want[0])
            assert np.array_equal(rs, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=len(want[0])))]).astype(np.uint64))
            assert np.array_equal(col, cols.astype(np.uint32)) and np.array_equal(cnt, want[0][rows, cols])
    g.close()


def _cli(*args, env=None):
    p = subprocess.run([BIN, *args], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-3000:]
    assert p.stdout.startswith(BANNER)
    return p.stdout[len(BANNER):], p.stderr


def test_cli_gpus_equals_single_gpu(orc, tmp_path):
    """`colorid search|read_id --devices 0,0[,0]` (two / three ranks sharing the one GPU) == `--device 0`; `--gpus 1` and the RCCL
    reduction with one rank (COLORID_REDUCE=rccl --devices 0 goes through ncclCommInitAll + ncclAllReduce) == the same."""
    tsv = tmp_path / "ref_file.txt"
    tsv.write_text("".join(f"{n}\t{os.path.join(REFS, n + '.fasta')}\n" for n in PHAGES))
    pre = str(tmp_path / "phage")
    _cli("build", "-s", "750000", "-n", "4", "-k", "27", "-b", pre, "-r", str(tsv))
    genomes = [b"".join(orc.read_fasta(os.path.join(REFS, n + ".fasta"))) for n in PHAGES]
    rng = np.random.default_rng(31)
    r1 = synth_fastq_records(rng, genomes, 4000, 120, mate=0)
    rng = np.random.default_rng(31)
    r2 = synth_fastq_records(rng, genomes, 4000, 120, mate=1)
    f1, f2 = str(tmp_path / "r_1.fastq.gz"), str(tmp_path / "r_2.fastq.gz")
    write_fastq_gz(f1, r1)
    write_fastq_gz(f2, r2)
    fasta = os.path.join(REFS, PHAGES[2] + ".fasta")
    cases = {
        "search_pe": ("search", "-b", pre + ".bxi", "-q", f1, "-r", f2, "-f", "1", "-p", "0.01"),
        "search_gene": ("search", "-b", pre + ".bxi", "-q", f1, "-g", "-f", "0", "-p", "0.01"),
        "search_fasta": ("search", "-b", pre + ".bxi", "-q", fasta, "-p", "0.01"),
        "perfect": ("search", "-b", pre + ".bxi", "-q", fasta, "-s"),
        "perfect_mf": ("search", "-b", pre + ".bxi", "-q", fasta, "-s", "-m"),
    }
    for name, args in cases.items():
        base, _ = _cli(*args, "--device", "0")
        assert base.strip(), name
        for extra, env in ((("--devices", "0,0"), None), (("--devices", "0,0,0"), None), (("--gpus", "1"), None),
                           (("--devices", "0"), {"COLORID_REDUCE": "rccl"})):
            out, err = _cli(*args, *extra, env=env)
            assert sorted(out.splitlines()) == sorted(base.splitlines()), (name, extra)
        # lower-case / k > 32 inputs take the host k-mer map: the host-pointer group calls
        out, _ = _cli(*args, "--devices", "0,0", env={"COLORID_HOST_KMERS": "1"})
        assert sorted(out.splitlines()) == sorted(base.splitlines()), (name, "host map")
        # the query counted on rank 0 only and sliced to the ranks (the default counts it over all ranks: cid_group_kmerset)
        out, _ = _cli(*args, "--devices", "0,0,0", env={"COLORID_ONE_GPU_KMERS": "1"})
        assert sorted(out.splitlines()) == sorted(base.splitlines()), (name, "rank-0 counting")
    # read_id: per-read rows in input order, identical files
    for tag, q in (("se", (f1,)), ("pe", (f1, f2))):
        _cli("read_id", "-b", pre + ".bxi", "-q", *q, "-n", str(tmp_path / f"one_{tag}"), "-c", "700", "-t", "8")
        _cli("read_id", "-b", pre + ".bxi", "-q", *q, "-n", str(tmp_path / f"two_{tag}"), "-c", "700", "--devices", "0,0,0")
        assert open(tmp_path / f"one_{tag}_reads.txt").read() == open(tmp_path / f"two_{tag}_reads.txt").read()
        assert open(tmp_path / f"one_{tag}_counts.txt").read() == open(tmp_path / f"two_{tag}_counts.txt").read()
    # the RCCL path is what --gpus N takes on N distinct GPUs; with one rank it must say so
    _, err = _cli(*cases["search_fasta"], "--devices", "0", env={"COLORID_REDUCE": "rccl"})
    _, err2 = _cli(*cases["search_fasta"], "--devices", "0,0")
    assert "RCCL all-reduce" in err and "through the host" in err2


def test_round2_entry_points_refuse_misuse(orc, hip_ctx):
    """Every misuse of the round-2 entry points comes back as an error code + message (nothing aborts across the ABI)."""
    import ctypes as C

    import torch

    import colorid_amd
    from colorid_amd._lib import vp
    lib = hip_ctx.lib
    gh = vp()
    assert lib.cid_group_create(None, 2, C.byref(gh)) == -1
    assert lib.cid_group_create((C.c_int * 1)(0), 0, C.byref(gh)) == -1
    assert lib.cid_group_create((C.c_int * 1)(99), 1, C.byref(gh)) < 0 and not gh.value          # no such device
    assert lib.cid_ctx_tune(hip_ctx.h, b"no_such_knob", 1) == -1 and b"unknown tunable" in lib.cid_last_error()
    assert lib.cid_ctx_tune(hip_ctx.h, b"order_bits", 40) == -1 and lib.cid_ctx_tune(hip_ctx.h, b"search_unroll", 3) == -1
    assert lib.cid_ctx_tune(hip_ctx.h, b"search_persist", 1) == -4 and lib.cid_ctx_tune(None, b"order_bits", 0) == -1
    rng = np.random.default_rng(3)
    oix = random_index(orc, rng, 5003, 2, 21, 70, density=0.2, zero_row_frac=0.1)
    g = colorid_amd.Group([0, 0])
    hx = colorid_amd.Index(g.ctxs[0], oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.put_dense(oix.rows())
    arr = (vp * 2)()
    assert lib.cid_group_replicate_index(g.h, hx.h, arr) == -5                                      # not finalized
    hx.finalize()
    g.replicate(hx)
    other = colorid_amd.Index(g.ctxs[1], 4001, 2, 21, 70).finalize()                                # a replica of another shape
    bad = (vp * 2)(hx.h.value, other.h.value)
    hits = np.zeros(70, np.uint64)
    km = random_kmers(rng, 10, 21)
    assert lib.cid_group_search_count(g.h, bad, km.ctypes.data, None, 10, hits.ctypes.data, None, None, None) == -1
    assert b"differs" in lib.cid_last_error()
    wrong_dev = (vp * 2)(hx.h.value, hx.h.value)                                                     # rank 1's replica must live in rank 1's ctx?  same device: accepted
    assert lib.cid_group_search_count(g.h, wrong_dev, km.ctypes.data, None, 10, hits.ctypes.data, None, None, None) == 0
    words = np.zeros(3, np.uint32)
    miss = C.c_int(0)
    assert lib.cid_group_search_perfect(g.h, g._replica_handles, km.ctypes.data, 0, words.ctypes.data, C.byref(miss)) == -1    # no k-mers
    ks = colorid_amd.KmerSet(g.ctxs[0], 40)                                                          # a byte-string set against a k = 21 index
    ks.add_seqs([bytes(rng.choice(list(b"ACGT"), size=200).astype(np.uint8))], 0)
    ks.finalize()
    assert lib.cid_group_search_count_set(g.h, g._replica_handles, ks.h, hits.ctypes.data, None, None, None) == -1 and b"k=" in lib.cid_last_error()
    with pytest.raises(colorid_amd.CidError):
        ks.order_for_index(hx)
    ks.close()
    # stripes: read_id passes refuse wide stripes and nonsense maxima; the packed fact refuses more than 2^20 colours
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(8, dtype=torch.int64, device="cuda")
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    c0 = g.ctxs[0]
    assert lib.cid_readid_stripe_zero_dev(c0.h, hx.h, vp(d.data_ptr()), vp(o.data_ptr()), vp(o.data_ptr()), 1, 1, 64, 0, vp(z.data_ptr()),
                                          vp(z.data_ptr()), vp(d.data_ptr())) == -1                 # max_read_windows = 0
    assert lib.cid_readid_stripe_count_dev(c0.h, hx.h, vp(d.data_ptr()), vp(o.data_ptr()), vp(o.data_ptr()), 1, 1, 3, 64, 44, 60, 100, 1,
                                           vp(z.data_ptr()), vp(z.data_ptr()), vp(z.data_ptr()), vp(d.data_ptr())) == -1   # stripe outside the colour range
    assert lib.cid_search_unique_finalize_dev(c0.h, vp(z.data_ptr()), None, 64, (1 << 20) + 1, vp(o.data_ptr()), vp(o.data_ptr()), vp(z.data_ptr())) == -1
    assert lib.cid_index_set_hash_variant(hx.h, 7) == -4
    # everything still works afterwards
    got = g.search_count(km)
    assert np.array_equal(got[0], oix.search_count(km, None)[0])
    other.close()
    g.close()
