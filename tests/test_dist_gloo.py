"""The N > 1 path on CPU: world_size-2 gloo processes shard the query k-mers with colorid_amd.dist.shard_bounds,
compute their partial counters (the oracle stands in for the per-rank GPU call — a test double, not a product
path) and combine them with the same allreduce_counts / allgather_and bench.py and the drivers use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from colorid_amd.dist import allgather_and, allreduce_counts, shard_bounds
    from oracle import orc
    from util import plant, random_index, random_kmers
    rng = np.random.default_rng(123)                      # same index and query on every rank
    oix = random_index(orc, rng, 20_011, 3, 31, 200, density=0.3, zero_row_frac=0.05)
    kmers = random_kmers(rng, 5001, 31)
    plant(oix, rng, kmers, frac=0.6)
    for km in kmers[:600]:
        oix.insert(7, km.tobytes())
    freq = rng.integers(1, 100, size=len(kmers)).astype(np.uint64)
    lo, hi = shard_bounds(len(kmers), rank, world)
    hits, nu, sf, _ = oix.search_count(kmers[lo:hi], freq[lo:hi])
    counts = torch.from_numpy(np.concatenate([hits, nu, sf]).astype(np.int64))
    allreduce_counts(counts)
    plo, phi = shard_bounds(600, rank, world)
    words, missing = oix.search_perfect(kmers[plo:phi])
    gw, gm = allgather_and(torch.from_numpy(words.astype(np.int64)), missing)
    if rank == 0:
        full = oix.search_count(kmers, freq)
        want = np.concatenate(full[:3]).astype(np.int64)
        fw, fm = oix.search_perfect(kmers[:600])
        ok = bool(np.array_equal(counts.numpy(), want)) and bool(np.array_equal(gw.numpy(), fw.astype(np.int64))) and gm == fm
        ok = ok and bool(fw[0] >> 7 & 1) and int(want[:200].sum()) > 0
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def _stripe_worker(rank, world, port, out_dir):
    """Colour-striped placement: rank r holds colours [r*128, (r+1)*128) of a 256-colour index; the per-stripe facts
    (numpy stand-ins for what cid_search_count_stripe_dev / cid_search_perfect_stripe_dev leave in HBM) are combined
    with colorid_amd.striped.reduce_* and must reproduce the whole-index oracle result."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from colorid_amd.striped import reduce_perfect_facts, reduce_stripe_facts
    from oracle import orc
    from util import plant, random_index, random_kmers
    rng = np.random.default_rng(77)
    C, n, k, m = 256, 3, 21, 9_973
    oix = random_index(orc, rng, m, n, k, C, density=0.02, zero_row_frac=0.3)
    kmers = random_kmers(rng, 1500, k)
    plant(oix, rng, kmers, frac=0.8, max_colours=2)
    rows64 = oix.rows().astype(np.uint64)
    lo, hi = rank * 128, (rank + 1) * 128
    mine = rows64[:, lo // 32:hi // 32]                                   # this rank's stripe: 4 u32 words per row
    K = len(kmers)
    fact = np.zeros(K, np.uint32); zero_acc = np.full(K, -1, np.int32)       # fact: n << 26 | colour + 1 (include/colorid_hip.h)
    hits_full = np.zeros(C, np.int64)
    and_words = np.full(4, 0xFFFFFFFF, np.uint64)
    for j, km in enumerate(kmers):
        ridx = [orc.xxh3(km.tobytes(), s) % m for s in range(n)]
        a = mine[ridx[0]].copy()
        zmask = 0
        for s, r in enumerate(ridx):
            a &= mine[r]
            if not mine[r].any():
                zmask |= 1 << s
        zero_acc[j] &= zmask
        and_words &= a
        bits = [c for c in range(128) if (int(a[c // 32]) >> (c % 32)) & 1]
        fact[j] = (min(len(bits), 2) << 26) | (lo + bits[0] + 1 if len(bits) == 1 else 0)
        for c in bits:
            hits_full[lo + c] += 1
    tf, th = torch.from_numpy(fact.view(np.int32)), torch.from_numpy(hits_full)
    reduce_stripe_facts(tf, th)                                           # ONE 4-byte-per-k-mer collective + the tiny hits vector
    and_full = np.full(4, -1, np.int64)                                   # 4 u64 words for 256 colours
    w64 = (and_words[0::2] | (and_words[1::2] << np.uint64(32))).view(np.int64)
    and_full[rank * 2:rank * 2 + 2] = w64
    tz, tw = reduce_perfect_facts(torch.from_numpy(zero_acc), torch.from_numpy(and_full))
    if rank == 0:
        full = oix.search_count(kmers, None)
        summed = tf.numpy().view(np.uint32)
        uniq = (summed >> 26) == 1
        col = (summed & ((1 << 26) - 1)).astype(np.int64) - 1
        nu = np.bincount(col[uniq], minlength=C)
        ok = np.array_equal(th.numpy().astype(np.uint64), full[0]) and np.array_equal(nu.astype(np.uint64), full[1])
        uc = np.where(uniq, col.astype(np.uint32), np.uint32(0xFFFFFFFF))
        ok = ok and np.array_equal(uc, full[3]) and int(full[1].sum()) > 50
        pw, pm = oix.search_perfect(kmers)
        missing = bool(((tz.numpy() & ((1 << n) - 1)) != 0).any())
        ok = ok and missing == pm
        if not pm:
            ok = ok and np.array_equal(tw.numpy().view(np.uint32)[:8], pw)
        with open(os.path.join(out_dir, "stripe.txt"), "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    sys.path.insert(0, ROOT)
    from colorid_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard_bounds(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_reduction_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result.txt").read() == "ok"


@pytest.mark.timeout(300)
def test_two_rank_colour_stripes_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_stripe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "stripe.txt").read() == "ok"
