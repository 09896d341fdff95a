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
