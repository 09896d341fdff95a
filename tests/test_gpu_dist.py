"""Two RANKS of the real HIP path, one process each, sharing the one GPU (device 0), with torch.distributed for the exchange
step (gloo here: RCCL will not put two ranks on one GPU; bench.py --gpus N uses the same helpers with backend nccl = RCCL):
  * read-sharded / index-replicated: each rank searches its shard of the k-mers with cid_search_count_dev on
    its own replica, the 3*C counters are all-reduced: == the single-rank result == the oracle; perfect search likewise;
  * colour-striped (§8e.2): rank r holds stripe r only, every rank sees every k-mer, one SUM all-reduce of the packed per-k-mer
    facts + the tiny per-colour vector: == the whole-index oracle.
(tests/test_dist_gloo.py runs the same reductions on CPU with the oracle standing in for the kernels.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(orc):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import plant, random_index, random_kmers
    rng = np.random.default_rng(2024)
    C, n, k, m = 384, 3, 31, 50_021   # two stripes of 192 colours: stripes start on a multiple of 64
    oix = random_index(orc, rng, m, n, k, C, density=0.05, zero_row_frac=0.2)
    kmers = random_kmers(rng, 20_003, k)
    plant(oix, rng, kmers, frac=0.8, max_colours=2)
    for km in kmers[:500]:
        oix.insert(3, km.tobytes())
        oix.insert(C - 2, km.tobytes())
    freq = rng.integers(1, 40, size=len(kmers)).astype(np.uint32)
    return oix, kmers, freq, (C, n, k, m)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import colorid_amd
    from colorid_amd.dist import allgather_and, allreduce_counts, shard_bounds
    from colorid_amd.striped import StripedIndex
    from oracle import orc
    oix, kmers, freq, (C, n, k, m) = _case(orc)
    dev = torch.device("cuda", 0)                 # both ranks on the one GPU
    ctx = colorid_amd.Context(0)
    ok = True
    fails = []

    def chk(i, cond):
        if not cond:
            fails.append(i)
        return bool(cond)
    # ---- read-sharded, index replicated
    hx = colorid_amd.Index(ctx, m, n, k, C)
    hx.put_dense(oix.rows())
    hx.finalize()
    lo, hi = shard_bounds(len(kmers), rank, world)
    dk = torch.from_numpy(kmers[lo:hi].reshape(-1).copy()).to(dev).reshape(hi - lo, k)
    df = torch.from_numpy(freq[lo:hi].astype(np.int32)).to(dev)
    out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
    uc = torch.empty(hi - lo, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    hx.search_count_dev(dk.data_ptr(), df.data_ptr(), hi - lo, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr())
    ctx.synchronize()
    allreduce_counts(out)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    got = out.cpu().numpy().astype(np.uint64)
    ok = chk(1, all(np.array_equal(got[i * C:(i + 1) * C], want[i]) for i in range(3))) and ok
    ok = chk(2, np.array_equal(uc.cpu().numpy().view(np.uint32), want[3][lo:hi])) and ok
    single = hx.search_count(kmers, freq)          # the single-rank call on the same GPU
    ok = chk(3, all(np.array_equal(a, b) for a, b in zip(single, want))) and ok
    # perfect search over a sharded subset: AND of the ranks' words, OR of the flags
    sub = kmers[:500]
    slo, shi = shard_bounds(len(sub), rank, world)
    w_r, m_r = hx.search_perfect(sub[slo:shi])
    words, missing = allgather_and(torch.from_numpy(w_r.astype(np.int64)).to(dev), m_r)
    pw, pm = oix.search_perfect(sub)
    ok = chk(4, missing == pm and np.array_equal(words.cpu().numpy().astype(np.uint32), pw)) and ok
    hx.close()
    # ---- colour stripes: rank r holds colours [r*192, (r+1)*192)
    per = C // world
    base = rank * per
    w32s = (per + 31) // 32
    rows = oix.rows()[:, base // 32:base // 32 + w32s].copy()
    hs = colorid_amd.Index(ctx, m, n, k, per)
    hs.put_dense(rows)
    hs.finalize()
    si = StripedIndex(ctx, [(hs, base)], C)
    dka = torch.from_numpy(kmers.reshape(-1).copy()).to(dev).reshape(len(kmers), k)
    dfa = torch.from_numpy(freq.astype(np.int32)).to(dev)
    h, nu, sf, ucs = si.search_count(dka, dfa)
    ok = chk(5, np.array_equal(h.cpu().numpy().astype(np.uint64), want[0]) and np.array_equal(nu.cpu().numpy().astype(np.uint64), want[1])) and ok
    ok = chk(6, np.array_equal(sf.cpu().numpy().astype(np.uint64), want[2]) and np.array_equal(ucs.cpu().numpy().view(np.uint32), want[3])) and ok
    ok = chk(7, int(want[1].sum()) > 1000) and ok
    ds = torch.from_numpy(sub.reshape(-1).copy()).to(dev).reshape(len(sub), k)
    aw, miss = si.search_perfect(ds, (C + 63) // 64, lambda b: b // 64)
    ok = chk(8, miss == pm and np.array_equal(aw.cpu().numpy().view(np.uint32)[:oix.w32], pw)) and ok
    aw2, miss2 = si.search_perfect(dka[:3000].contiguous(), (C + 63) // 64, lambda b: b // 64)
    pw2, pm2 = oix.search_perfect(kmers[:3000])
    ok = chk(9, miss2 == pm2 and np.array_equal(aw2.cpu().numpy().view(np.uint32)[:oix.w32], pw2)) and ok
    hs.close()
    ctx.close()
    if not ok:
        print(f"rank {rank}: checks failed: {fails}", file=sys.stderr)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write("ok" if int(flag.item()) else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_hip_ranks_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result.txt").read() == "ok"
