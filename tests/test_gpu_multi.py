"""SURVEY.md App. F `tests/multi_gpu`: the same counts on 1 / 2 / 4 / 8 GPUs, for the replicated and the colour-striped placements.
Skipped where fewer devices exist (the build and the driver's test boxes have one GPU; the one-GPU stand-ins are tests/test_gpu_group.py —
several ranks sharing device 0 — and tests/test_gpu_dist.py — two processes on device 0 with gloo).  On a multi-GPU node these run the
real thing: cid_group with one rank per device and the RCCL all-reduce, and one process per GPU with torch.distributed backend nccl."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

from util import plant, random_index, random_kmers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_devices():
    import colorid_amd
    n = C.c_int(0)
    colorid_amd.load_library().cid_device_count(C.byref(n))
    return n.value


@pytest.mark.parametrize("n_gpus", [2, 4, 8])
def test_group_on_distinct_gpus(orc, n_gpus):
    if n_devices() < n_gpus:
        pytest.skip(f"needs {n_gpus} GPUs")
    import colorid_amd
    rng = np.random.default_rng(n_gpus)
    oix = random_index(orc, rng, 60_013, 4, 31, 256, density=0.2, zero_row_frac=0.05)
    kmers = random_kmers(rng, 100_003, 31)
    plant(oix, rng, kmers[:20_000], frac=0.9)
    freq = rng.integers(1, 30, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    g = colorid_amd.Group(list(range(n_gpus)))
    assert g.uses_rccl                                           # one rank per device: ncclAllReduce over xGMI
    hx = colorid_amd.Index(g.ctxs[0], oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.put_dense(oix.rows())
    hx.finalize()
    g.replicate(hx)                                              # device-to-device copies
    got = g.search_count(kmers, freq)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    ks = colorid_amd.KmerSet(g.ctxs[0], 31)
    ks.add_seqs([bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)) for _ in range(8)], 0)
    ks.finalize()
    assert all(np.array_equal(a, b) for a, b in zip(ks.search_count(hx), g.search_count_set(ks)))   # slices travel GPU to GPU
    pw, pm = oix.search_perfect(kmers[:5000])
    gw, gm = g.search_perfect(kmers[:5000])
    assert pm == gm and np.array_equal(pw, gw)
    g.close()


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
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import colorid_amd
    from colorid_amd.dist import allreduce_counts, shard_bounds
    from colorid_amd.striped import StripedIndex
    from oracle import orc
    rng = np.random.default_rng(5)
    C_, n, k, m = 64 * world * 2, 3, 31, 50_021
    oix = random_index(orc, rng, m, n, k, C_, density=0.05, zero_row_frac=0.2)
    kmers = random_kmers(rng, 30_011, k)
    plant(oix, rng, kmers, frac=0.8, max_colours=2)
    freq = rng.integers(1, 40, size=len(kmers)).astype(np.uint32)
    want = oix.search_count(kmers, freq.astype(np.uint64))
    ctx = colorid_amd.Context(rank)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ok = True
    # replicated: my shard of the k-mers, RCCL all-reduce of the counters
    hx = colorid_amd.Index(ctx, m, n, k, C_)
    hx.put_dense(oix.rows())
    hx.finalize()
    lo, hi = shard_bounds(len(kmers), rank, world)
    dk = torch.from_numpy(kmers[lo:hi].reshape(-1).copy()).to(dev).reshape(hi - lo, k)
    df = torch.from_numpy(freq[lo:hi].astype(np.int32)).to(dev)
    out = torch.zeros(3 * C_, dtype=torch.int64, device=dev)
    uc = torch.empty(hi - lo, dtype=torch.int32, device=dev)
    hx.search_count_dev(dk.data_ptr(), df.data_ptr(), hi - lo, out.data_ptr(), out.data_ptr() + 8 * C_, out.data_ptr() + 16 * C_, uc.data_ptr())
    allreduce_counts(out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.uint64)
    ok &= all(np.array_equal(got[i * C_:(i + 1) * C_], want[i]) for i in range(3))
    hx.close()
    # striped: my 128 colours, every k-mer, one all-reduce of the packed facts
    per = C_ // world
    base = rank * per
    hs = colorid_amd.Index(ctx, m, n, k, per)
    hs.put_dense(oix.rows()[:, base // 32:base // 32 + per // 32].copy())
    hs.finalize()
    si = StripedIndex(ctx, [(hs, base)], C_)
    dka = torch.from_numpy(kmers.reshape(-1).copy()).to(dev).reshape(len(kmers), k)
    dfa = torch.from_numpy(freq.astype(np.int32)).to(dev)
    h, nu, sf, ucs = si.search_count(dka, dfa)
    ok &= np.array_equal(h.cpu().numpy().astype(np.uint64), want[0]) and np.array_equal(nu.cpu().numpy().astype(np.uint64), want[1])
    ok &= np.array_equal(sf.cpu().numpy().astype(np.uint64), want[2]) and np.array_equal(ucs.cpu().numpy().view(np.uint32), want[3])
    hs.close()
    ctx.close()
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write("ok" if int(flag.item()) else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_gpus", [2, 4, 8])
@pytest.mark.timeout(900)
def test_one_process_per_gpu_rccl(tmp_path, n_gpus):
    if n_devices() < n_gpus:
        pytest.skip(f"needs {n_gpus} GPUs")
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (the host driver only supports dmabuf IPC: RCCL between processes needs it)
    mp.spawn(_worker, args=(n_gpus, _free_port(), str(tmp_path)), nprocs=n_gpus, join=True)   # fresh interpreters (spawn): children, not an exec of this process
    assert open(tmp_path / "result.txt").read() == "ok"


@pytest.mark.parametrize("n_gpus", [2, 4, 8])
@pytest.mark.timeout(900)
def test_bench_self_launch_over_rccl_reproduces_the_digests(n_gpus):
    """`python bench.py --gpus N --scale-check` with the default backend (RCCL): the launcher starts its own N ranks as child processes
    before anything touches the GPU, and the all-reduced counters carry the digest one GPU standing in for N committed
    (tests/golden/scale_digests.json) — replicated and colour-striped"""
    if n_devices() < n_gpus:
        pytest.skip(f"needs {n_gpus} GPUs")
    import json
    import subprocess
    fast = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-variants", "--scale-check"]
    for extra in ([], ["--placement", "striped", "--stripe-log2-bloom", "27"]):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus)] + fast + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, env=env, cwd=ROOT, timeout=800)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == n_gpus and line["config"]["backend"] == "nccl" and line["scale_check"]["ok"] is True, line["scale_check"]
        assert len(line["per_rank"]) == n_gpus and line["config"]["setup_s"] < 60
