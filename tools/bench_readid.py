#!/usr/bin/env python3
"""Secondary measurement (not the driver's bench line): cid_readid_count_dev on BASELINE.json configs[2]'s shape —
m = 30,000,000, n = 2, k = 21, 256 colours, synthetic 150-bp reads resident in HBM; reports reads/s and the
line-request rate the gather implies.  Usage: python tools/bench_readid.py [--reads N] [--paired] [-d D] [-B S]"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import colorid_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--bloom", type=int, default=30_000_000)
    ap.add_argument("--hashes", type=int, default=2)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--colours", type=int, default=256)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("-d", type=int, default=1)
    ap.add_argument("-B", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--check", type=int, default=2000, help="reads compared with the oracle")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = colorid_amd.Context(0)
    ctx.set_stream(stream.cuda_stream)
    C, n, k, m = a.colours, a.hashes, a.k, a.bloom
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    p_bg = 1.0 - math.exp(-n * a.genome_len / m)
    bench.fill_background(dev, ptr, m, rs, C, p_bg, seed=7)
    mates = 2 if a.paired else 1
    n_seq = a.reads * mates
    kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, n_seq, a.read_len, k, C, 0.01, return_reads=True)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
    ctx.synchronize()
    hx.finalize()
    del kk, ff, cc
    bases = reads.reshape(-1).contiguous()
    seq_off = (torch.arange(n_seq + 1, device=dev, dtype=torch.int64) * a.read_len).contiguous()
    read0 = (torch.arange(a.reads + 1, device=dev, dtype=torch.int64) * mates).contiguous()
    report = torch.empty((a.reads, C + 1), dtype=torch.int32, device=dev)
    nk = torch.empty(a.reads, dtype=torch.int32, device=dev)
    st = torch.empty(a.reads, dtype=torch.uint8, device=dev)
    max_bytes = a.read_len * mates
    max_win = ((a.read_len - k) // a.d + 1) * mates

    def step():
        hx.readid_count_dev(bases.data_ptr(), seq_off.data_ptr(), read0.data_ptr(), a.reads, a.d, a.B, max_bytes, max_win,
                            report.data_ptr(), nk.data_ptr(), st.data_ptr())
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(a.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    nk_sum = int(nk.to(torch.int64).sum().item())
    out = {"kernel": "k_readid", "reads": a.reads, "paired": a.paired, "d": a.d, "B": a.B, "ms": ms,
           "reads_per_s": a.reads / ms * 1e3, "distinct_kmers_per_s": nk_sum / ms * 1e3,
           "row_gathers_per_s": nk_sum * n / ms * 1e3, "config": {"m": m, "n": n, "k": k, "C": C, "density": p_bg},
           "alg_GBs": (nk_sum * n * rs * 8 + bases.numel() + a.reads * (C + 1) * 4) / ms / 1e6}
    if a.check:
        from oracle import orc
        import numpy as np
        S = min(a.check, a.reads)
        rows = np.empty((m, rs * 2), np.uint32)
        bench.hip_memcpy(rows.ctypes.data, ptr, rows.nbytes, 2)
        oix = orc.Index(m, n, k, C)
        oix.rows()[:] = rows[:, :oix.w32]
        hb = bases[:S * max_bytes].cpu().numpy()
        want = oix.readid_counts(hb, seq_off[:S * mates + 1].cpu().numpy().astype(np.uint64), read0[:S + 1].cpu().numpy().astype(np.uint64), a.d, a.B)
        out["bit_exact_vs_oracle"] = bool(np.array_equal(want[0], report[:S].cpu().numpy().view(np.uint32)) and
                                          np.array_equal(want[1], nk[:S].cpu().numpy().view(np.uint32)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
