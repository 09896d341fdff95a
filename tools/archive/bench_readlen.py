#!/usr/bin/env python3
"""read_id throughput against read length (host-pointer sparse call, i.e. H2D of the reads and D2H of the sparse report
included): which of the two per-read set builders — the per-wave LDS table (k_readid) or the sort-based lists
(readid_long + k_readid_list) — takes which reads.  The library routes per read (cid_api_readid.hip: kLdsReadBytesMax); the
table in profiles/r01_readlen.md was measured with a temporary switch that forced the sort path.
Usage: python tools/bench_readlen.py [--bases 150000000]"""
import argparse, json, math, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np
    import torch
    import bench
    import colorid_amd
    dev = torch.device("cuda", 0)
    ctx = colorid_amd.Context(0)
    C, n, k, m = 256, 2, 21, 30_000_000
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
    hx.finalize()
    rng = np.random.default_rng(1)
    L = a.read_len
    R = max(1, a.bases // L)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, R * L)]
    seq_off = (np.arange(R + 1, dtype=np.uint64) * L)
    r0 = np.arange(R + 1, dtype=np.uint64)
    hx.readid_count_sparse(bases, seq_off, r0, 1, 3)
    t = time.time()
    for _ in range(a.steps):
        out = hx.readid_count_sparse(bases, seq_off, r0, 1, 3)
    ms = (time.time() - t) / a.steps * 1e3
    print(json.dumps({"read_len": L, "reads": R, "ms": round(ms, 2),
                      "Mbases_per_s": round(R * L / ms / 1e3, 1)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=150_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--read-len", type=int, default=0)
    a = ap.parse_args()
    if a.read_len:
        child(a)
    else:
        for L in (150, 300, 600, 1000, 2000, 4000, 8000, 20000, 100000):
            subprocess.run([sys.executable, __file__, "--bases", str(a.bases), "--steps", str(a.steps), "--read-len", str(L)])
