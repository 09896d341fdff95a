#!/usr/bin/env python3
"""Experiment: which outputs of k_search_count cost time at a given colour count?  Times, interleaved and repeated:
full (hits + unique stats + unique_colour + multiplicities), no unique_colour array, no multiplicities, hits only.
Usage: python tools/exp_search_outputs.py --colours 1024"""
import argparse, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd

ap = argparse.ArgumentParser()
ap.add_argument("--colours", type=int, default=1024)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
C, n, k, m = a.colours, 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kmers, freq, colour = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01)
torch.cuda.synchronize()
hx.insert_kmers_dev(kmers.data_ptr(), colour.data_ptr(), kmers.shape[0]); ctx.synchronize(); hx.finalize()
K = kmers.shape[0]
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
uc = torch.empty(K, dtype=torch.int32, device=dev)
o = out.data_ptr()
variants = {
    "full": lambda: hx.search_count_dev(kmers.data_ptr(), freq.data_ptr(), K, o, o + 8 * C, o + 16 * C, uc.data_ptr()),
    "no_uc": lambda: hx.search_count_dev(kmers.data_ptr(), freq.data_ptr(), K, o, o + 8 * C, o + 16 * C, None),
    "no_freq": lambda: hx.search_count_dev(kmers.data_ptr(), None, K, o, o + 8 * C, o + 16 * C, uc.data_ptr()),
    "hits_only": lambda: hx.search_count_dev(kmers.data_ptr(), None, K, o, None, None, None),
}
res = {v: [] for v in variants}
for rep in range(a.reps):
    for name, fn in variants.items():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5): fn()
        e1.record(stream); torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / 5, 2))
for name, v in res.items():
    print(f"C={C} {name:10s} {v}  min {min(v)}")
