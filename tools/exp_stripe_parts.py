#!/usr/bin/env python3
"""Experiment: where does one rank's step of the colour-striped search (config E: m = 2^30, n = 3, a 512-colour stripe) spend its
time — clearing the per-k-mer arrays, the stripe kernel, the unique finalize?  Usage: python tools/exp_stripe_parts.py [log2_m]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
C, n, k, m = 512, 3, 31, 1 << lg
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
kk, ff, cc = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
K = kk.shape[0]
hits = torch.zeros(C, dtype=torch.int64, device=dev); fact = torch.zeros(K, dtype=torch.int32, device=dev)
nu = torch.zeros(4096, dtype=torch.int64, device=dev); sf = torch.zeros(4096, dtype=torch.int64, device=dev); uc = torch.empty(K, dtype=torch.int32, device=dev)
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
parts = {
    "clear": lambda: (fact.zero_(), nu.zero_(), sf.zero_()),
    "stripe_kernel": lambda: check(hx.lib.cid_search_count_stripe_dev(ctx.h, hx.h, vp(kk.data_ptr()), None, K, 1024, vp(hits.data_ptr()), vp(fact.data_ptr()))),
    "finalize": lambda: check(hx.lib.cid_search_unique_finalize_dev(ctx.h, vp(fact.data_ptr()), vp(ff.data_ptr()), K, 4096, vp(nu.data_ptr()), vp(sf.data_ptr()), vp(uc.data_ptr()))),
    "plain_search_same_index": lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr()),
}
for name, fn in parts.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(5): fn()
    e1.record(stream); torch.cuda.synchronize()
    print(f"m=2^{lg} {name:26s} {e0.elapsed_time(e1) / 5:.2f} ms")
