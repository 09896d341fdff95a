#!/usr/bin/env python3
"""Experiment: cid_search_count over a device-resident k-mer set (2-bit codes) in code order vs ordered by the index line
of each k-mer's first row (cid_kmerset_order_for_index), incl. the cost of counting and of the reordering."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, colorid_amd

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = colorid_amd.Context(0)
C, n, k, m = 256, 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
host_reads = reads.cpu().numpy()
seqs_off = (np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150)
ks = colorid_amd.KmerSet(ctx, k)
t = time.perf_counter()
from colorid_amd._lib import check, vp
check(ks.lib.cid_kmerset_add_seqs(ks.h, host_reads.ctypes.data_as(vp), seqs_off.ctypes.data_as(vp), host_reads.shape[0], 0))
nd = ks.finalize()
t_count = time.perf_counter() - t
res = {"distinct": nd, "count_s_incl_h2d": t_count}
def timed():
    hits = np.zeros(C, np.uint64); nu = np.zeros(C, np.uint64); sf = np.zeros(C, np.uint64); uc = np.zeros(nd, np.uint32)
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        check(ks.lib.cid_search_count_set(ctx.h, hx.h, ks.h, hits.ctypes.data_as(vp), nu.ctypes.data_as(vp), sf.ctypes.data_as(vp), uc.ctypes.data_as(vp)))
        best = min(best, time.perf_counter() - t)
    return best, hits
res["search_set_code_order_s_incl_d2h"], h1 = timed()
t = time.perf_counter(); ks.order_for_index(hx); res["reorder_s"] = time.perf_counter() - t
res["search_set_row0_order_s_incl_d2h"], h2 = timed()
res["same_hits"] = bool(np.array_equal(h1, h2))
print(json.dumps(res))
