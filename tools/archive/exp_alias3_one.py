#!/usr/bin/env python3
"""For counter collection: the 8 indexes of tools/exp_alias3.py, then k_search_count 3x against index 0 and 3x against index 7
(dispatch order: 0,0,0,7,7,7), per-k-mer output on."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
C, n, m, k = 256, 4, 50_000_000, 31
kk, ff, cc, codes = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_codes=True)
K = kk.shape[0]
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
uc = torch.empty(K, dtype=torch.int32, device=dev)
pad, idx = [], []
for i in range(8):
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / 50_000_000), seed=7)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), K); ctx.synchronize(); hx.finalize()
    idx.append((hx, ptr))
    pad.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device=dev))
for i in (0, 7):
    hx = idx[i][0]
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr())
        e1.record(stream); torch.cuda.synchronize()
        print(json.dumps({"index": i, "rep": rep, "ms": round(e0.elapsed_time(e1), 3), "matrix": hex(idx[i][1])}), flush=True)
