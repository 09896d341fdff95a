#!/usr/bin/env python3
"""Does the kernel's time depend on which allocation the INDEX sits in?  8 identical indexes (configs[1] shape) alive at once,
the same query arrays; the time of k_search_count against each (per-k-mer output on and off).
usage: python tools/exp_alias3.py [out.jsonl]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
C, n, m, k = 256, 4, 50_000_000, 31
kk, ff, cc, codes = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_codes=True)
K = kk.shape[0]
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
uc = torch.empty(K, dtype=torch.int32, device=dev)
pad = []
idx = []
for i in range(8):
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / 50_000_000), seed=7)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), K); ctx.synchronize(); hx.finalize()
    idx.append((hx, ptr))
    pad.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device=dev))   # odd-sized gaps between the indexes

def timed(fn, reps=7):
    ms = []
    for rep in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms[2:])[len(ms[2:]) // 2]

ref = None
for rnd in range(2):
    for i, (hx, ptr) in enumerate(idx):
        a = timed(lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr()))
        res = out.clone()
        if ref is None: ref = res
        b = timed(lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, 0))
        row = {"round": rnd, "index": i, "matrix": hex(ptr), "ms_with_output": round(a, 3), "ms_no_output": round(b, 3), "same": bool(torch.equal(ref, res))}
        print(json.dumps(row), flush=True)
        if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()
