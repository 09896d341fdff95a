#!/usr/bin/env python3
"""Follow-up to tools/exp_alias.py: is a 'slow' allocation slow by itself (plain fill / read bandwidth), or only as the output of
k_search_count?  12 separate 480 MB allocations; per allocation: the kernel's time with the per-k-mer output there, a fill, a read.
usage: python tools/exp_alias2.py [out.jsonl]"""
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
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / 50_000_000), seed=7)
kk, ff, cc, codes = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_codes=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
K = kk.shape[0]
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
del cc, codes
torch.cuda.empty_cache()

def timed(fn, reps=7):
    ms = []
    for rep in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms[2:])[len(ms[2:]) // 2]

segs = [torch.empty(K + 1024 * i, dtype=torch.int32, device=dev) for i in range(12)]   # distinct sizes: one hipMalloc each
for i, t in enumerate(segs):
    ks = timed(lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, t.data_ptr()))
    fill = timed(lambda: t.fill_(7))
    rd = timed(lambda: t.sum())
    row = {"seg": i, "ptr": hex(t.data_ptr()), "search_ms": round(ks, 3), "fill_ms": round(fill, 3), "fill_GBs": round(t.numel() * 4 / fill / 1e6, 0),
           "read_ms": round(rd, 3), "read_GBs": round(t.numel() * 4 / rd / 1e6, 0)}
    print(json.dumps(row), flush=True)
    if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()
base = timed(lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, 0))
print(json.dumps({"no_output_ms": round(base, 3)}))
