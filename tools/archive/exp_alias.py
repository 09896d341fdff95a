#!/usr/bin/env python3
"""Does k_search_count's time depend on WHERE its streamed arrays live?  (tools/exp_unroll.py showed 9.72 vs 10.17 ms for the
same kernel with the per-k-mer output at two different addresses.)  configs[1] shape; the unique-colour output is placed at a
series of offsets inside one pool, then in fresh allocations; the same for the multiplicity input.
usage: python tools/exp_alias.py [out.jsonl]"""
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

def run(uc_ptr, ff_ptr, kk_ptr, tag, **extra):
    ms = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        hx.search_count_dev(kk_ptr, ff_ptr, K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc_ptr)
        e1.record(stream); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    t = sorted(ms[2:])[len(ms[2:]) // 2]
    row = {"what": tag, "ms": round(t, 3), "uc": hex(uc_ptr), "freq": hex(ff_ptr), "kmers": hex(kk_ptr), "matrix": hex(ptr), **extra}
    print(json.dumps(row), flush=True)
    if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()

pool = torch.empty(K * 4 + (96 << 20), dtype=torch.uint8, device=dev)
for off in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, (32 << 20) + (1 << 18), 64 << 20):
    run(pool.data_ptr() + off, ff.data_ptr(), kk.data_ptr(), "uc at pool+off", off=off)
run(0, ff.data_ptr(), kk.data_ptr(), "no per-k-mer output")
keep = []
for i in range(6):
    t = torch.empty(K, dtype=torch.int32, device=dev); keep.append(t)
    run(t.data_ptr(), ff.data_ptr(), kk.data_ptr(), "uc in a fresh allocation", i=i)
fpool = torch.empty(K * 4 + (96 << 20), dtype=torch.uint8, device=dev)
for off in (0, 4096, 1 << 20, (2 << 20) + 4096, 64 << 20):
    fv = fpool[off:off + 4 * K].view(torch.int32); fv.copy_(ff)
    run(keep[0].data_ptr(), fv.data_ptr(), kk.data_ptr(), "freq at pool+off", off=off)
kpool = torch.empty(K * k + (96 << 20), dtype=torch.uint8, device=dev)
for off in (0, 4096, 1 << 20, (2 << 20) + 4096, 64 << 20):
    kv = kpool[off:off + k * K]; kv.copy_(kk.view(-1))
    run(keep[0].data_ptr(), ff.data_ptr(), kv.data_ptr(), "kmers at pool+off", off=off)
