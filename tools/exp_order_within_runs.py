#!/usr/bin/env python3
"""Does k_search_count care about the ORDER of the k-mers inside a stretch of a set built for the index, or only about which k-mers share a
stretch?  The set of 1 M reads (120 M distinct 31-mers, (first row, code) order) searched as it is, and with the k-mers of every block of B
consecutive ones shuffled (B = 256 ... 65536): same counters, HIP-event time per search.  -> profiles/r05_order_within_runs.json"""
import ctypes, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import colorid_amd
from colorid_amd._lib import check, vp

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0)
ctx.set_stream(stream.cuda_stream)
C, n, k, m = 256, 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
ctx.synchronize()
hx.finalize()
del kk, ff, cc
host_reads = reads.cpu().numpy()
so = np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150
ks = colorid_amd.KmerSet(ctx, k)
ks.set_target_index(hx)
check(ks.lib.cid_kmerset_add_seqs(ks.h, vp(host_reads.ctypes.data), vp(so.ctypes.data), host_reads.shape[0], 0))
nd = ks.finalize()
d_codes, d_counts, nn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
check(ks.lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
codes = torch.empty(nd, dtype=torch.int64, device=dev)
counts = torch.empty(nd, dtype=torch.int32, device=dev)
bench.hip_memcpy(codes.data_ptr(), d_codes.value, 8 * nd, 3)
bench.hip_memcpy(counts.data_ptr(), d_counts.value, 4 * nd, 3)
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
uc = torch.empty(nd, dtype=torch.int32, device=dev)

def search(cd, ct, steps=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(steps + 1):
        if i == 1:
            e0.record(stream)
        check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(cd.data_ptr()), vp(ct.data_ptr()), nd, vp(out.data_ptr()), vp(out.data_ptr() + 8 * C),
                                                vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out.clone()

rows = []
t0, ref = search(codes, counts)
rows.append({"order": "(first row, code)", "search_ms": round(t0, 3)})
g = torch.Generator(device=dev); g.manual_seed(3)
for B in (256, 2048, 16384, 65536):
    nb = nd // B
    perm = torch.rand((nb, B), device=dev, generator=g).argsort(dim=1) + (torch.arange(nb, device=dev) * B)[:, None]
    idx = torch.cat([perm.reshape(-1), torch.arange(nb * B, nd, device=dev)])
    t, o = search(codes[idx].contiguous(), counts[idx].contiguous())
    rows.append({"order": f"shuffled inside blocks of {B}", "search_ms": round(t, 3), "same_counters": bool(torch.equal(o, ref))})
    del perm, idx
idx = torch.randperm(nd, device=dev, generator=g)
t, o = search(codes[idx].contiguous(), counts[idx].contiguous())
rows.append({"order": "shuffled altogether", "search_ms": round(t, 3), "same_counters": bool(torch.equal(o, ref))})
print(json.dumps({"distinct_kmers": int(nd), "rows": rows}))
