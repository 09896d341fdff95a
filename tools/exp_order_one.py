#!/usr/bin/env python3
"""One variant of tools/exp_order.py for counter collection: python tools/exp_order_one.py <order_bits|none> <persist 0|1> [steps]"""
import ctypes, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp

order, persist = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
from colorid_amd import _lib as _cl   # the TUNE build holds the persistent scheduling (make -C colorid_amd/csrc tune)
ctx = colorid_amd.Context(0, lib=_cl.open_library(_cl.TUNE_LIB_PATH))
ctx.set_stream(stream.cuda_stream)
lib = ctx.lib
C, n, k, m = 256, 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
host_reads = reads.cpu().numpy(); so = (np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150)
del kk, ff, cc, reads
ks = colorid_amd.KmerSet(ctx, k)
check(lib.cid_kmerset_add_seqs(ks.h, host_reads.ctypes.data_as(vp), so.ctypes.data_as(vp), host_reads.shape[0], 0))
ks.finalize()
if order != "none":
    ctx.tune("order_bits", int(order))
    ks.order_for_index(hx)
ctx.tune("search_persist", persist)
d_codes, d_counts, nn = vp(), vp(), ctypes.c_uint64(0)
check(lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
K = nn.value
out = torch.zeros(3 * C, dtype=torch.int64, device=dev); uc = torch.empty(K, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(steps):
    check(lib.cid_search_count_codes_dev(ctx.h, hx.h, d_codes, d_counts, K, vp(out.data_ptr()), vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
e1.record(stream); torch.cuda.synchronize()
print(json.dumps({"order": order, "persist": persist, "ms": e0.elapsed_time(e1) / steps, "kmers": K}))
