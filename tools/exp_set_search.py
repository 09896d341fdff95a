#!/usr/bin/env python3
"""The search step of `colorid search` over a device-resident k-mer set, for rocprofv3 (tools/profile_set_search.sh): the headline's
index (m = 50 M, n = 4, k = 31, 256 colours, Bernoulli background + planted k-mers) and the distinct canonical 31-mers of 1 M reads,
the set built in code order (argv[1] = code) or FOR the index (argv[1] = target: cid_kmerset_set_target_index).  Prints the search's
HIP-event time; the k_search_count dispatches of the run are the ones the profiler's counters describe."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import colorid_amd
from colorid_amd._lib import check, vp

how = sys.argv[1] if len(sys.argv) > 1 else "target"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0)
ctx.set_stream(stream.cuda_stream)
C, n, k, m = int(os.environ.get('EXP_COLOURS', 256)), 4, 31, 50_000_000
R = int(os.environ.get('EXP_READS', 1_000_000))
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, R, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
ctx.synchronize()
hx.finalize()
del kk, ff, cc
host_reads = reads.cpu().numpy()
so = np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150
ks = colorid_amd.KmerSet(ctx, k)
if how == "target":
    ks.set_target_index(hx)
check(ks.lib.cid_kmerset_add_seqs(ks.h, vp(host_reads.ctypes.data), vp(so.ctypes.data), host_reads.shape[0], 0))
nd = ks.finalize()
import ctypes
d_codes, d_counts, nn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
check(ks.lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)
uc = torch.empty(nd, dtype=torch.int32, device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "report":   # the report call of bench.py's e2e record (search + modes + one copy), wall time per call
    import time
    walls = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        rep = ks.search_count_report(hx)
        walls.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(json.dumps({"set": how, "distinct_kmers": int(nd), "report_ms": walls}))
    sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(steps + 1):
    if i == 1:
        e0.record(stream)
    check(hx.lib.cid_search_count_codes_dev(ctx.h, hx.h, d_codes, d_counts, nd, vp(out.data_ptr()), vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
e1.record(stream)
torch.cuda.synchronize()
print(json.dumps({"set": how, "distinct_kmers": int(nd), "search_ms": e0.elapsed_time(e1) / steps, "hits": int(out[:C].sum().item())}))
