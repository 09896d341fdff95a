#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (never bench.py's `value`): cid_search_count with 120 M ASCII k-mers
in pageable host memory, and the reads -> cid_kmerset -> cid_search_count_set pipeline for the same 1 M reads."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = colorid_amd.Context(0)
C, n, k, m = 256, 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
hk = kk.cpu().numpy(); hf = ff.cpu().numpy().astype(np.uint32); K = hk.shape[0]
res = {"kmers": K}
best = 1e9
times = []
for _ in range(3):
    t = time.perf_counter(); out = hx.search_count(hk, hf); times.append(time.perf_counter() - t); best = min(best, times[-1])
res["cid_search_count_host_s"] = best; res["host_path_kmers_per_s"] = K / best
res["cid_search_count_host_calls_s"] = times   # the first call meets host buffers the runtime has never pinned
# the same call with the output arrays allocated (and touched) once by the caller, as a C++/Rust host would hold them: the
# Python wrapper above allocates 480 MB of untouched pages per call, whose first touch by the copy-out is timed with it
hits = np.zeros(C, np.uint64); nu = np.zeros(C, np.uint64); sf = np.zeros(C, np.uint64); uc = np.full(K, 7, np.uint32)
pre = []
for _ in range(3):
    t = time.perf_counter()
    check(hx.lib.cid_search_count(ctx.h, hx.h, hk.ctypes.data_as(vp), hf.ctypes.data_as(vp), K, hits.ctypes.data_as(vp), nu.ctypes.data_as(vp),
                                  sf.ctypes.data_as(vp), uc.ctypes.data_as(vp)))
    pre.append(time.perf_counter() - t)
res["cid_search_count_caller_buffers_s"] = pre; res["caller_buffers_kmers_per_s"] = K / min(pre)
res["caller_buffers_same"] = bool(np.array_equal(hits, out[0]) and np.array_equal(uc, out[3]))
host_reads = reads.cpu().numpy(); so = (np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150)
best = 1e9
for _ in range(3):
    ks = colorid_amd.KmerSet(ctx, k)
    t = time.perf_counter()
    check(ks.lib.cid_kmerset_add_seqs(ks.h, host_reads.ctypes.data_as(vp), so.ctypes.data_as(vp), host_reads.shape[0], 0))
    nd = ks.finalize()
    o2 = ks.search_count(hx)
    best = min(best, time.perf_counter() - t); ks.close()
res["reads_to_per_kmer_results_s"] = best; res["per_kmer_pipeline_distinct_kmers_per_s"] = nd / best
res["same_hits"] = bool(np.array_equal(np.sort(out[0]), np.sort(o2[0])))
# the CLI's default report path: hits, n_unique, sum and MODE per colour computed on the device, nothing per k-mer comes back
best = 1e9
for _ in range(4):
    ks = colorid_amd.KmerSet(ctx, k)
    t = time.perf_counter()
    check(ks.lib.cid_kmerset_add_seqs(ks.h, host_reads.ctypes.data_as(vp), so.ctypes.data_as(vp), host_reads.shape[0], 0))
    nd = ks.finalize()
    o3 = ks.search_count_report(hx)
    best = min(best, time.perf_counter() - t); ks.close()
res["reads_to_report_s"] = best; res["pipeline_distinct_kmers_per_s"] = nd / best; res["pipeline_reads_per_s"] = host_reads.shape[0] / best
res["report_same_hits"] = bool(np.array_equal(o3[0], o2[0]) and np.array_equal(o3[1], o2[1]) and np.array_equal(o3[2], o2[2]))
print(json.dumps(res))
