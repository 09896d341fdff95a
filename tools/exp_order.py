#!/usr/bin/env python3
"""Experiment (round 2): does grouping the query k-mers by the index slice of their first row pay, and how finely?
cid_search_count_codes_dev over the bench workload (m = 50 M, n = 4, C = 256, 120 M distinct 31-mers as 2-bit codes) with the
set in code order / grouped into 2^b slices / sorted by exact line, each with the plain and the persistent XCD-queue kernel.
Device times (HIP events); the reorder's own cost is reported next to it.  Usage: python tools/exp_order.py [bits ...]"""
import ctypes, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
from colorid_amd import _lib as _cl   # the TUNE build holds the persistent scheduling (make -C colorid_amd/csrc tune)
ctx = colorid_amd.Context(0, lib=_cl.open_library(_cl.TUNE_LIB_PATH))
ctx.set_stream(stream.cuda_stream)
lib = ctx.lib
C, n, k, m = int(os.environ.get("EXP_C", 256)), 4, 31, 50_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
host_reads = reads.cpu().numpy(); so = (np.arange(host_reads.shape[0] + 1, dtype=np.uint64) * 150)
del kk, ff, cc, reads
out = torch.zeros(3 * C, dtype=torch.int64, device=dev)


def make_set():
    ks = colorid_amd.KmerSet(ctx, k)
    check(lib.cid_kmerset_add_seqs(ks.h, host_reads.ctypes.data_as(vp), so.ctypes.data_as(vp), host_reads.shape[0], 0))
    ks.finalize()
    return ks


def timed(ks, steps=10):
    d_codes, d_counts, nn = vp(), vp(), ctypes.c_uint64(0)
    check(lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
    K = nn.value
    uc = torch.empty(K, dtype=torch.int32, device=dev)
    def run():
        check(lib.cid_search_count_codes_dev(ctx.h, hx.h, d_codes, d_counts, K, vp(out.data_ptr()), vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        run()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out.cpu().numpy().copy(), K


# ASCII input (the bench headline's input form), both schedulings
kk, ff, cc = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01)
K0 = kk.shape[0]
uc0 = torch.empty(K0, dtype=torch.int32, device=dev)
for persist in (0, 1):
    ctx.tune("search_persist", persist)
    def run_ascii():
        hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K0, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc0.data_ptr())
    run_ascii(); run_ascii(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        run_ascii()
    e1.record(stream); torch.cuda.synchronize()
    print(json.dumps({"input": "ascii, random order", "persist": persist, "ms": round(e0.elapsed_time(e1) / 10, 3), "C": C}), flush=True)
del kk, ff, cc, uc0

res = []
ref = None
orders = [("code order", None)] + [(f"2^{b} slices", int(b)) for b in (sys.argv[1:] or ["8", "10", "12", "14", "16"])] + [("exact line", 0)]
for name, bits in orders:
    ks = make_set()
    t_re = 0.0
    if bits is not None:
        ctx.tune("order_bits", bits)
        torch.cuda.synchronize(); t = time.perf_counter(); ks.order_for_index(hx); t_re = time.perf_counter() - t
    row = {"order": name, "reorder_ms": round(t_re * 1e3, 2)}
    for persist in (0, 1):
        ctx.tune("search_persist", persist)
        ms, o, K = timed(ks)
        if ref is None:
            ref = o
        row["persist" if persist else "plain"] = round(ms, 3)
        row["same"] = bool(np.array_equal(o, ref)) and row.get("same", True)
    res.append(row)
    print(json.dumps(row), flush=True)
    ks.close()
