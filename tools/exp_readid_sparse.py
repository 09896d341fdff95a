#!/usr/bin/env python3
"""Where the time of the host-pointer read_id call goes (cid_readid_count_sparse + cid_readid_sparse_fetch, what the CLI's GPU stage
runs per batch): configs[2]'s shape, 1 M x 150 bp reads in host memory, in one call and in 20 calls of 50 000 reads.
COLORID_TIMING-style wall times around the two ABI calls; run under rocprofv3 --stats for the device side."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import colorid_amd
import ctypes as C
from colorid_amd._lib import check

dev = torch.device("cuda", 0)
ctx = colorid_amd.Context(0)
Cc, n, k, m, R, L = 256, int(os.environ.get('EXP_N', 2)), int(os.environ.get('EXP_K', 21)), int(os.environ.get('EXP_M', 30_000_000)), 1_000_000, 150
hx = colorid_amd.Index(ctx, m, n, k, Cc)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, Cc, 1.0 - math.exp(-n * 5e6 / m), seed=7)
kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, R, L, k, Cc, 0.01, return_reads=True)
torch.cuda.synchronize()
hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
ctx.synchronize()
hx.finalize()
bases = reads.reshape(-1).cpu().numpy()
seq_off = (np.arange(R + 1, dtype=np.uint64) * L)
read0 = np.arange(R + 1, dtype=np.uint64)
lib = hx.lib
p = lambda a: a.ctypes.data_as(C.c_void_p)


FRESH = False


def call(lo, hi):
    nr = hi - lo
    so = seq_off[lo:hi + 1] - seq_off[lo]
    r0 = read0[lo:hi + 1] - read0[lo]
    b = bases[lo * L:hi * L]
    if FRESH:   # a buffer the runtime has never seen (what a CLI batch is)
        b = b.copy()
    nk = np.zeros(nr, np.uint32); st = np.zeros(nr, np.uint8); ne = C.c_uint64(0)
    t0 = time.perf_counter()
    check(lib.cid_readid_count_sparse(ctx.h, hx.h, p(b), p(so), nr, p(r0), nr, 1, 3, p(nk), p(st), C.byref(ne)))
    t1 = time.perf_counter()
    rsx = np.zeros(nr + 1, np.uint64); col = np.zeros(ne.value, np.uint32); cnt = np.zeros(ne.value, np.uint32)
    t2 = time.perf_counter()
    check(lib.cid_readid_sparse_fetch(ctx.h, p(rsx), p(col), p(cnt)))
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t3 - t2) * 1e3, ne.value


res = {}
# the first calls of a process, one by one (a CLI run is all "first calls": scratch slots grow, rocPRIM temp storage is sized, ...)
first = []
if os.environ.get("EXP_WARMUP"):
    t0 = time.perf_counter()
    check(lib.cid_warmup(ctx.h, 1))
    res["warmup_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
if os.environ.get("EXP_TINY_FIRST"):
    a, b, e = call(0, 1)
    res["tiny_first_call_ms"] = round(a + b, 2)
for lo in range(0, R, 50_000):
    a, b, e = call(lo, min(R, lo + 50_000))
    first.append(round(a + b, 2))
res["first_20_calls_ms"] = first
for name, step in (("one_call", R), ("calls_of_50000", 50_000), ("calls_of_250000", 250_000)):
    for rep in range(3):
        tc = tf = 0.0; ent = 0
        for lo in range(0, R, step):
            a, b, e = call(lo, min(R, lo + step))
            tc += a; tf += b; ent += e
        res[name] = {"count_sparse_ms": round(tc, 1), "sparse_fetch_ms": round(tf, 1), "entries": ent}
FRESH = True
for rep in range(2):
    tc = tf = 0.0
    for lo in range(0, R, 50_000):
        a, b, e = call(lo, min(R, lo + 50_000))
        tc += a; tf += b
    res["calls_of_50000_fresh_buffers"] = {"count_sparse_ms": round(tc, 1), "sparse_fetch_ms": round(tf, 1)}
print(json.dumps(res))
