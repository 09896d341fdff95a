#!/usr/bin/env python3
"""cid_readid_count on a 48 MB batch of 10 kb reads whose bases lie in pageable and in page-locked memory (cid_pinned_alloc): the library
copies page-locked bases to the device from where they are (round 6) instead of through its staging arena: 5.3 -> 3.6 ms per call.
(The command line does not use it: batches of long reads in page-locked blocks were measured there — alternating runs on one box, 150
Mbases of FASTA: 70-84 ms with the batches on the heap, 86-105 with them page-locked; profiles/HISTORY.md.)
Run on the GPU box: python3 tools/exp_locked_bases.py"""
import ctypes as C, os, sys, time, math
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench, colorid_amd
from colorid_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda", 0)
ctx = colorid_amd.Context(0)
Cn, n, k, m = 256, 2, 21, 30_000_000
hx = colorid_amd.Index(ctx, m, n, k, Cn)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, Cn, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
hx.finalize()
total = 48_000_000; L = 10_000; R = total // L
rng = np.random.default_rng(1)
src = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total)]
p = C.c_void_p(); assert lib.cid_pinned_alloc(total + 64, C.byref(p)) == 0
locked = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(total + 64,))[16:16 + total]
locked[:] = src
seq_off = (np.arange(R + 1, dtype=np.uint64) * L); r0 = np.arange(R + 1, dtype=np.uint64)
for name, arr in (("pageable", src), ("page-locked", locked), ("pageable", src), ("page-locked", locked)):
    ts = []
    for i in range(6):
        t = time.perf_counter(); out = hx.readid_count_sparse(arr, seq_off, r0, 1, 3) if hasattr(hx, "readid_count_sparse") else hx.readid_count(arr, seq_off, r0, 1, 3); ts.append((time.perf_counter() - t) * 1e3)
    print(name, [round(x, 1) for x in ts])
