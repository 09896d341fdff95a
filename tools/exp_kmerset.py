#!/usr/bin/env python3
"""GPU k-mer counting of configs[1]'s query side (1 M x 150 bp reads -> ~120 M distinct 31-mers): wall time of
cid_kmerset_add_seqs (H2D of the reads + window codes) and cid_kmerset_finalize (sort + run-length), per call, a few times.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/profile_kmerset.sh)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import colorid_amd
from colorid_amd._lib import check, vp

n_reads, L, k = int(os.environ.get("EXP_READS", 1_000_000)), 150, 31
rng = np.random.default_rng(42)
if os.environ.get("EXP_GENOME"):     # reads of a random genome of that many bases (coverage n_reads * L / EXP_GENOME), 1 % substitutions
    g = rng.integers(0, 4, size=int(os.environ["EXP_GENOME"]) + L, dtype=np.uint8)
    at = rng.integers(0, g.shape[0] - L, size=n_reads)
    codes = g[at[:, None] + np.arange(L)[None, :]]
    err = rng.random(codes.shape) < 0.01
    codes = np.where(err, (codes + rng.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes)
    rc = rng.random(n_reads) < 0.5
    codes[rc] = 3 - codes[rc][:, ::-1]
    reads = np.frombuffer(b"ACGT", np.uint8)[codes]
else:
    reads = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n_reads, L))]
reads = np.ascontiguousarray(reads)
so = (np.arange(n_reads + 1, dtype=np.uint64) * L)
ctx = colorid_amd.Context(0)
target = None
if os.environ.get("EXP_TARGET"):     # the set built FOR an index (cid_kmerset_set_target_index): m = 50 M, n = 4, 256 colours
    target = colorid_amd.Index(ctx, 50_000_000, 4, k, 256)
    target.finalize()
rows = []
for it in range(int(os.environ.get("EXP_ITERS", 5))):
    ks = colorid_amd.KmerSet(ctx, k)
    if target is not None:
        ks.set_target_index(target)
    t0 = time.perf_counter()
    check(ks.lib.cid_kmerset_add_seqs(ks.h, reads.ctypes.data_as(vp), so.ctypes.data_as(vp), n_reads, 0))
    t1 = time.perf_counter()
    nd = ks.finalize()
    t2 = time.perf_counter()
    rows.append({"add_seqs_ms": round((t1 - t0) * 1e3, 2), "finalize_ms": round((t2 - t1) * 1e3, 2), "distinct": nd})
    if os.environ.get("EXP_DIGEST") and it == 0:     # the set's contents, to compare two settings of the library across processes
        import ctypes, hashlib
        d_codes, d_counts, nn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        check(ks.lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
        hc, hn = np.empty(nn.value, np.uint64), np.empty(nn.value, np.uint32)
        import bench
        bench.hip_memcpy(hc.ctypes.data, d_codes.value, 8 * nn.value, 2)
        bench.hip_memcpy(hn.ctypes.data, d_counts.value, 4 * nn.value, 2)
        rows[-1]["digest"] = hashlib.sha256(hc.tobytes() + hn.tobytes()).hexdigest()[:16]
        rows[-1]["windows_counted"] = int(hn.astype(np.uint64).sum())
    ks.close()
print(json.dumps({"reads": n_reads, "windows": n_reads * (L - k + 1), "iters": rows}))
