#!/usr/bin/env python3
"""What the FIRST `search` of a process costs beyond the steady state: a fresh process per case, the e2e step of
bench.py (reads in host memory -> cid_kmerset for the index -> search + report, 1 M reads of 150 bp, 256 colours) four times, with the
library's scratch allocations traced (CID_ALLOC_TRACE=1).  Cases: cold = nothing warmed; warmed = cid_warmup(CID_WARM_SEARCH | CID_WARM_PIPES) first (code
objects + a dry run of a few reads on a context of its own + this context's queues); small_first = warmed + a query of 2 000 reads before;
host_touched = warmed + the reads' host buffer copied to the device once by the CALLER before the library sees it; pinned_input = warmed +
the reads in page-locked memory from cid_pinned_alloc (where the command line keeps its batches).
   python3 tools/exp_first_use.py > gpurun_out/r06_first_use.txt"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, math
sys.path.insert(0, %r)
import numpy as np, torch
import bench, colorid_amd
from colorid_amd._lib import check, vp
case = sys.argv[1]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
t = time.perf_counter()
ctx = colorid_amd.Context(0)
C, n, k, m, R, L = 256, 4, 31, 50_000_000, 1_000_000, 150
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 0.2, seed=3)
hx.finalize()
print("setup_ms", round((time.perf_counter() - t) * 1e3, 1), flush=True)
rng = np.random.default_rng(1)
reads = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(R, L))].copy()
so = (np.arange(R + 1, dtype=np.uint64) * L)
if case != "cold":
    t = time.perf_counter(); check(ctx.lib.cid_warmup(ctx.h, 2 | 32)); print("warmup_ms", round((time.perf_counter() - t) * 1e3, 1), flush=True)
if case == "small_first":   # a query of 2 000 reads first: whatever does not grow with the query is paid there
    t = time.perf_counter()
    ks = colorid_amd.KmerSet(ctx, k); ks.set_target_index(hx)
    check(ks.lib.cid_kmerset_add_seqs(ks.h, vp(reads.ctypes.data), vp(so.ctypes.data), 2000, 0)); ks.finalize(); ks.search_count_report(hx); ks.close()
    print("small_query_ms", round((time.perf_counter() - t) * 1e3, 1), flush=True)
if case == "host_touched":   # the reads' host pages through a plain copy first (not the library's): is it the CALLER's buffer that is new to the bus?
    t = time.perf_counter(); tmp = torch.from_numpy(reads).to(dev); torch.cuda.synchronize(); del tmp
    print("plain_copy_of_the_reads_ms", round((time.perf_counter() - t) * 1e3, 1), flush=True)
if case == "pinned_input":   # the reads where the command line keeps its batches: page-locked memory from cid_pinned_alloc
    import ctypes
    pp = ctypes.c_void_p()
    check(ctx.lib.cid_pinned_alloc(reads.nbytes, ctypes.byref(pp)))
    ctypes.memmove(pp.value, reads.ctypes.data, reads.nbytes)
    reads = np.ctypeslib.as_array(ctypes.cast(pp.value, ctypes.POINTER(ctypes.c_uint8)), shape=(R * L,)).reshape(R, L)
for it in range(4):
    ks = colorid_amd.KmerSet(ctx, k); ks.set_target_index(hx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(ks.lib.cid_kmerset_add_seqs(ks.h, vp(reads.ctypes.data), vp(so.ctypes.data), R, 0)); t1 = time.perf_counter()
    nd = ks.finalize(); t2 = time.perf_counter()
    rep = ks.search_count_report(hx); t3 = time.perf_counter()
    print("iter", it, "ms", round((t3 - t0) * 1e3, 2), "upload", round((t1 - t0) * 1e3, 2), "sort", round((t2 - t1) * 1e3, 2), "search", round((t3 - t2) * 1e3, 2), flush=True)
    sys.stderr.write("--- end of iteration %%d\n" %% it)
    ks.close()
''' % ROOT
for case in (sys.argv[1:] or ["cold", "warmed", "small_first", "host_touched", "pinned_input"]):
    p = subprocess.run([sys.executable, "-c", CHILD, case], capture_output=True, text=True, env=dict(os.environ, CID_ALLOC_TRACE="1"))
    print(f"== {case} (rc {p.returncode})")
    print(p.stdout.strip())
    if p.returncode:
        print(p.stderr[-1500:])
    per_iter, cur = [], []
    for ln in p.stderr.splitlines():
        if ln.startswith("--- end of iteration"):
            per_iter.append(cur); cur = []
        elif ln.startswith("cid alloc"):
            m = re.search(r"(\d+) B in ([\d.]+) ms", ln)
            cur.append((int(m.group(1)), float(m.group(2))))
    for i, a in enumerate(per_iter):
        print(f"   iteration {i}: {len(a)} hipMalloc calls, {sum(x for x, _ in a) / 1e9:.2f} GB, {sum(y for _, y in a):.1f} ms" + (": " + ", ".join(f"{x >> 20} MB {y:.1f}" for x, y in sorted(a, reverse=True)[:8]) if a else ""))
