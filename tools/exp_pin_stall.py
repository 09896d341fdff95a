#!/usr/bin/env python3
"""What page-locking memory costs, and what it costs the OTHER threads' runtime calls meanwhile (round 6: should the block-gzip stretches of
the device front end lie in page-locked memory?).  Thread A copies 64 MB host -> device from pageable memory in a loop and times every copy
and a small kernel launch; thread B, after a second, page-locks six fresh buffers of 70 MB (hipHostMalloc through cid_pinned_alloc, and
hipHostRegister of malloc'd memory through torch).  Run on the GPU box: python3 tools/exp_pin_stall.py"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from colorid_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda", 0)
src = torch.empty(64 << 20, dtype=torch.uint8); src.fill_(1)
dst = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
small = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
stop = False
log = []
def copier():
    while not stop:
        t = time.perf_counter(); dst.copy_(src); torch.cuda.synchronize(); t1 = time.perf_counter()
        small.add_(1); torch.cuda.synchronize(); t2 = time.perf_counter()
        log.append((t, (t1 - t) * 1e3, (t2 - t1) * 1e3))
th = threading.Thread(target=copier); th.start()
time.sleep(1.0)
marks = []
for kind in ("hipHostMalloc", "hipHostRegister"):
    for i in range(6):
        t = time.perf_counter()
        if kind == "hipHostMalloc":
            p = C.c_void_p()
            rc = lib.cid_pinned_alloc(70 << 20, C.byref(p)); assert rc == 0
        else:
            b = torch.empty(70 << 20, dtype=torch.uint8); b.fill_(0)
            t = time.perf_counter()
            rc = torch.cuda.cudart().cudaHostRegister(b.data_ptr(), b.numel(), 0)
        marks.append((kind, t, (time.perf_counter() - t) * 1e3))
        time.sleep(0.05)
    time.sleep(0.5)
stop = True; th.join()
base = [c for (t, c, k) in log if t < marks[0][1] - 0.2]
print(f"copies of 64 MB from pageable memory before any locking: median {np.median(base):.2f} ms, max {max(base):.2f} ms ({len(base)} copies)")
for kind, t, ms in marks:
    during = [(c, k) for (tt, c, k) in log if tt + c / 1e3 > t and tt < t + ms / 1e3]
    print(f"{kind} of 70 MB: {ms:.1f} ms; copies meanwhile: {[round(c, 1) for c, k in during]} ms, small kernels {[round(k, 2) for c, k in during]} ms")
