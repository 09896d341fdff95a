#!/usr/bin/env python3
"""Would the long-read path gain from working on two halves of a batch at once (one half's table kernel beside the other half's search)?
Two contexts on one GPU, each with its own copy of configs[2]'s index, each classifying 75 Mbases of 10 kb reads resident in HBM: one
after the other, and from two threads at once.  Run on the GPU box: python3 tools/exp_two_calls.py"""
import math, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, colorid_amd
dev = torch.device("cuda", 0)
C, n, k, m = 256, 2, 21, 30_000_000
L, half = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000, 75_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
acgt = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)
parts = []
for i in range(2):
    ctx = colorid_amd.Context(0)
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
    hx.finalize()
    bases = acgt[torch.randint(0, 4, (half,), device=dev, generator=g)].contiguous()
    R = half // L
    so = np.arange(R + 1, dtype=np.uint64) * L; r0 = np.arange(R + 1, dtype=np.uint64)
    rep = torch.empty((R, C + 1), dtype=torch.int32, device=dev); nk = torch.empty(R, dtype=torch.int32, device=dev); st = torch.empty(R, dtype=torch.uint8, device=dev)
    parts.append((ctx, hx, bases, so, r0, rep, nk, st))
torch.cuda.synchronize()
def call(p):
    ctx, hx, bases, so, r0, rep, nk, st = p
    hx.readid_count_resident(bases.data_ptr(), so, r0, 1, 3, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
for p in parts: call(p); call(p)
torch.cuda.synchronize()
for rnd in range(4):
    t = time.perf_counter(); call(parts[0]); call(parts[1]); torch.cuda.synchronize(); seq = (time.perf_counter() - t) * 1e3
    ths = [threading.Thread(target=call, args=(p,)) for p in parts]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize(); par = (time.perf_counter() - t) * 1e3
    print(f"two halves of 75 Mbases ({L} b reads): one after the other {seq:.2f} ms, at once {par:.2f} ms")
