#!/usr/bin/env python3
"""Which read lengths the long-read path (cid_readlong.hip) should take: 150 Mbases of reads of one length, resident in HBM, through
cid_readid_count_resident with every read forced onto the per-wave LDS kernels (where they fit) and onto the long-read path
(cid_ctx_tune readid_long_from).  configs[2]'s index.  Output: one JSON line per length -> profiles/r05_readlen_route.jsonl"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import colorid_amd

dev = torch.device("cuda", 0)
ctx = colorid_amd.Context(0)
C, n, k, m = 256, 2, 21, 30_000_000
hx = colorid_amd.Index(ctx, m, n, k, C)
ptr, rs = hx.device_matrix()
bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
hx.finalize()
total = 150_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
bases = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)[torch.randint(0, 4, (total,), device=dev, generator=g)].contiguous()
for L in [int(x) for x in (sys.argv[1:] or "150 300 600 1000 1500 2000 2600 4000 10000 30000 100000 1000000".split())]:
    R = total // L
    seq_off = (np.arange(R + 1, dtype=np.uint64) * L)
    r0 = np.arange(R + 1, dtype=np.uint64)
    rep = torch.empty((R, C + 1), dtype=torch.int32, device=dev)
    nk = torch.empty(R, dtype=torch.int32, device=dev); st = torch.empty(R, dtype=torch.uint8, device=dev)
    out = {"read_len": L, "reads": R}
    ref = None
    for name, frm in (("lds_ms", 1 << 40), ("long_ms", 0), ("shipped_ms", -1)):
        ctx.tune("readid_long_from", frm)
        try:
            ts = []
            for i in range(5):
                torch.cuda.synchronize(); t = time.perf_counter()
                hx.readid_count_resident(bases.data_ptr(), seq_off, r0, 1, 3, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            out[name] = round(float(np.mean(ts[2:])), 3)
            dig = (int(rep.to(torch.int64).sum().item()), int(nk.to(torch.int64).sum().item()))
            if ref is None: ref = dig
            out["same"] = out.get("same", True) and dig == ref
        except colorid_amd.CidError as e:
            out[name] = None
    print(json.dumps(out), flush=True)
