#!/usr/bin/env python3
"""cid_readid_count_dev of TWO BUILDS of the library in one process (same box, same reads, same index contents), alternating:
configs[2]'s shape (m = 30 M, n = 2, 256 colours, 1 M x 150 bp), single-end and paired.  Reports must be identical.
CAVEAT: each build searches its OWN copy of the index, and where an allocation lands in HBM moves k_readid by up to +-5 % (two copies of
ONE build measured 9.15 against 9.65 ms): differences below that are not evidence — switch inside one build with cid_ctx_tune where a switch
exists (tools/exp_readid_tune.py: same index, same context).
usage: python tools/exp_readid_ab.py OTHER.so [out.jsonl]     (OTHER.so e.g. a build of an earlier commit; relative to the repo root)
env EXP_CASES=k:len[:mates],...   EXP_TUNE_NEW=name=value,...  (cid_ctx_tune settings of the current build)"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd
from colorid_amd._lib import open_library, LIB_PATH

other = os.path.join(ROOT, sys.argv[1])
out_f = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
builds = {"new": colorid_amd.Context(0), "old": colorid_amd.Context(0, lib=open_library(other))}
for c in builds.values(): c.set_stream(stream.cuda_stream)
for kv in filter(None, os.environ.get("EXP_TUNE_NEW", "").split(",")):
    builds["new"].tune(kv.split("=")[0], int(kv.split("=")[1]))
cases = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("EXP_CASES", "21:150,21:150:2").split(",")]
for case in cases:
    k, L = case[0], case[1]
    mates = case[2] if len(case) > 2 else 1
    C, n, m, R = 256, 2, 30_000_000, 1_000_000
    kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, R * mates, L, k, C, 0.01, return_reads=True)
    idx = {}
    for name, ctx in builds.items():
        hx = colorid_amd.Index(ctx, m, n, k, C)
        ptr, rs = hx.device_matrix()
        bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
        torch.cuda.synchronize()
        hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
        idx[name] = hx
    del kk, ff, cc
    bases = reads.reshape(-1).contiguous()
    so = (torch.arange(R * mates + 1, device=dev, dtype=torch.int64) * L).contiguous()
    r0 = (torch.arange(R + 1, device=dev, dtype=torch.int64) * mates).contiguous()
    rep = torch.empty((R, C + 1), dtype=torch.int32, device=dev); nk = torch.empty(R, dtype=torch.int32, device=dev); st = torch.empty(R, dtype=torch.uint8, device=dev)
    nwin = (L - k + 1) * mates
    ref = None
    for rnd in range(3):
        for name in ("old", "new"):
            hx = idx[name]
            ms = []
            for _ in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                hx.readid_count_dev(bases.data_ptr(), so.data_ptr(), r0.data_ptr(), R, 1, 3, L * mates, nwin, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
                e1.record(stream); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            res = (rep.clone(), nk.clone(), st.clone())
            if ref is None: ref = res
            same = bool(all(torch.equal(a, b) for a, b in zip(ref, res)))
            del res
            row = {"k": k, "read_len": L, "mates": mates, "build": name, "round": rnd, "ms": round(sorted(ms[2:])[3], 3), "same_report": same}
            print(json.dumps(row), flush=True)
            if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()
    for hx in idx.values(): hx.close()
    del idx, reads, bases, rep, ref
    torch.cuda.empty_cache()
