#!/usr/bin/env python3
"""k_search_count switches, A/B in one process with every array at a fixed address (the time of this kernel moves by several
per cent with the allocation its streamed arrays sit in — tools/exp_alias*.py — so variants must share them):
  search_prefetch 0/1 — tile t+1's k-mer and multiplicity loads issued before tile t is worked on,
  search_unroll 1/2/4 — sub-passes whose row loads are issued together (rows of 64 bytes and more).
C = 256 (configs[1]), 512 x n=3 (the stripe of configs[4]), 1024 (configs[3]), 2048; ASCII and 2-bit-code input.
usage: python tools/exp_unroll.py [out.jsonl]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
lib = ctx.lib
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for C, n, m in ((256, 4, 50_000_000), (512, 3, 1 << 28), (1024, 4, 50_000_000), (2048, 4, 50_000_000)):
    k = 31
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / 50_000_000), seed=7)
    kk, ff, cc, codes = bench.make_reads_kmers(dev, 42, 1_000_000, 150, k, C, 0.01, return_codes=True)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
    K = kk.shape[0]
    out = torch.zeros(3 * C, dtype=torch.int64, device=dev); uc = torch.empty(K, dtype=torch.int32, device=dev)
    ref = None
    variants = [(0, 1), (1, 1), (0, 2), (1, 2), (1, 4)] if C > 256 else [(0, 1), (1, 1)]
    for use_codes in (False, True):
        for rnd in range(2):
            for prefetch, unroll in variants:
                ctx.tune("search_unroll", unroll)
                ms = []
                for rep in range(8):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    if use_codes:
                        check(lib.cid_search_count_codes_dev(ctx.h, hx.h, vp(codes.data_ptr()), vp(ff.data_ptr()), K, vp(out.data_ptr()),
                                                             vp(out.data_ptr() + 8 * C), vp(out.data_ptr() + 16 * C), vp(uc.data_ptr())))
                    else:
                        hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C, uc.data_ptr())
                    e1.record(stream); torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                res = (out.clone(), uc.clone())
                if ref is None: ref = res
                same = bool(torch.equal(ref[0], res[0]) and torch.equal(ref[1], res[1]))
                del res
                t = sorted(ms[2:])[len(ms[2:]) // 2]
                row = {"n_colors": C, "row_bytes": rs * 8, "n_hash": n, "input": "codes" if use_codes else "ascii", "prefetch": prefetch,
                       "unroll": unroll, "round": rnd, "ms": round(t, 3), "same_result": same, "G_rows_per_s": round(K * n / t / 1e6, 2)}
                print(json.dumps(row), flush=True)
                if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()
    ctx.tune("search_unroll", 1); 
    del hx, kk, ff, cc, codes, out, uc, ref
    torch.cuda.empty_cache()
