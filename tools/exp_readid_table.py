#!/usr/bin/env python3
"""k_readid's per-read k-mer set with 12-byte slots (code + window index) and with ONE u64 per slot (code << idx_bits | index;
cid_ctx_tune "readid_packed_table"), A/B in one process on configs[2]'s shape (m = 30 M, n = 2, k = 21, 256 colours, 1 M x 150 bp),
single-end and paired, and at k = 27 / k = 31 (31 cannot pack).  Reports must be identical.
usage: python tools/exp_readid_table.py [out.jsonl]"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, colorid_amd
from colorid_amd._lib import check

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
lib = ctx.lib
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for k in (21, 27, 31):
    C, n, m, R = 256, 2, 30_000_000, 1_000_000
    for mates in (1, 2):
        hx = colorid_amd.Index(ctx, m, n, k, C)
        ptr, rs = hx.device_matrix()
        bench.fill_background_fast(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
        kk, ff, cc, reads = bench.make_reads_kmers(dev, 42, R * mates, 150, k, C, 0.01, return_reads=True)
        torch.cuda.synchronize()
        hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0]); ctx.synchronize(); hx.finalize()
        del kk, ff, cc
        bases = reads.reshape(-1).contiguous()
        so = (torch.arange(R * mates + 1, device=dev, dtype=torch.int64) * 150).contiguous()
        r0 = (torch.arange(R + 1, device=dev, dtype=torch.int64) * mates).contiguous()
        rep = torch.empty((R, C + 1), dtype=torch.int32, device=dev); nk = torch.empty(R, dtype=torch.int32, device=dev); st = torch.empty(R, dtype=torch.uint8, device=dev)
        nwin = (150 - k + 1) * mates
        ref = None
        for rnd in range(2):
            for packed in (0, 1):
                ctx.tune("readid_packed_table", packed)
                ms = []
                for rep_i in range(8):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    hx.readid_count_dev(bases.data_ptr(), so.data_ptr(), r0.data_ptr(), R, 1, 3, 150 * mates, nwin, rep.data_ptr(), nk.data_ptr(), st.data_ptr())
                    e1.record(stream); torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                res = (rep.clone(), nk.clone())
                if ref is None: ref = res
                same = bool(torch.equal(ref[0], res[0]) and torch.equal(ref[1], res[1]))
                del res
                row = {"k": k, "mates": mates, "packed_table": packed, "round": rnd, "ms": round(sorted(ms[2:])[3], 3), "same_report": same}
                print(json.dumps(row), flush=True)
                if out_f: out_f.write(json.dumps(row) + "\n"); out_f.flush()
        ctx.tune("readid_packed_table", 1)
        del hx, reads, bases, rep, ref
        torch.cuda.empty_cache()
