#!/usr/bin/env python3
"""Measure the BASELINE.json configs' shapes on one MI355X (synthetic data; the driver's bench line stays bench.py):
  A' perfect search (`search -s`): 2.9 M distinct 31-mers of one genome vs m=50M n=4 C=46                      (k_search_perfect)
  B  proportional search, m=50M n=4 C=256, 120 M distinct k-mers of 1 M reads                                   (k_search_count; = bench.py)
  C  read_id, m=30M n=2 k=21 C=256, 1 M x 150 bp SE and PE                                                      (k_readid)
  D  one rank of config D: m=50M n=4 C=1024                                                                     (k_search_count, 128-byte rows)
  E  one rank of config E: m=2^30 n=3, a 512-colour stripe (64 GiB), striped search incl. the per-k-mer facts   (stripe kernel + finalize)
Each GPU figure is the device time of the resident call (HIP events); each CPU figure is the single-threaded oracle on a
bounded sample of the same inputs against a host copy of the same index, with the results compared bit for bit."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, colorid_amd
from colorid_amd._lib import check, vp
from oracle import orc

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = colorid_amd.Context(0); ctx.set_stream(stream.cuda_stream)
only = set(sys.argv[1:]) or set("ABCDE")
out = []


def timed(fn, steps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def make_index(C, n, k, m, genome_len, reads=1_000_000, want_reads=False):
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * genome_len / m), seed=7)
    r = bench.make_reads_kmers(dev, 42, reads, 150, k, C, 0.01, return_reads=want_reads)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(r[0].data_ptr(), r[2].data_ptr(), r[0].shape[0]); ctx.synchronize()
    return hx, ptr, rs, r


def host_index(hx, ptr, rs, C, n, k, m):
    oix = orc.Index(m, n, k, C)
    rows = oix.rows(); step = 4_000_000
    torch.cuda.synchronize()
    for r0 in range(0, m, step):
        nr = min(step, m - r0)
        blk = np.empty((nr, rs * 2), np.uint32)
        bench.hip_memcpy(blk.ctypes.data, ptr + r0 * rs * 8, blk.nbytes, 2)
        rows[r0:r0 + nr, :] = blk[:, :oix.w32]
    for c in range(C):
        oix.set_color(c, f"g{c}", 3_000_000)
    return oix


if "A" in only:
    C, n, k, m = 46, 4, 31, 50_000_000
    hx = colorid_amd.Index(ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
    g = torch.Generator(device=dev); g.manual_seed(5)
    genome = torch.randint(0, 4, (1, 2_944_528), device=dev, dtype=torch.int64, generator=g)   # EGD-e's length
    lut = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)
    hg = lut[genome[0]].cpu().numpy().tobytes()
    ks = colorid_amd.KmerSet(ctx, k); ks.add_seqs([hg], 0); K = ks.finalize()
    km, cnt = ks.download()
    dk = torch.from_numpy(km.reshape(-1)).to(dev)
    for col in (3, 17):                                        # the genome is "in" two of the 46 colours
        dc = torch.full((K,), col, dtype=torch.int32, device=dev); torch.cuda.synchronize()
        hx.insert_kmers_dev(dk.data_ptr(), dc.data_ptr(), K); ctx.synchronize()
    hx.finalize()
    d_codes, d_counts, _ = vp(), vp(), None
    import ctypes
    nn = ctypes.c_uint64(0)
    check(ks.lib.cid_kmerset_device_arrays(ks.h, ctypes.byref(d_codes), ctypes.byref(d_counts), ctypes.byref(nn)))
    d_and = torch.empty(rs, dtype=torch.int64, device=dev); d_zero = torch.full((K,), -1, dtype=torch.int32, device=dev)
    ms = timed(lambda: check(hx.lib.cid_search_perfect_stripe_dev(ctx.h, hx.h, None, d_codes, K, vp(d_and.data_ptr()), vp(d_zero.data_ptr()))))
    gw, gm = ks.search_perfect(hx)
    oix = host_index(hx, ptr, rs, C, n, k, m)
    t = time.perf_counter(); ow, om = oix.search_perfect(km); cpu_s = time.perf_counter() - t
    out.append({"config": "A' search -s (perfect), m=50M n=4 C=46, one 2.94 Mbp genome", "kmers": K, "gpu_ms": ms, "gpu_kmers_per_s": K / ms * 1e3,
                "alg_GBs": K * (n * 8 + 8) / ms / 1e6, "cpu_oracle_1thread_kmers_per_s": K / cpu_s,
                "bit_exact": bool(gm == om and np.array_equal(gw, ow)), "hits": [int(c) for c in range(C) if gw[c // 32] >> (c % 32) & 1]})
    hx.close(); ks.close(); del oix

for name, C in (("B", 256), ("D", 1024)):
    if name not in only:
        continue
    n, k, m = 4, 31, 50_000_000
    hx, ptr, rs, (kk, ff, cc) = make_index(C, n, k, m, 3_000_000)
    hx.finalize(); K = kk.shape[0]
    o = torch.zeros(3 * C, dtype=torch.int64, device=dev); uc = torch.empty(K, dtype=torch.int32, device=dev)
    ms = timed(lambda: hx.search_count_dev(kk.data_ptr(), ff.data_ptr(), K, o.data_ptr(), o.data_ptr() + 8 * C, o.data_ptr() + 16 * C, uc.data_ptr()))
    oix = host_index(hx, ptr, rs, C, n, k, m)
    S = 2_000_000
    hk, hf = kk[:S].cpu().numpy(), ff[:S].cpu().numpy()
    t = time.perf_counter(); want = oix.search_count(hk, hf.astype(np.uint64)); cpu_s = time.perf_counter() - t
    got = hx.search_count(hk, hf.astype(np.uint32))
    w64 = (C + 63) // 64
    out.append({"config": f"{name} search, m=50M n=4 k=31 C={C}, distinct k-mers of 1 M reads", "kmers": K, "gpu_ms": ms, "gpu_kmers_per_s": K / ms * 1e3,
                "alg_GBs": K * (n * w64 * 8 + k + 8) / ms / 1e6, "cpu_oracle_1thread_kmers_per_s": S / cpu_s,
                "bit_exact": all(np.array_equal(a, b) for a, b in zip(want, got))})
    hx.close(); del oix, kk, ff, cc, uc

if "C" in only:
    C, n, k, m = 256, 2, 21, 30_000_000
    for paired in (False, True):
        mates = 2 if paired else 1
        hx, ptr, rs, (kk, ff, cc, reads) = make_index(C, n, k, m, 5_000_000, reads=1_000_000 * mates, want_reads=True)
        hx.finalize()
        R = 1_000_000
        bases = reads.reshape(-1).contiguous()
        so = (torch.arange(R * mates + 1, device=dev, dtype=torch.int64) * 150).contiguous()
        r0 = (torch.arange(R + 1, device=dev, dtype=torch.int64) * mates).contiguous()
        rep = torch.empty((R, C + 1), dtype=torch.int32, device=dev); nk = torch.empty(R, dtype=torch.int32, device=dev); st = torch.empty(R, dtype=torch.uint8, device=dev)
        ms = timed(lambda: hx.readid_count_dev(bases.data_ptr(), so.data_ptr(), r0.data_ptr(), R, 1, 3, 150 * mates, 130 * mates, rep.data_ptr(), nk.data_ptr(), st.data_ptr()))
        oix = host_index(hx, ptr, rs, C, n, k, m)
        S = 20_000
        t = time.perf_counter()
        want = oix.readid_counts(bases[:S * 150 * mates].cpu().numpy(), so[:S * mates + 1].cpu().numpy().astype(np.uint64), r0[:S + 1].cpu().numpy().astype(np.uint64), 1, 3)
        cpu_s = time.perf_counter() - t
        ncore = os.cpu_count() or 1
        S2 = min(R, 2000 * ncore)      # all host cores, the reference's rayon pool (-t): a larger sample so every thread has work
        hb2, so2, r02 = bases[:S2 * 150 * mates].cpu().numpy(), so[:S2 * mates + 1].cpu().numpy().astype(np.uint64), r0[:S2 + 1].cpu().numpy().astype(np.uint64)
        t = time.perf_counter()
        want_mt = oix.readid_counts(hb2, so2, r02, 1, 3, n_threads=ncore)
        cpu_mt_s = time.perf_counter() - t
        mt_ok = bool(np.array_equal(want_mt[0], rep[:S2].cpu().numpy().view(np.uint32)))
        nks = int(nk.to(torch.int64).sum().item())
        out.append({"config": f"C read_id {'PE' if paired else 'SE'}, m=30M n=2 k=21 C=256, 1 M x 150 bp", "reads": R, "gpu_ms": ms, "gpu_reads_per_s": R / ms * 1e3,
                    "gpu_kmers_per_s": nks / ms * 1e3, "alg_GBs": (nks * n * rs * 8 + bases.numel() + R * (C + 1) * 4) / ms / 1e6,
                    "cpu_oracle_1thread_reads_per_s": S / cpu_s,
                    "cpu_oracle_all_cores_reads_per_s": S2 / cpu_mt_s, "cpu_cores": ncore, "cpu_all_cores_sample_reads": S2, "bit_exact_all_cores_sample": mt_ok,
                    "bit_exact": bool(np.array_equal(want[0], rep[:S].cpu().numpy().view(np.uint32)) and np.array_equal(want[1], nk[:S].cpu().numpy().view(np.uint32)))})
        hx.close(); del oix, kk, ff, cc, reads, rep

if "E" in only:
    C, n, k, m = 512, 3, 31, 1 << 30        # one GPU's stripe of the 4096-colour, 512 GiB index
    hx, ptr, rs, (kk, ff, cc) = make_index(C, n, k, m, 5_000_000)
    hx.finalize(); K = kk.shape[0]
    hits = torch.zeros(C, dtype=torch.int64, device=dev); fact = torch.zeros(K, dtype=torch.int32, device=dev)
    nu = torch.zeros(4096, dtype=torch.int64, device=dev); sf = torch.zeros(4096, dtype=torch.int64, device=dev); uc = torch.empty(K, dtype=torch.int32, device=dev)

    def stripe():
        fact.zero_(); nu.zero_(); sf.zero_()
        check(hx.lib.cid_search_count_stripe_dev(ctx.h, hx.h, vp(kk.data_ptr()), None, K, 1024, vp(hits.data_ptr()), vp(fact.data_ptr())))
        check(hx.lib.cid_search_unique_finalize_dev(ctx.h, vp(fact.data_ptr()), vp(ff.data_ptr()), K, 4096, vp(nu.data_ptr()), vp(sf.data_ptr()), vp(uc.data_ptr())))
    ms = timed(stripe, steps=5)
    want = None
    S = 200_000
    hk = kk[:S].cpu().numpy()
    # oracle on a sample: rows of the 64 GiB matrix are fetched row by row through the library (no 64 GiB host copy)
    ridx = np.array([[orc.xxh3(hk[j].tobytes(), s) % m for s in range(n)] for j in range(2000)], np.uint64)
    rows = hx.get_rows(ridx.reshape(-1)).reshape(2000, n, -1)
    a = rows[:, 0].copy()
    for s in range(1, n):
        a &= rows[:, s]
    exp_hits = np.unpackbits(a.view(np.uint8), axis=1, bitorder="little")[:, :C].sum(axis=0)
    g2 = hx.search_count(hk[:2000], None, want_unique=False, want_unique_colour=False)[0]
    out.append({"config": "E one rank: m=2^30 n=3 k=31, 512-colour stripe (64 GiB), striped search + unique finalize", "kmers": K, "gpu_ms": ms,
                "gpu_kmers_per_s": K / ms * 1e3, "alg_GBs": K * (n * 64 + k + 12) / ms / 1e6, "index_GiB": m * rs * 8 / 2**30,
                "bit_exact": bool(np.array_equal(exp_hits.astype(np.uint64), g2)), "sample": "2000 k-mers re-derived on the host from rows read back through cid_index_get_rows"})
    hx.close()

for o in out:
    print(json.dumps(o))
