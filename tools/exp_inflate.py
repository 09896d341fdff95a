#!/usr/bin/env python3
"""cid_bgzf_inflate on the members of 1 M synthetic 150-bp reads (310 MB of FASTQ text, ~4750 members): wall time of the call (H2D of
the compressed bytes, the kernel, D2H of the text) per batch size, against zlib on one host thread.  Kernel time: run under
rocprofv3 --kernel-trace --stats."""
import ctypes as C, json, os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import colorid_amd

R = int(os.environ.get("EXP_READS", 1_000_000))
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
reads = acgt[rng.integers(0, 4, (R, 150))]
qual = np.frombuffer(b"FFFFFFFF:,#", np.uint8)[rng.integers(0, 11, (R, 150))]
blob = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(R))
level = int(os.environ.get("EXP_LEVEL", 6))
members, lens = [], []
t0 = time.time()
for i in range(0, len(blob), 65280):
    c = blob[i:i + 65280]
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(c) + co.flush()
    members.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, (12 + 6 + len(body) + 8 - 1) & 0xFFFF) + body +
                   struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
    lens.append(len(c))
ctx = colorid_amd.Context(0)
lib = ctx.lib
res = {"reads": R, "text_MB": len(blob) / 1e6, "compressed_MB": sum(map(len, members)) / 1e6, "members": len(members), "level": level}
t0 = time.perf_counter()
for m in members:
    zlib.decompress(m, 31)
res["zlib_one_thread_ms"] = (time.perf_counter() - t0) * 1e3
for per in (256, 1024, len(members)):
    best = None
    for rep in range(3):
        out_all = []
        t = 0.0
        for a in range(0, len(members), per):
            ms = members[a:a + per]; tl = np.array(lens[a:a + per], np.uint32)
            comp = np.frombuffer(b"".join(ms) + b"\0", np.uint8)
            off = np.cumsum([0] + [len(x) for x in ms[:-1]]).astype(np.uint32); ln = np.array([len(x) for x in ms], np.uint32)
            to = np.cumsum(np.concatenate([[0], tl[:-1]])).astype(np.uint32)
            out = np.empty(int(tl.sum()) + 1, np.uint8)
            bad = C.c_size_t(0)
            t0 = time.perf_counter()
            rc = lib.cid_bgzf_inflate(ctx.h, comp.ctypes.data, len(comp) - 1, off.ctypes.data, ln.ctypes.data, to.ctypes.data, tl.ctypes.data, len(ms),
                                      out.ctypes.data, int(tl.sum()), C.byref(bad))
            t += time.perf_counter() - t0
            assert rc == 0, lib.cid_last_error()
            if rep == 0:
                out_all.append(out[:-1].tobytes())
        if rep == 0:
            assert b"".join(out_all) == blob
        best = t if best is None else min(best, t)
    res[f"gpu_call_ms_batches_of_{per}"] = best * 1e3
print(json.dumps(res))
