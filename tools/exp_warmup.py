#!/usr/bin/env python3
"""What cid_warmup loads, piece by piece, in a fresh process (code objects are inflated and registered on first use of a kernel of their
unit): usage exp_warmup.py <flags...> — every flag (1 read_id, 2 search, 4 inflate, 8 fastq) is warmed in the order given and timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import colorid_amd
from colorid_amd._lib import check
t1 = time.perf_counter()
ctx = colorid_amd.Context(0)
t2 = time.perf_counter()
out = [f"import {1e3 * (t1 - t0):.0f} ms", f"ctx {1e3 * (t2 - t1):.0f} ms"]
for f in sys.argv[1:]:
    t = time.perf_counter()
    check(ctx.lib.cid_warmup(ctx.h, int(f)))
    out.append(f"warm({f}) {1e3 * (time.perf_counter() - t):.0f} ms")
print(", ".join(out))
os._exit(0)
