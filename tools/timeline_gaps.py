#!/usr/bin/env python3
"""GPU timeline of one `colorid read_id` run through the device front end, from rocprofv3's kernel trace (csv): how much of the loop the
GPU is busy, per kernel family, per stream, and where it idles.  Usage: timeline_gaps.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    fam = ("inflate" if "k_bgzf_inflate" in name else "readid" if "k_readid<" in name else "fq" if "k_fq_" in name else
           "rows" if "k_row_" in name else "rocprim" if "rocprim" in name else "copy/fill" if "rocclr" in name else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam, r.get("Queue_Id", "?"), name))
ev.sort()
first_readid = min(e[0] for e in ev if e[2] == "readid")
last_readid = max(e[1] for e in ev if e[2] == "readid")
loop = [e for e in ev if e[1] >= first_readid - 30_000_000 and e[0] <= last_readid]
t0, t1 = min(e[0] for e in loop), max(e[1] for e in loop)
print(f"loop span {1e-6 * (t1 - t0):.1f} ms, {len(loop)} kernels")


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot


print(f"GPU busy (any kernel) {1e-6 * union([(e[0], e[1]) for e in loop]):.1f} ms")
fam = defaultdict(list)
for e in loop: fam[e[2]].append((e[0], e[1]))
for f, iv in sorted(fam.items(), key=lambda kv: -union(kv[1])):
    print(f"  {f:10s} n={len(iv):5d} busy {1e-6 * union(iv):8.1f} ms  sum {1e-6 * sum(b - a for a, b in iv):8.1f} ms")
not_inflate = [(e[0], e[1]) for e in loop if e[2] != "inflate"]
print(f"busy without the inflate kernels {1e-6 * union(not_inflate):.1f} ms")
q = defaultdict(list)
for e in loop: q[e[3]].append((e[0], e[1]))
for k, iv in q.items(): print(f"  queue {k}: n={len(iv)} busy {1e-6 * union(iv):.1f} ms")
# per readid launch: the gap since the previous readid ended and what ran in it
rd = [e for e in loop if e[2] == "readid"]
print("per stretch: readid ms | gap before it ms | non-inflate kernel busy inside the gap ms")
prev_end = None
for e in rd:
    if prev_end is not None:
        inside = [(max(a, prev_end), min(b, e[0])) for a, b, f, *_ in loop if f != "inflate" and b > prev_end and a < e[0]]
        inside = [(a, b) for a, b in inside if b > a]
        print(f"  {1e-6 * (e[1] - e[0]):6.2f} | {1e-6 * (e[0] - prev_end):6.2f} | {1e-6 * union(inside):6.2f}")
    prev_end = e[1]

# one gap in detail: every kernel between the end of readid #5 and the end of readid #6, times relative to the former's end
if len(rd) > 6:
    a, b = rd[5][1], rd[6][1]
    print("kernels from the end of readid #5 to the end of readid #6 (start ms, duration ms, queue, name):")
    for e in loop:
        if e[1] > a and e[0] < b:
            print(f"  {1e-6 * (e[0] - a):8.3f} {1e-6 * (e[1] - e[0]):8.3f}  q{e[3]}  {e[4][:90]}")
