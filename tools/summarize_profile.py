#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_bench.sh) into the committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels + the top rows)
   profiles/<tag>_pmc_search_count.csv  per-dispatch counter rows of k_search_count (all PMC passes)
   profiles/pmc/k_search_count_C<colours>_m<bloom>_n<hashes>_k<k>.json   HBM traffic per launch of THIS workload (tools/pmc_store.py),
                                     corrected as MI355X_MICROARCH.md §HBM prescribes; bench.py looks its workload up by that name
   profiles/<tag>_summary.md         the numbers side by side
"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows[:12]:
        r = dict(r)
        r["Name"] = r["Name"][:120]
        w.writerow(r)
ks = next(r for r in rows if "k_search_count" in r["Name"])
avg_ns = float(ks["AverageNs"])

counters = collections.defaultdict(list)
allrows = []
hdr = None
for name in ("pmc_rdreq", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    p = os.path.join(src, name + ".csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        hdr = hdr or list(r.keys())
        allrows.append(r)
        counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc_search_count.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=hdr)
    w.writeheader()
    w.writerows(allrows)
mean = {k: sum(v) / len(v) for k, v in counters.items()}
grid = int(allrows[0]["Grid_Size"])

bench = json.loads([l for l in open(os.path.join(src, "bench_stats.log")) if l.startswith("{")][-1])
K = bench["config"]["kmers_per_gpu"]
# HBM read bytes: the L2's memory-side requests by size; FETCH_SIZE (KiB) counts each 128-B request as 64 B on
# gfx950, so it is doubled before comparing (MI355X_MICROARCH.md §HBM).  WRITE_SIZE (KiB) is exact.
rd = 128 * mean.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * mean.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * mean.get("TCC_EA0_RDREQ_32B_sum", 0)
rd_fetch = 2 * 1024 * mean.get("FETCH_SIZE", 0)
wr = 1024 * mean.get("WRITE_SIZE", 0)
traffic = rd + wr
alg = bench["roofline"]["alg_bytes_per_kmer"] * K
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_store  # noqa: E402
cfg = bench["config"]
print("wrote", pmc_store.write("k_search_count", cfg["n_colors"], cfg["bloom_size"], cfg["num_hash"], cfg["k_size"], K,
                               bench["roofline"]["alg_bytes_per_kmer"], mean, avg_ns, tag,
                               [f"profiles/{tag}_pmc_search_count.csv", f"profiles/{tag}_kernel_stats.csv"], ks["Name"]))

row_bytes = bench["config"]["row_bytes"]
n_hash = bench["config"]["num_hash"]
wait = mean.get('SQ_WAIT_ANY', 0) / max(1, mean.get('SQ_WAVE_CYCLES', 1))
if row_bytes < 128:
    reading = (f"Reading: every random {row_bytes}-byte row costs one 128-byte L2 line fill ({mean.get('TCC_EA0_RDREQ_128B_sum',0)/1e6:.0f} M requests for "
               f"{n_hash*K/1e6:.0f} M row reads + the streamed k-mer bytes), so the kernel moves {traffic/alg:.1f}x its algorithmic bytes and sits at "
               f"{traffic/avg_ns/8000:.0%} of the HBM peak in REAL traffic while waves wait on memory {wait:.0%} of their cycles.\n"
               "tools/gather_probe (a bare 2-lanes-per-row gather with no hashing or counting) runs the same 480 M row reads in\n"
               "8.85 ms; load flavours nt / sc1 / sc0 sc1 and fine-grained / uncached allocations all fetch 128-byte lines at the same\n"
               "rate; only scalar loads (s_load_dwordx8) fetch 64-byte lines, at <= 24 G rows/s (gpurun logs summarised in DESIGN.md).")
else:
    reading = (f"Reading: rows of {row_bytes} bytes fill their 128-byte lines ({mean.get('TCC_EA0_RDREQ_128B_sum',0)/1e6:.0f} M line requests for "
               f"{n_hash*K/1e6:.0f} M row reads of {row_bytes//128} line(s) each + the streamed k-mer bytes): traffic = {traffic/alg:.2f}x the algorithmic bytes, "
               f"nothing is fetched in vain, and the kernel runs at {traffic/avg_ns/8000:.0%} of the HBM peak = "
               f"{mean.get('TCC_EA0_RDREQ_128B_sum',0)/avg_ns:.1f} G random lines/s (the bare gather of tools/gather_probe sustains 54 G lines/s) "
               f"while waves wait on memory {wait:.0%} of their cycles.")

md = f"""# {tag}: k_search_count on MI355X — rocprofv3 evidence

Command (tools/profile_bench.sh): `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3`,
PMC counters in separate `--pmc` passes of the same command (3 steps).  Workload: {bench['config']['workload']}.

| quantity | value |
|---|---|
| k-mers per launch | {K:,} |
| kernel average (rocprofv3 --stats, {ks['Calls']} calls) | {avg_ns/1e6:.3f} ms |
| kernel average (bench.py HIP events, same run) | {bench['roofline']['kernel_ms']:.3f} ms |
| algorithmic bytes per launch ({bench['roofline']['alg_bytes_per_kmer']} B/k-mer) | {alg/1e9:.2f} GB |
| achieved algorithmic bandwidth | {alg/avg_ns:.0f} GB/s = {alg/avg_ns/8000:.3f} of 8 TB/s |
| L2->fabric read requests per launch | {mean.get('TCC_EA0_RDREQ_sum',0)/1e6:.1f} M, of which 128-B: {mean.get('TCC_EA0_RDREQ_128B_sum',0)/1e6:.1f} M, 64-B: {mean.get('TCC_EA0_RDREQ_64B_sum',0)/1e6:.3f} M, 32-B: {mean.get('TCC_EA0_RDREQ_32B_sum',0)/1e6:.3f} M |
| HBM read bytes (request sizes) | {rd/1e9:.2f} GB |
| HBM read bytes (FETCH_SIZE x 1024 x 2, gfx950 correction) | {rd_fetch/1e9:.2f} GB |
| HBM write bytes (WRITE_SIZE x 1024) | {wr/1e9:.2f} GB |
| **HBM traffic per launch** | **{traffic/1e9:.2f} GB = {traffic/avg_ns:.0f} GB/s = {traffic/avg_ns/8000:.2f} of 8 TB/s** |
| traffic / algorithmic | {traffic/alg:.2f}x |
| L2 hit rate TCC_HIT/(HIT+MISS) | {mean.get('TCC_HIT_sum',0)/max(1,mean.get('TCC_HIT_sum',0)+mean.get('TCC_MISS_sum',0)):.4f} |
| SQ_WAIT_ANY / SQ_WAVE_CYCLES | {mean.get('SQ_WAIT_ANY',0)/max(1,mean.get('SQ_WAVE_CYCLES',1)):.2f} |
| SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | {mean.get('SQ_ACTIVE_INST_ANY',0)/max(1,mean.get('SQ_WAVE_CYCLES',1)):.3f} |
| SQ_INSTS_VALU per k-mer | {mean.get('SQ_INSTS_VALU',0)*64/K if K else 0:.1f} lane-instr (wave instr x 64 / k-mers) |
| grid (threads) | {grid} |

{reading}
"""
open(os.path.join(dst, f"{tag}_summary.md"), "w").write(md)
print(md)
