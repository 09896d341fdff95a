#!/usr/bin/env python3
"""gpurun_out/<tag>_stripe/ (tools/profile_stripe.sh) -> profiles/<tag>_stripe_summary.md, _stripe_kernel_stats.csv, _stripe_pmc_search_count.csv,
<tag>_bench_striped_1gpu.json: the colour-striped step's kernel (k_search_count in stripe mode) with its fabric read requests."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"{tag}_stripe")
dst = os.path.join(ROOT, "profiles")
d = json.load(open(f"{src}/bench.json"))
rows = list(csv.DictReader(open(f"{src}/kernel_stats.csv")))
ks = next(r for r in rows if "k_search_count" in r["Name"])
with open(f"{dst}/{tag}_stripe_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in [ks] + [r for r in rows if "cid::" in r["Name"] and r is not ks][:8]:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
acc = collections.defaultdict(list)
with open(f"{dst}/{tag}_stripe_pmc_search_count.csv", "w") as out:
    first = True
    for fn in ("pmc_rdreq", "pmc_write"):
        lines = open(f"{src}/{fn}.csv").read().splitlines()
        out.write("\n".join(lines if first else lines[1:]) + "\n"); first = False
        for r in csv.DictReader(open(f"{src}/{fn}.csv")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
K = d["config"]["kmers"]; alg = d["roofline"]["alg_bytes_per_kmer"] * K
rd = 128 * m["TCC_EA0_RDREQ_128B_sum"] + 64 * m["TCC_EA0_RDREQ_64B_sum"] + 32 * m["TCC_EA0_RDREQ_32B_sum"]; wr = 1024 * m["WRITE_SIZE"]
avg = float(ks["AverageNs"])
json.dump(d, open(f"{dst}/{tag}_bench_striped_1gpu.json", "w"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_store  # noqa: E402
c = d["config"]
print("wrote", pmc_store.write("k_search_count_stripe", c["n_colors_total"], c["bloom_size"], c["num_hash"], c["k_size"], K,
                               d["roofline"]["alg_bytes_per_kmer"], m, avg, f"{tag}_stripe",
                               [f"profiles/{tag}_stripe_pmc_search_count.csv", f"profiles/{tag}_stripe_kernel_stats.csv"], ks["Name"]))
open(f"{dst}/{tag}_stripe_summary.md", "w").write(f"""# {tag}_stripe: k_search_count in stripe mode on MI355X — rocprofv3 evidence

`tools/profile_stripe.sh`: `rocprofv3 --kernel-trace --stats -- python3 bench.py --placement striped --no-cpu-baseline --steps 10 --warmup 2`,
PMC counters in separate `--pmc` passes (2 steps).  Workload: BASELINE configs[4]'s per-GPU share — one 512-colour stripe (64-byte rows)
of an m = 2^30, n = 3 index = 64 GiB resident, the 120,000,000 distinct canonical 31-mers of 1 M reads (`profiles/{tag}_bench_striped_1gpu.json`).

| quantity | value |
|---|---|
| kernel `k_search_count<2,false,false,2>` average (rocprofv3 --stats, {ks['Calls']} calls) | {avg/1e6:.3f} ms |
| kernel average (bench.py HIP events, plain run) | {d['kernel_ms']:.3f} ms; whole step {d['ms_per_step']:.3f} ms (+ unique finalize {d['finalize_ms']:.3f} ms) |
| algorithmic bytes per launch ({d['roofline']['alg_bytes_per_kmer']} B/k-mer: 3 rows x 64 B + 31 B k-mer + the packed fact read and written) | {alg/1e9:.2f} GB |
| achieved algorithmic bandwidth | {alg/avg:.0f} GB/s = {alg/avg/8000:.3f} of 8 TB/s |
| L2->fabric read requests per launch | {m['TCC_EA0_RDREQ_sum']/1e6:.1f} M, of which 128-B: {m['TCC_EA0_RDREQ_128B_sum']/1e6:.1f} M, 64-B: {m['TCC_EA0_RDREQ_64B_sum']/1e6:.3f} M (360 M row reads) |
| HBM read bytes (request sizes) | {rd/1e9:.2f} GB |
| HBM write bytes (WRITE_SIZE x 1024) | {wr/1e9:.2f} GB |
| **HBM traffic per launch** | **{(rd+wr)/1e9:.2f} GB = {(rd+wr)/avg:.0f} GB/s = {(rd+wr)/avg/8000:.2f} of 8 TB/s** |
| traffic / algorithmic | {(rd+wr)/alg:.2f}x |

Reading: a 64-byte row still costs a whole 128-byte line (no 64-byte requests leave the L2), so half of every fetched line is
unused: {alg/avg/8000:.2f} of the peak in algorithmic bytes is {(rd+wr)/avg/8000:.2f} in fetched bytes — between the 32-byte rows of the metric config
(0.24 / 0.80) and the 128-byte rows of configs[3] (0.76 / 0.76, `{tag}_c1024_summary.md`).
""")
print(open(f"{dst}/{tag}_stripe_summary.md").read())
