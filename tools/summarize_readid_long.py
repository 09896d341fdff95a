#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/profile_readid_long.sh) -> profiles/<tag>_summary.md, _kernel_stats.csv, _pmc.csv, _bench.json: the long-read
read_id path (cid_readlong.hip + k_readid_slices) on 150 Mbases of 10 kb reads and of a 2 kb / 10 kb / 100 kb mix."""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_readid_long"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
bench = json.loads([l for l in open(f"{src}/bench.json") if l.startswith("{")][-1])["readid_long"]
json.dump(bench, open(f"{dst}/{tag}_bench.json", "w"), indent=1)
rows = [r for r in csv.DictReader(open(f"{src}/kernel_stats.csv")) if "cid::" in r["Name"]]
with open(f"{dst}/{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows:
        r = dict(r); r["Name"] = r["Name"][:100]; w.writerow(r)
acc = collections.defaultdict(list)
with open(f"{dst}/{tag}_pmc.csv", "w") as out:
    first = True
    for fn in ("pmc_rdreq", "pmc_write"):
        p = f"{src}/{fn}.csv"
        if not os.path.exists(p):
            continue
        lines = open(p).read().splitlines()
        out.write("\n".join(lines if first else lines[1:]) + "\n"); first = False
        for r in csv.DictReader(open(p)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
ks = next(r for r in rows if "k_readid_slices" in r["Name"])
t10, mix = bench["reads_10kb"], bench["mix_2k_10k_100k"]
n = bench["config"]["num_hash"]
# the bench makes 7 calls per shape, 10 kb first: the first 7 dispatches of the plain k_readid_slices are the 10 kb ones (the BYTES
# instantiation — the soft-masked reads' — follows every one of them and finds nothing to do on these reads)
acc10 = collections.defaultdict(list)
with open(f"{dst}/{tag}_pmc_10kb.csv", "w") as out10:
    first = True
    for fn in ("pmc_rdreq", "pmc_write"):
        p = f"{src}/{fn}.csv"
        if not os.path.exists(p):
            continue
        rows_p = list(csv.DictReader(open(p)))
        plain = [r for r in rows_p if "k_readid_slices" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"].split("(")[0]]
        ids = sorted({int(r["Dispatch_Id"]) for r in plain})[:7]
        keep = [r for r in plain if int(r["Dispatch_Id"]) in ids]
        w10 = csv.DictWriter(out10, fieldnames=rows_p[0].keys())
        if first:
            w10.writeheader(); first = False
        w10.writerows(keep)
        for r in keep:
            acc10[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean10 = {c: sum(v) / len(v) for c, v in acc10.items()}
lines10 = mean10.get("TCC_EA0_RDREQ_128B_sum", 0)
wr10 = mean10.get("WRITE_SIZE", 0) * 1024
trace = {r["Name"][:60]: (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6) for r in rows}
ms10 = float(ks["MaxNs"]) / 1e6   # the 10 kb shape holds more long-path k-mers than the mix: its dispatches are the slow ones
md = f"""# {tag}: the long-read read_id path on MI355X — rocprofv3 evidence

`tools/profile_readid_long.sh {tag}`: `bench.py --only readid_long` plain, under `rocprofv3 --kernel-trace --stats`, and in separate `--pmc` passes of
`k_readid_slices`.  Workload: configs[2]'s index (m = 30 M, n = 2, k = 21, 256 colours, 32-byte rows), 150 Mbases resident in HBM through
`cid_readid_count_resident`, `-d 1 -B 3`; wall time of one call (host side included), mean of 5.

| quantity | 10 kb reads | 2 kb / 10 kb / 100 kb mix |
|---|---|---|
| reads | {t10['reads']:,} | {mix['reads']:,} |
| ms per call | {t10['ms']:.2f} | {mix['ms']:.2f} |
| distinct k-mers | {t10['distinct_kmers']:,} | {mix['distinct_kmers']:,} |
| row gathers/s (n rows per distinct k-mer, whole call) | {t10['row_gathers_per_s']/1e9:.1f} G | {mix['row_gathers_per_s']/1e9:.1f} G |
| algorithmic fraction of 8 TB/s (whole call) | {t10['frac']:.3f} | {mix['frac']:.3f} |
| rows of a sample of reads == the oracle's | {t10.get('bit_exact')} | {mix.get('bit_exact')} |
| the oracle's loop on {t10.get('cpu_baseline', {}).get('cores', '?')} host threads | {t10.get('cpu_baseline', {}).get('value', 0)/1e6:.1f} M bases/s | {mix.get('cpu_baseline', {}).get('value', 0)/1e6:.1f} M bases/s |

Kernels (`{tag}_kernel_stats.csv`; both shapes in one run, 7 calls each; min / max = the two shapes):

| kernel | calls | average ms | min | max |
|---|---|---|---|---|
""" + "\n".join(f"| `{k}` | {v[0]} | {v[1]:.3f} | {v[2]:.3f} | {v[3]:.3f} |" for k, v in trace.items()) + f"""

`k_readid_slices` on the 10 kb shape (`{tag}_pmc.csv`): {lines10/1e6:.1f} M fabric read requests per launch, all of 128 bytes, for
{t10['distinct_kmers']*n/1e6:.1f} M row gathers (+ the code array: 8 B per window) = {lines10*128/1e9:.1f} GB fetched, {wr10/1e6:.0f} MB written, in {ms10:.2f} ms =
**{lines10/ms10/1e6:.1f} G lines/s = {(lines10*128+wr10)/ms10/1e6/8000:.2f} of the HBM peak in fetched bytes** — the rate of a bare gather (54 G lines/s,
`tools/gather_probe`).  The path's own passes (window codes, first-occurrence tables, scan) take the rest of the call.
"""
open(f"{dst}/{tag}_summary.md", "w").write(md)
import importlib.util
spec = importlib.util.spec_from_file_location("pmc_store", os.path.join(ROOT, "tools", "pmc_store.py")); store = importlib.util.module_from_spec(spec); spec.loader.exec_module(store)
c = bench["config"]
pj = store.write("k_readid_slices", c["n_colors"], c["bloom_size"], c["num_hash"], c["k_size"], t10["reads"], int(t10["alg_bytes"] / t10["reads"]), mean10,
                 ms10 * 1e6, tag, [f"profiles/{tag}_pmc_10kb.csv", f"profiles/{tag}_kernel_stats.csv"], ks["Name"][:100], match="k_readid_slices")
print(os.path.relpath(pj, ROOT))
print(md)
