#!/usr/bin/env python3
"""gpurun_out/<tag>_readid/ (tools/profile_readid.sh) -> profiles/<tag>_readid_{kernel_stats.csv,pmc.csv,summary.md}"""
import collections, csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"{tag}_readid"); dst = os.path.join(ROOT, "profiles")
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_readid_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in rows[:10]:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
ks = next(r for r in rows if "k_readid<" in r["Name"]); avg_ns = float(ks["AverageNs"])
mean, allrows, hdr = {}, [], None
acc = collections.defaultdict(list)
for name in ("pmc_rdreq", "pmc_write", "pmc_sq", "pmc_sq2"):
    p = os.path.join(src, name + ".csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        hdr = hdr or list(r.keys()); allrows.append(r); acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_readid_pmc.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=hdr); w.writeheader(); w.writerows(allrows)
mean = {k: sum(v) / len(v) for k, v in acc.items()}
b = json.loads([l for l in open(os.path.join(src, "bench_stats.log")) if l.startswith("{")][-1])
rd = 128 * mean.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * mean.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * mean.get("TCC_EA0_RDREQ_32B_sum", 0)
wr = 1024 * mean.get("WRITE_SIZE", 0)
alg = b["alg_GBs"] * 1e9 * b["ms"] * 1e-3
md = f"""# {tag}: k_readid on MI355X — rocprofv3 evidence

`tools/profile_readid.sh`: `rocprofv3 --kernel-trace --stats -- python3 tools/bench_readid.py --check 0`, PMC in separate passes.
Workload: BASELINE.json configs[2] shape — m = {b['config']['m']:,}, n = {b['config']['n']}, k = {b['config']['k']}, C = {b['config']['C']},
{b['reads']:,} synthetic 150-bp {'read PAIRS' if b.get('paired') else 'single-end reads'} resident in HBM, `read_id -d {b['d']} -B {b['B']}`.

| quantity | value |
|---|---|
| kernel average (rocprofv3 --stats, {ks['Calls']} calls) | {avg_ns/1e6:.3f} ms = {b['reads']/avg_ns*1e3:.1f} M reads/s, {b['distinct_kmers_per_s']/1e9:.1f} G distinct k-mers/s (HIP events: {b['ms']:.3f} ms) |
| row gathers | {b['row_gathers_per_s']/1e9:.1f} G/s (the chip's random 128-byte line limit is ~54 G/s) |
| algorithmic bytes per launch (rows n x 32 B per distinct k-mer + bases in + report rows out) | {alg/1e9:.2f} GB -> {alg/avg_ns:.0f} GB/s = {alg/avg_ns/8000:.2f} of 8 TB/s |
| HBM reads (L2 request sizes: {mean.get('TCC_EA0_RDREQ_128B_sum',0)/1e6:.1f} M x 128 B) | {rd/1e9:.2f} GB |
| HBM writes (WRITE_SIZE x 1024) | {wr/1e9:.2f} GB |
| **HBM traffic** | **{(rd+wr)/1e9:.2f} GB = {(rd+wr)/avg_ns:.0f} GB/s = {(rd+wr)/avg_ns/8000:.2f} of peak** |
| SQ_WAIT_ANY / SQ_WAVE_CYCLES | {mean.get('SQ_WAIT_ANY',0)/max(1,mean.get('SQ_WAVE_CYCLES',1)):.2f} |
| SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | {mean.get('SQ_ACTIVE_INST_ANY',0)/max(1,mean.get('SQ_WAVE_CYCLES',1)):.2f} |
| VALU / SALU / LDS wave-instructions per read | {mean.get('SQ_INSTS_VALU',0)/b['reads']:.0f} / {mean.get('SQ_INSTS_SALU',0)/b['reads']:.0f} / {mean.get('SQ_INSTS_LDS',0)/b['reads']:.0f} |
| SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | {mean.get('SQ_LDS_BANK_CONFLICT',0)/max(1,mean.get('SQ_LDS_IDX_ACTIVE',1)):.2f} |

Reading: as in k_search_count every 32-byte row costs a 128-byte line; this kernel additionally spends ~{mean.get('SQ_INSTS_VALU',0)/b['reads']:.0f} VALU
wave-instructions per read on window extraction, the per-read hash-table set and hashing, so it sits at {b['row_gathers_per_s']/54e9:.0%} of the
line-rate limit rather than at it.
"""
open(os.path.join(dst, f"{tag}_readid_summary.md"), "w").write(md)
# the per-workload traffic summary bench.py's `readid` record looks up (units per launch = reads)
import importlib.util
spec = importlib.util.spec_from_file_location("pmc_store", os.path.join(ROOT, "tools", "pmc_store.py")); store = importlib.util.module_from_spec(spec); spec.loader.exec_module(store)
c = b["config"]
pj = store.write("k_readid_pe" if b.get("paired") else "k_readid_se", c["C"], c["m"], c["n"], c["k"], b["reads"], int(alg / b["reads"]), mean, avg_ns, tag,
                 [f"profiles/{tag}_readid_pmc.csv", f"profiles/{tag}_readid_kernel_stats.csv"], ks["Name"][:120], match="k_readid<")
print(os.path.relpath(pj, ROOT))
print(md)
