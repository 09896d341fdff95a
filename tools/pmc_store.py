#!/usr/bin/env python3
"""One JSON file per (kernel, workload) under profiles/pmc/: the HBM traffic per launch that bench.py quotes beside its live kernel
time.  PMC counters cannot be collected inside the timed run (separate `rocprofv3 --pmc` passes, MI355X_MICROARCH.md §HBM), so the
bench line cites a committed file — and round 4 lost the headline's figure because every pass wrote the SAME file name: the C = 1024
pass overwrote the C = 256 one.  The file name now IS the workload; bench.py (`pmc_path`, the consumer, owns the naming) looks its own
workload up and finds either that workload's counters or nothing.

  python3 tools/pmc_store.py <counter rows .csv> <kernel_stats .csv> --kernel k_search_count --colours 256 --bloom 50000000 \
          --hashes 4 --k 31 --kmers 120000000 --alg-bytes-per-kmer 167 --tag r05

reads the per-dispatch counter rows and the `--stats` summary as they are committed under profiles/ (so every JSON can be re-derived
from committed evidence) and writes profiles/pmc/<kernel>_C<colours>_m<bloom>_n<hashes>_k<k>.json.
"""
import argparse
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_path(kernel, C, m, n, k):
    """The same rule as bench.py's pmc_path (asserted equal by tests/test_profiles_cpu.py): tools must not import torch for this."""
    return os.path.join("profiles", "pmc", f"{kernel}_C{C}_m{m}_n{n}_k{k}.json")


def counters_mean(csv_path, kernel_substr):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(csv_path)):
        if kernel_substr in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {c: sum(v) / len(v) for c, v in acc.items()}


def traffic_of(mean):
    """HBM bytes per launch: the L2's memory-side read requests by size (FETCH_SIZE counts a 128-byte request as 64 B on gfx950 and is
    kept, doubled, as the cross-check only) + WRITE_SIZE (KiB, exact)."""
    rd = 128 * mean.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * mean.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * mean.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr = 1024 * mean.get("WRITE_SIZE", 0)
    return rd, wr, 2 * 1024 * mean.get("FETCH_SIZE", 0)


def write(kernel, C, m, n, k, K, alg_bytes_per_kmer, mean, avg_ns, tag, sources, kernel_name=None, match=None):
    rd, wr, rd_fetch = traffic_of(mean)
    out = {"kernel": kernel, "kernel_name": kernel_name, "tag": tag, "kmers_per_launch": K, "n_colors": C, "bloom_size": m, "num_hash": n,
           "k_size": k, "traffic_bytes": rd + wr, "read_bytes_rdreq": rd, "read_bytes_fetch_size_x2": rd_fetch, "write_bytes": wr,
           "algorithmic_bytes": alg_bytes_per_kmer * K, "alg_bytes_per_kmer": alg_bytes_per_kmer, "rocprof_avg_kernel_ns": avg_ns,
           "sources": sources, "counters_mean_per_launch": mean}
    if match:
        out["match"] = match   # the substring of the kernel's name that selects its rows in sources[0] (default: k_search_count)
    path = os.path.join(ROOT, pmc_path(kernel, C, m, n, k))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("counters_csv")
    ap.add_argument("kernel_stats_csv")
    ap.add_argument("--kernel", default="k_search_count", help="file-name key (k_search_count, k_search_count_stripe, ...)")
    ap.add_argument("--match", default="k_search_count", help="substring of the kernel's name in the csv rows")
    ap.add_argument("--colours", type=int, required=True)
    ap.add_argument("--bloom", type=int, required=True)
    ap.add_argument("--hashes", type=int, required=True)
    ap.add_argument("--k", type=int, required=True)
    ap.add_argument("--kmers", type=int, required=True)
    ap.add_argument("--alg-bytes-per-kmer", type=int, required=True)
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    mean = counters_mean(os.path.join(ROOT, a.counters_csv), a.match)
    ks = next(r for r in csv.DictReader(open(os.path.join(ROOT, a.kernel_stats_csv))) if a.match in r["Name"])
    p = write(a.kernel, a.colours, a.bloom, a.hashes, a.k, a.kmers, a.alg_bytes_per_kmer, mean, float(ks["AverageNs"]), a.tag,
              [a.counters_csv, a.kernel_stats_csv], ks["Name"])
    j = json.load(open(p))
    print(f"{os.path.relpath(p, ROOT)}: traffic {j['traffic_bytes'] / 1e9:.2f} GB = {j['traffic_bytes'] / j['algorithmic_bytes']:.2f}x algorithmic, "
          f"kernel {j['rocprof_avg_kernel_ns'] / 1e6:.3f} ms")


if __name__ == "__main__":
    sys.exit(main())
