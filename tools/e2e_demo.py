#!/usr/bin/env python3
"""End-to-end wall times of the C++ CLI on one MI355X box (synthetic data written to /tmp): build -> search -g -> read_id ->
search -s, with the phases the reference itself reports on stderr.  Not a bench line; results go to DESIGN.md."""
import gzip, json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "colorid_amd", "bin", "colorid")
W = "/tmp/cid_e2e"; os.makedirs(W, exist_ok=True)
G, LG, R = int(os.environ.get("E2E_GENOMES", 64)), 1_000_000, int(os.environ.get("E2E_READS", 1_000_000))
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
genomes = []
with open(f"{W}/refs.tsv", "w") as tsv:
    for g in range(G):
        s = acgt[rng.integers(0, 4, LG)]
        genomes.append(s)
        with open(f"{W}/g{g:03d}.fasta", "wb") as f:
            f.write(f">genome{g}\n".encode())
            body = s.tobytes()
            f.write(b"\n".join(body[i:i + 80] for i in range(0, LG, 80)) + b"\n")
        tsv.write(f"genome{g:03d}\t{W}/g{g:03d}.fasta\n")
t = time.time()
src = rng.integers(0, G, R); pos = rng.integers(0, LG - 150, R)
reads = np.stack([genomes[src[i]][pos[i]:pos[i] + 150] for i in range(R)])
err = rng.random(reads.shape) < 0.01
reads[err] = acgt[rng.integers(0, 4, int(err.sum()))]
lines = []
qual = b"I" * 150
blob = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(R))
with gzip.open(f"{W}/reads.fastq.gz", "wb", compresslevel=1) as f:
    f.write(blob)
gen_s = time.time() - t
res = {"genomes": G, "genome_len": LG, "reads": R, "fastq_gz_MB": os.path.getsize(f"{W}/reads.fastq.gz") / 1e6}


def run(*args):
    t = time.time()
    p = subprocess.run([BIN, *args], capture_output=True, text=True, env=dict(os.environ, COLORID_TIMING="1"))
    dt = time.time() - t
    if p.returncode != 0:
        print(p.stderr[-2000:]); sys.exit(1)
    return dt, p.stdout, p.stderr


dt, out, err = run("build", "-s", "50000000", "-n", "4", "-k", "31", "-b", f"{W}/idx", "-r", f"{W}/refs.tsv")
res["build_s"] = dt; res["bxi_GB"] = os.path.getsize(f"{W}/idx.bxi") / 1e9
res["build_stderr"] = [l for l in err.splitlines() if "timing:" in l]
dt, out, err = run("info", "-b", f"{W}/idx.bxi")
res["info_s"] = dt
dt, out, err = run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-g", "-f", "0", "-p", "0.005")
res["search_g_total_s"] = dt; res["search_g_rows"] = len(out.strip().splitlines()) - 1
res["search_stderr"] = [l for l in err.splitlines() if "Index loaded" in l or "k-mers in query" in l or "Search:" in l or "timing:" in l]
dt, out, err = run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-f", "0", "-p", "0.005")   # default report: mean / mode / unique
res["search_default_total_s"] = dt; res["search_default_rows"] = len(out.strip().splitlines()) - 1
dt, out, err = run("read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-n", f"{W}/rid")
res["read_id_total_s"] = dt
res["read_id_stderr"] = [l.split("\r")[-1] for l in err.splitlines() if "Classified" in l or "Index loaded" in l or "timing:" in l]
counts = dict(l.split("\t") for l in open(f"{W}/rid_counts.txt").read().splitlines())
res["read_id_accept_frac"] = 1.0 - int(counts.get("reject", 0)) / R
dt, out, err = run("read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", f"{W}/reads.fastq.gz", "-n", f"{W}/rid_pe")
res["read_id_pe_total_s"] = dt
res["read_id_pe_stderr"] = [l.split("\r")[-1] for l in err.splitlines() if "Classified" in l or "timing:" in l]
# the same reads as block gzip (BGZF, what bgzip / htslib / Illumina's converters write): members are inflated by 8 threads
import struct, zlib
with open(f"{W}/reads.bgzf.fastq.gz", "wb") as f:
    for i in range(0, len(blob), 65280):
        c = blob[i:i + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(c) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
        f.write(body + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
dt, out, err = run("read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.bgzf.fastq.gz", "-n", f"{W}/rid_b")
res["read_id_bgzf_total_s"] = dt
res["read_id_bgzf_stderr"] = [l.split("\r")[-1] for l in err.splitlines() if "Classified" in l or "timing:" in l]
res["read_id_bgzf_same_rows"] = open(f"{W}/rid_b_reads.txt").read() == open(f"{W}/rid_reads.txt").read()
dt, out, err = run("read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.bgzf.fastq.gz", f"{W}/reads.bgzf.fastq.gz", "-n", f"{W}/rid_pe_b")
res["read_id_pe_bgzf_total_s"] = dt
res["read_id_pe_bgzf_stderr"] = [l.split("\r")[-1] for l in err.splitlines() if "Classified" in l or "timing:" in l]
dt, out, err = run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.bgzf.fastq.gz", "-g", "-f", "0", "-p", "0.005")
res["search_g_bgzf_total_s"] = dt
if G >= 128 and os.environ.get("E2E_GROUPS", "1") != "0":
    # the multi-rank call sequences at this size (two ranks share the one GPU here, so this shows overhead and agreement, not speed-up):
    # index replicated + the query's k-mers counted over the ranks; index cut into colour stripes
    base = sorted(run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-f", "0", "-p", "0.005")[1].splitlines())
    for tag, extra in (("replicated_2ranks", ("--devices", "0,0")), ("striped_2ranks", ("--devices", "0,0", "--placement", "striped"))):
        dt, out, err = run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-f", "0", "-p", "0.005", *extra)
        res[f"search_default_{tag}_s"] = dt
        res[f"search_default_{tag}_same_rows"] = sorted(out.splitlines()) == base
        dt, out, err = run("read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/reads.fastq.gz", "-n", f"{W}/rid_{tag}", *extra)
        res[f"read_id_{tag}_s"] = dt
        res[f"read_id_{tag}_same_rows"] = open(f"{W}/rid_{tag}_reads.txt").read() == open(f"{W}/rid_reads.txt").read()
dt, out, err = run("search", "-b", f"{W}/idx.bxi", "-q", f"{W}/g007.fasta", "-s")
res["search_s_total_s"] = dt; res["search_s_out"] = out.strip().splitlines()[-1] if out.strip() else ""
print(json.dumps(res))
