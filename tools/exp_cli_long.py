#!/usr/bin/env python3
"""`colorid read_id` on LONG reads through the whole command line (round 6): 150 Mbases of 10 kb reads and of a
2 kb / 10 kb / 100 kb mix, as block-gzip FASTQ (the device front end: inflate, records, packing and classification in HBM — until
round 5 it refused such reads and the CLI re-ran the whole input on the host) and as plain FASTA (the host reader ->
cid_readid_count_sparse); the same with 1 % of the reads soft-masked (a lower-case stretch: those reads alone take the byte-string path).
Index: configs[2]'s parameters (k = 21, 30 M rows, 2 hashes) over 256 synthetic 1 Mbp genomes.  Run on the GPU box from the repo root:
    python3 tools/exp_cli_long.py > gpurun_out/r06_cli_long.txt
Prints, per case and repetition, the wall clock of the process and the CLI's own phase lines (COLORID_TIMING=1)."""
import os, re, struct, subprocess, sys, time, zlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "colorid_amd", "bin", "colorid")
W = "/tmp/cid_long"; os.makedirs(W, exist_ok=True)
G, LG, TOTAL = int(os.environ.get("LONG_GENOMES", 256)), 1_000_000, int(os.environ.get("LONG_BASES", 150_000_000))
rng = np.random.default_rng(5)
acgt = np.frombuffer(b"ACGT", np.uint8)


def log(*a):
    print(*a, flush=True)


genomes = []
with open(f"{W}/refs.tsv", "w") as tsv:
    for g in range(G):
        s = acgt[rng.integers(0, 4, LG)]
        genomes.append(s)
        with open(f"{W}/g{g:03d}.fasta", "wb") as f:
            f.write(f">genome{g}\n".encode() + s.tobytes() + b"\n")
        tsv.write(f"genome{g:03d}\t{W}/g{g:03d}.fasta\n")


def run(label, args, reps=3, env=None):
    for rep in range(reps):
        t = time.time()
        p = subprocess.run([BIN, *args], capture_output=True, text=True, env=dict(os.environ, COLORID_TIMING="1", **(env or {})))
        dt = time.time() - t
        if p.returncode != 0:
            log(label, "FAILED", p.stderr[-1500:]); sys.exit(1)
        err = p.stderr.replace("\r", "\n")
        ph = re.findall(r"timing: ([A-Za-z ]+?) (\d+) ms \(at (\d+) ms\)", err)
        notes = [l for l in err.splitlines() if "front end" in l or "starting over" in l]
        log(f"{label} rep {rep}: wall {dt:.3f} s | " + ", ".join(f"{n} {ms}" for n, ms, _ in ph) + (" | " + " ; ".join(n.strip()[:160] for n in notes) if notes else ""))
    return p


t = time.time()
p = run("build -k 21 -s 30000000 -n 2 (256 x 1 Mbp)", ["build", "-s", "30000000", "-n", "2", "-k", "21", "-b", f"{W}/idx", "-r", f"{W}/refs.tsv"], reps=1)
log(f"index: {os.path.getsize(f'{W}/idx.bxi') / 1e9:.2f} GB")


def make_reads(lengths, soft_frac):
    reads = []
    for i, L in enumerate(lengths):
        g = genomes[int(rng.integers(0, G))]
        st = int(rng.integers(0, LG - L))
        s = g[st:st + L].copy()
        errs = rng.integers(0, L, L // 50)                       # 2 % substitutions: a long-read technology's order of magnitude
        s[errs] = acgt[rng.integers(0, 4, errs.size)]
        b = s.tobytes()
        if soft_frac and rng.random() < soft_frac:
            a = int(rng.integers(0, L - 300))
            b = b[:a] + b[a:a + 300].lower() + b[a + 300:]
        reads.append(b)
    return reads


def write_bgzf(path, blob):
    with open(path, "wb") as f:
        for i in range(0, len(blob), 65280):
            c = blob[i:i + 65280]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = co.compress(c) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1))
            f.write(body + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


third = TOTAL // 3
shapes = {"10kb": np.full(TOTAL // 10_000, 10_000),
          "mix_2k_10k_100k": np.concatenate([np.full(third // 2_000, 2_000), np.full(third // 10_000, 10_000), np.full(third // 100_000, 100_000)])}
for name, lens in shapes.items():
    np.random.default_rng(7).shuffle(lens)
    for soft in (0.0, 0.01):
        tag = f"{name}{'_soft1pct' if soft else ''}"
        reads = make_reads(lens.tolist(), soft)
        nb = sum(len(r) for r in reads)
        fq = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads))
        fa = b"".join(b">r%d\n" % i + r + b"\n" for i, r in enumerate(reads))
        write_bgzf(f"{W}/{tag}.fastq.gz", fq)
        open(f"{W}/{tag}.fasta", "wb").write(fa)
        log(f"== {tag}: {len(reads)} reads, {nb} bases; block-gzip FASTQ {os.path.getsize(f'{W}/{tag}.fastq.gz') / 1e6:.0f} MB, FASTA {len(fa) / 1e6:.0f} MB")
        run(f"read_id {tag} bgzf fastq (device front end)", ["read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/{tag}.fastq.gz", "-n", f"{W}/out_{tag}_dev", "-Q", "0"])
        run(f"read_id {tag} bgzf fastq (host front end) ", ["read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/{tag}.fastq.gz", "-n", f"{W}/out_{tag}_host", "-Q", "0"],
            env={"COLORID_DEVICE_FASTQ": "0"})
        run(f"read_id {tag} plain fasta                 ", ["read_id", "-b", f"{W}/idx.bxi", "-q", f"{W}/{tag}.fasta", "-n", f"{W}/out_{tag}_fa"])
        same = open(f"{W}/out_{tag}_dev_reads.txt").read() == open(f"{W}/out_{tag}_host_reads.txt").read()
        rows_fa = [l.split("\t")[1:] for l in open(f"{W}/out_{tag}_fa_reads.txt").read().splitlines()]
        rows_fq = [l.split("\t")[1:] for l in open(f"{W}/out_{tag}_dev_reads.txt").read().splitlines()]
        counts = dict(l.split("\t") for l in open(f"{W}/out_{tag}_dev_counts.txt").read().splitlines())
        log(f"   rows: device front end == host front end: {same}; FASTA rows (without ids) == FASTQ rows: {rows_fa == rows_fq}; accepted {1.0 - int(counts.get('reject', 0)) / len(reads):.4f}")
        del fq, fa, reads
