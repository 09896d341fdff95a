"""Shared synthetic-input helpers for the tests (deterministic; SplitMix64-free numpy Generator seeds)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)


def random_kmers(rng, n, k, alphabet=ACGT):
    return alphabet[rng.integers(0, len(alphabet), size=(n, k))].astype(np.uint8)


def random_index(orc, rng, m, n_hash, k, n_colors, density=0.3, zero_row_frac=0.2):
    """Oracle index with Bernoulli(density) bits, a fraction of rows forced absent (all-zero)."""
    ix = orc.Index(m, n_hash, k, n_colors)
    rows = ix.rows()
    w32 = ix.w32
    bits = rng.random((m, n_colors)) < density
    bits[rng.random(m) < zero_row_frac] = False
    packed = np.zeros((m, w32 * 32), bool)
    packed[:, :n_colors] = bits
    # BitVec<u32>: bit c -> word c//32, bit c%32 (LSB first)
    weights = (1 << np.arange(32, dtype=np.uint64)).astype(np.uint64)
    rows[:] = (packed.reshape(m, w32, 32).astype(np.uint64) * weights).sum(axis=2).astype(np.uint32)
    for c in range(n_colors):
        ix.set_color(c, f"acc_{c:05d}", 1000 + c)
    return ix


def plant(ix, rng, kmers, frac=0.5, max_colours=3):
    """Bloom-insert a fraction of the query k-mers into random colours (so AND results are non-trivial)."""
    n = len(kmers)
    for j in np.flatnonzero(rng.random(n) < frac):
        for c in rng.choice(ix.n_colors, size=rng.integers(1, max_colours + 1), replace=False):
            ix.insert(int(c), kmers[j].tobytes())


def to_hip_index(ctx, oix):
    import colorid_amd
    hx = colorid_amd.Index(ctx, oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.put_dense(oix.rows())
    return hx.finalize()


# ---------------------------------------------------------------------------------------------- fastq fixtures

def write_fastq_gz(path, records, multi_member=False):
    """records: list of (id, seq, qual) byte strings.  multi_member=True writes two concatenated gzip members
    (the reference reads with MultiGzDecoder)."""
    import gzip
    lines = [b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n" for i, s, q in records]
    if multi_member and len(lines) > 1:
        h = len(lines) // 2
        with open(path, "wb") as f:
            f.write(gzip.compress(b"".join(lines[:h])))
            f.write(gzip.compress(b"".join(lines[h:])))
    else:
        with gzip.open(path, "wb") as f:
            f.write(b"".join(lines))


def synth_fastq_records(rng, genomes, n, read_len, mate=0, err=0.01, lowq=0.05, n_rate=0.003, lower_rate=0.03):
    """Reads drawn from `genomes` (list of bytes) with substitutions, some low-quality bases (phred < 15),
    some N and a few lower-case / short reads.  Deterministic in rng; mate=1 gives the reverse mates."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    recs = []
    for i in range(n):
        g = genomes[int(rng.integers(len(genomes)))]
        L = int(read_len if rng.random() < 0.9 else rng.integers(10, read_len))
        frag = int(rng.integers(L, 2 * L + 20))
        st = int(rng.integers(0, max(1, len(g) - frag)))
        s = g[st:st + L] if mate == 0 else g[st + frag - L:st + frag].translate(comp)[::-1]
        a = np.frombuffer(s, np.uint8).copy()
        e = rng.random(len(a)) < err
        a[e] = ACGT[rng.integers(0, 4, int(e.sum()))]
        a[rng.random(len(a)) < n_rate] = ord("N")
        if rng.random() < lower_rate:
            a = np.frombuffer(a.tobytes().lower(), np.uint8).copy()
        q = np.full(len(a), ord("I"), np.uint8)
        lq = rng.random(len(a)) < lowq
        q[lq] = rng.integers(33, 48, int(lq.sum()))       # phred 0..14 -> masked at -Q 15
        recs.append((f"read{i}/{mate + 1} extra words".encode(), a.tobytes(), q.tobytes()))
    return recs
