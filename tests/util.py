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
