"""Minimizer (.mxi) indices — BigsyMapMiniNew (src/bigsi.rs:40-49): Bloom keys are find_minimizer(kmer, m) (kmer.rs:971-986).
Device Bloom inserts (build.rs:455-459) and read_id counts over per-read minimizer sets (kmer.rs:363-394) against the oracle."""
import numpy as np
import pytest

from test_gpu_readid import check, sample_reads
from util import random_kmers, to_hip_index

pytestmark = pytest.mark.gpu


def mini_index(orc, rng, m, n_hash, k, msz, n_colors, genomes):
    oix = orc.Index(m, n_hash, k, n_colors)
    oix.set_minimizer(msz)
    for c in range(n_colors):
        oix.set_color(c, f"acc{c:04d}", 3000)
    for gi, g in enumerate(genomes):
        km = orc.Kmers(k)
        km.kmerize_vector(g, 1)
        for key in km.keys():
            oix.insert(gi, key.tobytes())
            oix.insert(n_colors - 1 - gi, key.tobytes())
    return oix


def to_hip_mini(ctx, oix):
    import colorid_amd
    hx = colorid_amd.Index(ctx, oix.m, oix.n_hash, oix.k, oix.n_colors)
    hx.set_minimizer(oix.m_size)
    hx.put_dense(oix.rows())
    return hx.finalize()


@pytest.mark.parametrize("n_colors,n_hash,k,msz,m", [(4, 4, 27, 15, 75_011), (256, 2, 31, 15, 1 << 17), (300, 3, 21, 11, 30_011),
                                                     (64, 2, 35, 15, 20_011), (10_000, 2, 31, 16, 2_003), (40, 1, 21, 21, 9_001)])
def test_readid_on_minimizer_index(orc, hip_ctx, n_colors, n_hash, k, msz, m):
    rng = np.random.default_rng(n_colors + k + msz)
    genomes = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 4000)].tobytes() for _ in range(3)]
    oix = mini_index(orc, rng, m, n_hash, k, msz, n_colors, genomes)
    hx = to_hip_mini(hip_ctx, oix)
    for paired, d, S in ((True, 1, 3), (False, 1, 0), (True, 4, 2)):
        reads = sample_reads(orc, rng, genomes, 250, 150, paired)          # includes lower-case, N, short and repeat reads
        rep, nk, st = check(oix, hx, reads, d, S)
        assert rep[:, :n_colors].sum() > 0 and (nk.max() < 150 or msz == k)  # far fewer minimizers than windows
    # reads longer than the LDS kernel holds: sort-based path with the minimizer transform (2-bit codes; byte keys for
    # k > 32 or lower-case bases)
    long_read = genomes[0] + genomes[1] + genomes[2] * 12
    mixed = long_read[:9000] + long_read[9000:9700].lower() + long_read[9700:]
    check(oix, hx, [[long_read], [genomes[1][:500]], [mixed, genomes[0].lower()]], 1, 3)
    check(oix, hx, [[mixed], [long_read]], 3, 0)
    with pytest.raises(Exception):                                          # src/main.rs:569-573
        hx.search_count(random_kmers(rng, 10, k))
    hx.close()


@pytest.mark.parametrize("k,msz", [(31, 15), (21, 9), (40, 15), (27, 27)])
def test_bloom_insert_into_minimizer_index(orc, hip_ctx, k, msz):
    import colorid_amd
    rng = np.random.default_rng(k * 100 + msz)
    n_colors, m, n_hash = 70, 50_021, 3
    alphabet = np.frombuffer(b"ACGT" if k <= 32 else b"ACGTacgt", np.uint8)
    kmers = random_kmers(rng, 3000, k, alphabet)
    oix = orc.Index(m, n_hash, k, n_colors)
    oix.set_minimizer(msz)
    for km in kmers:
        oix.insert(5, km.tobytes())
    hx = colorid_amd.Index(hip_ctx, m, n_hash, k, n_colors).set_minimizer(msz)
    check_rc = hx.lib.cid_index_insert_kmers(hx.h, kmers.ctypes.data, 5, len(kmers))
    assert check_rc == 0
    if k <= 32:   # and through a device-resident k-mer set (colorid build on the GPU)
        ks = colorid_amd.KmerSet(hip_ctx, k)
        canon = []
        for km in kmers[:500]:
            o = orc.Kmers(k)
            o.kmerize_vector(km.tobytes(), 1)
            canon.append(o.keys()[0].tobytes())
            oix.insert(9, canon[-1])
        ks.add_seqs([km.tobytes() for km in kmers[:500]], 0)
        ks.finalize()
        assert hx.lib.cid_index_insert_kmerset(hx.h, ks.h, 9) == 0
        ks.close()
    hx.finalize()
    assert np.array_equal(hx.get_rows(np.arange(m, dtype=np.uint64)), oix.rows())
    hx.close()
