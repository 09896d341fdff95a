import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import colorid_amd
from oracle import orc
from util import random_index, to_hip_index
from test_gpu_readid import sample_reads, pack_reads

n_colors, n_hash, k, m = 256, 2, 21, 1 << 18
rng = np.random.default_rng(n_colors + k)
oix = random_index(orc, rng, m, n_hash, k, n_colors, density=0.05, zero_row_frac=0.02)
genomes = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)].tobytes() for _ in range(6)]
for gi, g in enumerate(genomes):
    km = orc.Kmers(k); km.kmerize_vector(g, 1)
    for key in km.keys():
        oix.insert(gi, key.tobytes()); oix.insert(n_colors - 1 - gi, key.tobytes())
ctx = colorid_amd.Context(0)
hx = to_hip_index(ctx, oix)
reads = sample_reads(orc, rng, genomes, 300, 100, True)
for d, S in ((1, 3), (1, 0)):
    bases, so, r0 = pack_reads(reads)
    want = oix.readid_counts(bases, so, r0, d, S); got = hx.readid_count(bases, so, r0, d, S)
    bad = np.flatnonzero((want[0] != got[0]).any(axis=1))
    print("d,S", d, S, "bad", len(bad), bad[:10])
    for b in bad[:3]:
        print(" read", b, reads[b], "nk", want[1][b], got[1][b])
        print("  want", {int(c): int(v) for c, v in enumerate(want[0][b]) if v})
        print("  got ", {int(c): int(v) for c, v in enumerate(got[0][b]) if v})
        # single-read rerun
        bb, sso, rr0 = pack_reads([reads[b]])
        g1 = hx.readid_count(bb, sso, rr0, d, S)
        print("  got alone", {int(c): int(v) for c, v in enumerate(g1[0][0]) if v})
