"""Parity at BASELINE.json's full index size (m = 50,000,000 rows x 256 colours = 1.6 GB, n = 4, k = 31) through
size-independent properties of the search — additivity over a partition of the k-mers, permutation invariance,
determinism of repeated launches — plus a direct oracle comparison on a sample against a host copy of the same index."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full(hip_ctx):
    import math

    import torch
    sys.path.insert(0, ROOT)
    import bench
    import colorid_amd
    dev = torch.device("cuda", 0)
    C, n, k, m = 256, 4, 31, 50_000_000
    hx = colorid_amd.Index(hip_ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 3_000_000 / m), seed=7)
    kmers, freq, colour = bench.make_reads_kmers(dev, 42, 200_000, 150, k, C, 0.01)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kmers.data_ptr(), colour.data_ptr(), kmers.shape[0])
    hip_ctx.synchronize()
    hx.finalize()
    yield hx, ptr, rs, kmers, freq, (C, n, k, m), bench
    hx.close()


def run(hx, torch, kmers, freq):
    C = hx.n_colors
    K = kmers.shape[0]
    out = torch.zeros(3 * C, dtype=torch.int64, device=kmers.device)
    uc = torch.empty(K, dtype=torch.int32, device=kmers.device)
    torch.cuda.synchronize()
    hx.search_count_dev(kmers.data_ptr(), freq.data_ptr(), K, out.data_ptr(), out.data_ptr() + 8 * C, out.data_ptr() + 16 * C,
                        uc.data_ptr())
    hx.ctx.synchronize()
    return out.cpu().numpy(), uc.cpu().numpy()


def test_additive_permutation_invariant_deterministic(full):
    import torch
    hx, _, _, kmers, freq, _, _ = full
    K = kmers.shape[0]
    assert K > 20_000_000
    whole, uc = run(hx, torch, kmers, freq)
    again, uc2 = run(hx, torch, kmers, freq)
    assert np.array_equal(whole, again) and np.array_equal(uc, uc2)             # deterministic
    cut = K // 3 + 17                                                            # ragged split, not tile aligned
    a, uca = run(hx, torch, kmers[:cut].clone(), freq[:cut].clone())
    b, ucb = run(hx, torch, kmers[cut:].clone(), freq[cut:].clone())
    assert np.array_equal(a + b, whole)                                          # additive over a partition
    assert np.array_equal(np.concatenate([uca, ucb]), uc)
    perm = torch.randperm(K, device=kmers.device, generator=torch.Generator(device=kmers.device).manual_seed(1))
    p, ucp = run(hx, torch, kmers[perm].contiguous(), freq[perm].contiguous())
    assert np.array_equal(p, whole) and np.array_equal(ucp, uc[perm.cpu().numpy()])   # order independent
    C = hx.n_colors
    hits, nu, sf = whole[:C], whole[C:2 * C], whole[2 * C:]
    assert hits.sum() >= K * 0.9 and nu.sum() == (uc != -1).sum()
    assert sf.sum() == freq.cpu().numpy().astype(np.int64)[uc != -1].sum()       # checksum of the unique-hit frequencies


def test_oracle_on_sample_of_full_index(full, orc):
    import torch
    hx, ptr, rs, kmers, freq, (C, n, k, m), bench = full
    oix = orc.Index(m, n, k, C)
    rows = oix.rows()
    step = 5_000_000
    for r0 in range(0, m, step):
        nr = min(step, m - r0)
        blk = np.empty((nr, rs * 2), np.uint32)
        bench.hip_memcpy(blk.ctypes.data, ptr + r0 * rs * 8, blk.nbytes, 2)
        rows[r0:r0 + nr, :] = blk[:, :oix.w32]
    S = 300_000
    hk = kmers[:S].cpu().numpy()
    hf = freq[:S].cpu().numpy()
    want = oix.search_count(hk, hf.astype(np.uint64))
    got = hx.search_count(hk, hf.astype(np.uint32))
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    pw, pm = oix.search_perfect(hk[:5000])
    gw, gm = hx.search_perfect(hk[:5000])
    assert pm == gm and np.array_equal(pw, gw)
