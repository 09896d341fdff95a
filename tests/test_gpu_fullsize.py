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


def test_readid_full_size_properties(hip_ctx, orc):
    """BASELINE.json configs[2] shape (m = 30,000,000, n = 2, k = 21, 256 colours), 300,000 paired 150-bp reads resident in
    HBM: per-read rows do not depend on batch composition or read order, mates swap symmetrically where the rules are
    symmetric (-B 0: the same k-mer set, the same absent-row outcome only without a stop), and a 1,500-read sample equals
    the oracle run on a host copy of the index."""
    import math

    import torch
    sys.path.insert(0, ROOT)
    import bench
    import colorid_amd
    dev = torch.device("cuda", 0)
    C, n, k, m, L, R = 256, 2, 21, 30_000_000, 150, 300_000
    hx = colorid_amd.Index(hip_ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    bench.fill_background(dev, ptr, m, rs, C, 1.0 - math.exp(-n * 5_000_000 / m), seed=7)
    kk, ff, cc, reads = bench.make_reads_kmers(dev, 43, 2 * R, L, k, C, 0.01, return_reads=True)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kk.data_ptr(), cc.data_ptr(), kk.shape[0])
    hip_ctx.synchronize()
    hx.finalize()
    del kk, ff, cc

    def run(read_ids, d, S):
        """rows for the read pairs `read_ids` (a device index tensor), in that order"""
        nr = read_ids.numel()
        sel = torch.stack([2 * read_ids, 2 * read_ids + 1], 1).reshape(-1)
        bases = reads[sel].reshape(-1).contiguous()
        seq_off = (torch.arange(2 * nr + 1, device=dev, dtype=torch.int64) * L).contiguous()
        read0 = (torch.arange(nr + 1, device=dev, dtype=torch.int64) * 2).contiguous()
        rep = torch.empty((nr, C + 1), dtype=torch.int32, device=dev)
        nk = torch.empty(nr, dtype=torch.int32, device=dev)
        st = torch.empty(nr, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        hx.readid_count_dev(bases.data_ptr(), seq_off.data_ptr(), read0.data_ptr(), nr, d, S, 2 * L, 2 * ((L - k) // d + 1), rep.data_ptr(),
                            nk.data_ptr(), st.data_ptr())
        hip_ctx.synchronize()
        return rep, nk, st, (bases, seq_off, read0)

    ids = torch.arange(R, device=dev)
    for d, S in ((1, 3), (1, 0), (2, 5)):
        rep, nk, st, packed = run(ids, d, S)
        assert int(st.sum()) == 0 and int(nk.max()) <= 2 * ((L - k) // d + 1) and int(nk.min()) > 0
        assert int(rep[:, :C].sum()) > R * 50                                     # planted k-mers are found
        rep2, nk2, _, _ = run(ids, d, S)
        assert torch.equal(rep, rep2) and torch.equal(nk, nk2)                    # deterministic
        perm = torch.randperm(R, device=dev, generator=torch.Generator(device=dev).manual_seed(d * 7 + S))
        repp, nkp, _, _ = run(perm, d, S)
        assert torch.equal(repp, rep[perm]) and torch.equal(nkp, nk[perm])         # a read's row does not depend on its neighbours
        cut = R // 3 + 11
        repa, nka, _, _ = run(ids[:cut], d, S)
        repb, nkb, _, _ = run(ids[cut:], d, S)
        assert torch.equal(torch.cat([repa, repb]), rep) and torch.equal(torch.cat([nka, nkb]), nk)   # nor on the batch cut
        if (d, S) == (1, 3):      # the oracle on a sample, against a host copy of the same 0.96 GB index
            rows = np.empty((m, rs * 2), np.uint32)
            bench.hip_memcpy(rows.ctypes.data, ptr, rows.nbytes, 2)
            oix = orc.Index(m, n, k, C)
            oix.rows()[:] = rows[:, :oix.w32]
            del rows
            Sn = 1500
            bases, seq_off, read0 = packed
            want = oix.readid_counts(bases[:Sn * 2 * L].cpu().numpy(), seq_off[:2 * Sn + 1].cpu().numpy().astype(np.uint64),
                                     read0[:Sn + 1].cpu().numpy().astype(np.uint64), d, S)
            assert np.array_equal(want[0], rep[:Sn].cpu().numpy().view(np.uint32))
            assert np.array_equal(want[1], nk[:Sn].cpu().numpy().view(np.uint32))
            del oix
    hx.close()
