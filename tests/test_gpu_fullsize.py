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


# ---------------------------------------------------------------------------------------------- configs[3] and configs[4] shapes

def _sample_rows_oracle(orc, stripes, m, n, k, hk):
    """Whole-index AND words of the k-mers `hk` (numpy [S, k]) re-derived on the host: the oracle's hash gives the row numbers,
    the rows themselves come back through cid_index_get_rows (no host copy of a 6.4 - 72 GiB matrix), stripe by stripe.
    Returns (and_bits bool[S, C_total], zero_rows bool[S]: some row of the k-mer is all-zero in every stripe)."""
    S = len(hk)
    ridx = np.array([[orc.xxh3(hk[j].tobytes(), s) % m for s in range(n)] for j in range(S)], np.uint64)
    parts, all_zero = [], np.ones((S, n), bool)
    for hx, _ in stripes:
        rows = hx.get_rows(ridx.reshape(-1)).reshape(S, n, -1)
        all_zero &= ~rows.any(axis=2)
        a = rows[:, 0].copy()
        for s in range(1, n):
            a &= rows[:, s]
        parts.append(np.unpackbits(a.view(np.uint8), axis=1, bitorder="little")[:, :hx.n_colors].astype(bool))
    return np.concatenate(parts, axis=1), all_zero.any(axis=1)


def test_config_d_shape_full_size(hip_ctx, orc):
    """BASELINE.json configs[3]'s per-GPU shape: m = 50,000,000 x 1024 colours (6.4 GB, 128-byte rows), n = 4, k = 31
    (src/batch_search_pe.rs:45-84, src/perfect_search.rs:25-52).  Additivity over a ragged partition, permutation invariance and
    determinism on 24 M distinct k-mers; the oracle on a 120,000-k-mer sample against the rows of the same index."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import colorid_amd
    dev = torch.device("cuda", 0)
    C, n, k, m = 1024, 4, 31, 50_000_000
    hx = colorid_amd.Index(hip_ctx, m, n, k, C)
    ptr, rs = hx.device_matrix()
    assert rs == 16
    bench.fill_background_fast(dev, ptr, m, rs, C, 0.2134, seed=11)
    kmers, freq, colour = bench.make_reads_kmers(dev, 44, 200_000, 150, k, C, 0.01)
    torch.cuda.synchronize()
    hx.insert_kmers_dev(kmers.data_ptr(), colour.data_ptr(), kmers.shape[0])
    hip_ctx.synchronize()
    hx.finalize()
    K = kmers.shape[0]
    assert K > 20_000_000
    whole, uc = run(hx, torch, kmers, freq)
    again, uc2 = run(hx, torch, kmers, freq)
    assert np.array_equal(whole, again) and np.array_equal(uc, uc2)
    cut = K // 2 + 29
    a, uca = run(hx, torch, kmers[:cut].clone(), freq[:cut].clone())
    b, ucb = run(hx, torch, kmers[cut:].clone(), freq[cut:].clone())
    assert np.array_equal(a + b, whole) and np.array_equal(np.concatenate([uca, ucb]), uc)
    perm = torch.randperm(K, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    p, ucp = run(hx, torch, kmers[perm].contiguous(), freq[perm].contiguous())
    assert np.array_equal(p, whole) and np.array_equal(ucp, uc[perm.cpu().numpy()])
    hits, nu, sf = whole[:C], whole[C:2 * C], whole[2 * C:]
    assert hits.sum() >= K * 0.9 and nu.sum() == (uc != -1).sum() and nu.sum() > K // 20   # ~2 false-positive colours per k-mer at this density
    assert sf.sum() == freq.cpu().numpy().astype(np.int64)[uc != -1].sum()
    # the oracle on a sample: its index holds just the rows the sample touches (the rest of its 6.4 GB stays untouched zero pages)
    S = 120_000
    hk = kmers[:S].cpu().numpy()
    hf = freq[:S].cpu().numpy()
    ridx = np.unique(np.array([orc.xxh3(hk[j].tobytes(), s) % m for j in range(S) for s in range(n)], np.uint64))
    oix = orc.Index(m, n, k, C)
    oix.rows()[ridx.astype(np.int64)] = hx.get_rows(ridx)
    want = oix.search_count(hk, hf.astype(np.uint64))
    got = hx.search_count(hk, hf.astype(np.uint32))
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    assert np.array_equal(got[3].view(np.int32), uc[:S])
    # perfect search over 4,000 of the sample's k-mers: AND words and the absent-row flag as the oracle computes them
    pw, pm = oix.search_perfect(hk[:4000])
    gw, gm = hx.search_perfect(hk[:4000])
    assert pm == gm and np.array_equal(pw, gw)
    del oix
    hx.close()


def test_config_e_shape_full_size_stripes(hip_ctx, orc):
    """BASELINE.json configs[4]'s per-GPU shape: one 512-colour stripe of an m = 2^30, n = 3 index (64 GiB resident, 64-byte
    rows) plus a second, 64-colour stripe (8 GiB) of the same index on the same GPU, searched as a StripedIndex
    (src/batch_search_pe.rs:45-84 with the exactly-one-colour rule :75-82 decided across stripes; src/perfect_search.rs:25-52).
    Checked against AND words re-derived on the host from rows read back through cid_index_get_rows (3,000-k-mer sample),
    and through additivity / permutation invariance / determinism on every k-mer."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import colorid_amd
    from colorid_amd.striped import StripedIndex
    dev = torch.device("cuda", 0)
    n, k, m = 3, 31, 1 << 30
    CA, CB = 512, 64
    C = CA + CB
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * (1 << 30):
        pytest.skip("needs ~80 GiB of free HBM")
    ha = colorid_amd.Index(hip_ctx, m, n, k, CA)
    hb = colorid_amd.Index(hip_ctx, m, n, k, CB)
    (pa, rsa), (pb, rsb) = ha.device_matrix(), hb.device_matrix()
    assert rsa == 8 and rsb == 1
    bench.fill_background_fast(dev, pa, m, rsa, CA, 1 / 64, seed=21, digits=6)
    bench.fill_background_fast(dev, pb, m, rsb, CB, 1 / 64, seed=22, digits=6)
    kmers, freq, colour = bench.make_reads_kmers(dev, 45, 200_000, 150, k, C, 0.01)
    K = kmers.shape[0]
    col_a = torch.where(colour < CA, colour, torch.full_like(colour, CA)).contiguous()
    col_b = torch.where(colour >= CA, colour - CA, torch.full_like(colour, CB)).contiguous()   # colour == C (not planted) -> CB = skipped
    # the first 6,000 k-mers also go into colour 7 of stripe A and colour 9 of stripe B: one hit on each side of the stripe cut
    extra_a = torch.full((6000,), 7, dtype=torch.int32, device=dev)
    extra_b = torch.full((6000,), 9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ha.insert_kmers_dev(kmers.data_ptr(), col_a.data_ptr(), K)
    hb.insert_kmers_dev(kmers.data_ptr(), col_b.data_ptr(), K)
    ha.insert_kmers_dev(kmers.data_ptr(), extra_a.data_ptr(), 6000)
    hb.insert_kmers_dev(kmers.data_ptr(), extra_b.data_ptr(), 6000)
    hip_ctx.synchronize()
    ha.finalize()
    hb.finalize()
    stripes = [(ha, 0), (hb, CA)]
    si = StripedIndex(hip_ctx, stripes, C)

    def srun(km, fr):
        h, nu, sf, uc = si.search_count(km, fr)
        return np.concatenate([h.cpu().numpy(), nu.cpu().numpy(), sf.cpu().numpy()]), uc.cpu().numpy()

    whole, uc = srun(kmers, freq)
    again, uc2 = srun(kmers, freq)
    assert np.array_equal(whole, again) and np.array_equal(uc, uc2)
    cut = K // 3 + 5
    a, uca = srun(kmers[:cut].clone(), freq[:cut].clone())
    b, ucb = srun(kmers[cut:].clone(), freq[cut:].clone())
    assert np.array_equal(a + b, whole) and np.array_equal(np.concatenate([uca, ucb]), uc)
    perm = torch.randperm(K, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    p, ucp = srun(kmers[perm].contiguous(), freq[perm].contiguous())
    assert np.array_equal(p, whole) and np.array_equal(ucp, uc[perm.cpu().numpy()])
    hits, nu, sf = whole[:C], whole[C:2 * C], whole[2 * C:]
    assert nu.sum() == (uc != -1).sum() and sf.sum() == freq.cpu().numpy().astype(np.int64)[uc != -1].sum()
    assert nu[:CA].sum() > K // 4 and nu[CA:].sum() > K // 40          # unique hits on both sides of the stripe cut
    assert (uc[:6000] == -1).all()                                      # one hit in each stripe: not unique in the whole index
    # the sample, re-derived from the rows
    S = 3000
    sel = np.r_[0:2000, K - 1000:K]
    hk = kmers.cpu().numpy()[sel]
    bits, _ = _sample_rows_oracle(orc, stripes, m, n, k, hk)
    pop = bits.sum(axis=1)
    want_uc = np.where(pop == 1, bits.argmax(axis=1), -1).astype(np.int32)
    assert np.array_equal(uc[sel], want_uc)
    dks = torch.from_numpy(hk.reshape(-1)).to(dev).reshape(S, k)
    dfs = freq[torch.from_numpy(sel).to(dev)].contiguous()
    sh, snu, ssf, _ = si.search_count(dks, dfs)
    assert np.array_equal(sh.cpu().numpy(), bits.sum(axis=0))
    assert np.array_equal(snu.cpu().numpy(), np.bincount(want_uc[want_uc >= 0], minlength=C))
    f_host = dfs.cpu().numpy().astype(np.int64)
    assert np.array_equal(ssf.cpu().numpy(), np.bincount(want_uc[want_uc >= 0], weights=f_host[want_uc >= 0], minlength=C).astype(np.int64))
    # perfect search over the stripes: the 2,000 doubly planted sample k-mers share colours 7 and 512+9 ...
    w64_total = (C + 63) // 64
    aw, missing = si.search_perfect(dks[:2000].contiguous(), w64_total, lambda base: base // 64)
    bits2k, zero2k = bits[:2000], _sample_rows_oracle(orc, stripes, m, n, k, hk[:2000])[1]
    assert not missing and not zero2k.any()
    got_bits = np.unpackbits(aw.cpu().numpy().view(np.uint8), bitorder="little")[:C].astype(bool)
    assert np.array_equal(got_bits, bits2k.all(axis=0)) and got_bits[7] and got_bits[CA + 9]
    # ... and random k-mer sets: a row is absent only if it is all-zero in BOTH stripes (about 1e-4 of the rows at this density)
    for nk_rand, seed in ((300, 9), (40_000, 10)):
        rk = torch.randint(0, 4, (nk_rand, k), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        rk = torch.tensor(list(b"ACGT"), device=dev, dtype=torch.uint8)[rk].contiguous()
        _, missing2 = si.search_perfect(rk, w64_total, lambda base: base // 64)
        _, zero_r = _sample_rows_oracle(orc, stripes, m, n, k, rk.cpu().numpy())
        assert missing2 == bool(zero_r.any())
    assert missing2                                                      # the large set does meet an absent row
    ha.close()
    hb.close()
