"""Colour-striped BIGSI search: the index is split by colour ranges into stripes, several per GPU
(any width, also beyond 8192 colours) and/or one set of stripes per rank (indices larger than one GPU's HBM).

Every stripe sees every k-mer.  Per-colour hits of a stripe are final.  The two per-k-mer facts that need all stripes —
"the AND word has exactly one set bit" and "some row is absent" — are merged into u32[K] device arrays by the stripe
kernels and combined across ranks (RCCL over xGMI; gloo in the CPU tests):
    fact      : ONE all_reduce(SUM) of 4 bytes per k-mer (n << 26 | colour + 1, include/colorid_hip.h) — the only
                bandwidth-relevant collective of this placement
    hits      : all_reduce(SUM) of the zero-padded full per-colour vector (8 * C bytes)
RCCL has no bitwise reduction, so the perfect search's zero_acc and AND words are all-gathered and ANDed locally.

All launches of one call go to the ctx stream back to back; when that stream is torch's current stream (bench.py) there is
no host synchronisation at all inside a call, otherwise one hand-over each way.
"""
import ctypes

import torch
import torch.distributed as dist

from ._lib import check, vp
from .dist import all_reduce_sum

MAX_RANKS = 31           # the summed n fields (2 per rank) must fit the fact word's 6 high bits
MAX_COLOURS = 1 << 20


def _active():
    return dist.is_available() and dist.is_initialized()


def reduce_stripe_facts(fact: torch.Tensor, hits_full: torch.Tensor):
    """Cross-rank combination for the proportional search (in place): int32[K] packed facts, int64[C_total]."""
    if _active():
        assert dist.get_world_size() <= MAX_RANKS
        all_reduce_sum(fact)
        all_reduce_sum(hits_full)


def reduce_perfect_facts(zero_acc: torch.Tensor, and_full: torch.Tensor):
    """Cross-rank combination for the perfect search: zero_acc int32[K] (bitwise AND over ranks), and_full int64[W64_total]
    (each rank filled only its own stripes' words, the others are all-ones: bitwise AND over ranks)."""
    if not _active():
        return zero_acc, and_full
    world = dist.get_world_size()
    dev = zero_acc.device
    if zero_acc.is_cuda and dist.get_backend() == "gloo":   # see dist.all_reduce_sum
        zero_acc, and_full = zero_acc.cpu(), and_full.cpu()
    zs = [torch.empty_like(zero_acc) for _ in range(world)]
    ws = [torch.empty_like(and_full) for _ in range(world)]
    dist.all_gather(zs, zero_acc)
    dist.all_gather(ws, and_full)
    z, w = zs[0].clone(), ws[0].clone()
    for i in range(1, world):
        z &= zs[i]
        w &= ws[i]
    return z.to(dev), w.to(dev)


class StripedIndex:
    """stripes: list of (colorid_amd.Index, colour_base) held by THIS rank; n_colors_total over all ranks."""

    def __init__(self, ctx, stripes, n_colors_total):
        assert n_colors_total <= MAX_COLOURS
        self.ctx, self.lib = ctx, ctx.lib
        self.stripes = list(stripes)
        self.n_colors = n_colors_total
        self.n_hash = self.stripes[0][0].n_hash
        self.k = self.stripes[0][0].k

    # torch's current stream and the ctx stream: the same one (no hand-over needed) or ordered by a host wait each way
    def _shared_stream(self, dev):
        return self.ctx.stream is not None and self.ctx.stream == torch.cuda.current_stream(dev).cuda_stream

    def _to_ctx(self, dev):
        if not self._shared_stream(dev):
            torch.cuda.current_stream(dev).synchronize()

    def _to_torch(self, dev):
        if not self._shared_stream(dev):
            self.ctx.synchronize()

    def search_count_local(self, d_kmers: torch.Tensor, fact: torch.Tensor, hits_full: torch.Tensor, codes: bool = False):
        """This rank's stripes only: merges into fact (int32[K], zeroed by the caller) and writes the stripes' slices of
        hits_full.  Asynchronous on the ctx stream."""
        K = d_kmers.shape[0]
        for ix, base in self.stripes:
            check(self.lib.cid_search_count_stripe_dev(self.ctx.h, ix.h, None if codes else vp(d_kmers.data_ptr()),
                                                       vp(d_kmers.data_ptr()) if codes else None, K, base,
                                                       vp(hits_full.data_ptr() + 8 * base), vp(fact.data_ptr())))

    def unique_finalize(self, fact, d_freq, nu, sf, uc):
        K = fact.shape[0]
        check(self.lib.cid_search_unique_finalize_dev(self.ctx.h, vp(fact.data_ptr()), vp(d_freq.data_ptr()) if d_freq is not None else None,
                                                      K, self.n_colors, vp(nu.data_ptr()), vp(sf.data_ptr()), vp(uc.data_ptr())))

    def search_count(self, d_kmers: torch.Tensor, d_freq: torch.Tensor = None, codes: bool = False):
        """d_kmers: uint8[K, k] (or int64[K] 2-bit codes with codes=True) on this rank's GPU, identical on every rank.
        Returns (hits, n_unique, sum_unique_freq: int64[C_total]; unique_colour: int32[K])."""
        dev = d_kmers.device
        K = d_kmers.shape[0]
        fact = torch.zeros(K, dtype=torch.int32, device=dev)
        hits_full = torch.zeros(self.n_colors, dtype=torch.int64, device=dev)
        nu = torch.zeros(self.n_colors, dtype=torch.int64, device=dev)
        sf = torch.zeros(self.n_colors, dtype=torch.int64, device=dev)
        uc = torch.empty(K, dtype=torch.int32, device=dev)
        self._to_ctx(dev)
        self.search_count_local(d_kmers, fact, hits_full, codes)
        if _active():
            self._to_torch(dev)
            reduce_stripe_facts(fact, hits_full)
            self._to_ctx(dev)
        self.unique_finalize(fact, d_freq, nu, sf, uc)
        self._to_torch(dev)
        return hits_full, nu, sf, uc

    def search_perfect(self, d_kmers: torch.Tensor, w64_total: int, word_base_of, codes: bool = False):
        """word_base_of(colour_base) -> index of the stripe's first u64 word in the full AND vector (colour_base // 64 when
        every stripe starts on a multiple of 64).  Returns (and_words int64[w64_total], any_row_missing)."""
        dev = d_kmers.device
        K = d_kmers.shape[0]
        zero_acc = torch.full((K,), -1, dtype=torch.int32, device=dev)
        and_full = torch.full((w64_total,), -1, dtype=torch.int64, device=dev)
        words = []
        for ix, _ in self.stripes:
            rs = ctypes.c_uint64(0)
            check(self.lib.cid_index_row_stride_words(ix.h, ctypes.byref(rs)))
            words.append(torch.empty(rs.value, dtype=torch.int64, device=dev))
        self._to_ctx(dev)
        for (ix, _), w in zip(self.stripes, words):
            check(self.lib.cid_search_perfect_stripe_dev(self.ctx.h, ix.h, None if codes else vp(d_kmers.data_ptr()),
                                                         vp(d_kmers.data_ptr()) if codes else None, K, vp(w.data_ptr()),
                                                         vp(zero_acc.data_ptr())))
        self._to_torch(dev)
        for (ix, base), w in zip(self.stripes, words):
            nw = (ix.n_colors + 63) // 64
            b = word_base_of(base)
            and_full[b:b + nw] = w[:nw]
        zero_acc, and_full = reduce_perfect_facts(zero_acc, and_full)
        seeds = (1 << self.n_hash) - 1 if self.n_hash < 32 else -1
        missing = bool(((zero_acc & seeds) != 0).any().item()) if K else False
        if missing:
            and_full.zero_()
        return and_full, missing

    def readid_count(self, d_bases: torch.Tensor, d_seq_off: torch.Tensor, d_read_seq0: torch.Tensor, n_reads: int, stride_d: int,
                     start_sample: int, max_read_bytes: int, max_read_windows: int):
        """read_id over the stripes (src/read_id_mt_pe.rs:66-165): device tensors as for cid_readid_count_dev, identical on every
        rank.  Returns (report int32[n_reads, C_total + 1], n_kmers int32[n_reads], status uint8[n_reads]).  Two passes: which rows
        are all-zero in every stripe (AND over the stripes, and over the ranks), then the ordered count per stripe."""
        dev = d_bases.device
        zero = torch.full((n_reads, max_read_windows), -1, dtype=torch.int32, device=dev)
        rep = torch.zeros((n_reads, self.n_colors + 1), dtype=torch.int32, device=dev)
        nk = torch.zeros(n_reads, dtype=torch.int32, device=dev)
        st = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
        args = (vp(d_bases.data_ptr()), vp(d_seq_off.data_ptr()), vp(d_read_seq0.data_ptr()), n_reads, stride_d)
        self._to_ctx(dev)
        for ix, _ in self.stripes:
            check(self.lib.cid_readid_stripe_zero_dev(self.ctx.h, ix.h, *args, max_read_bytes, max_read_windows, vp(zero.data_ptr()),
                                                      vp(nk.data_ptr()), vp(st.data_ptr())))
        rank = dist.get_rank() if _active() else 0
        if _active():
            self._to_torch(dev)
            zero, _ = reduce_perfect_facts(zero.reshape(-1), torch.zeros(1, dtype=torch.int64, device=dev))
            zero = zero.reshape(n_reads, max_read_windows).contiguous()
            self._to_ctx(dev)
        for i, (ix, base) in enumerate(self.stripes):
            check(self.lib.cid_readid_stripe_count_dev(self.ctx.h, ix.h, *args, start_sample, max_read_bytes, max_read_windows, base, self.n_colors,
                                                       1 if (rank == 0 and i == 0) else 0, vp(zero.data_ptr()), vp(rep.data_ptr()),
                                                       vp(nk.data_ptr()), vp(st.data_ptr())))
        self._to_torch(dev)
        if _active():
            all_reduce_sum(rep)       # the ranks' columns are disjoint; the no-hits column comes from rank 0 only
        return rep, nk, st

    def readid_count_routed(self, d_bases: torch.Tensor, seq_off, read_seq0, stride_d: int, start_sample: int):
        """read_id over the stripes for reads of any length (long reads, contigs, mixed batches) and stripes of any width:
        `d_bases` on the device, `seq_off` / `read_seq0` numpy uint64 on the host (cid_readid_stripe_zero / _count route every read
        between the LDS kernels and the sort-based path, per stripe).  Returns what readid_count returns."""
        import ctypes
        import numpy as np
        dev = d_bases.device
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        read_seq0 = np.ascontiguousarray(read_seq0, dtype=np.uint64)
        n_reads, n_seqs = len(read_seq0) - 1, len(seq_off) - 1
        nw = ctypes.c_uint64(0)
        check(self.lib.cid_readid_stripe_mask_words(self.stripes[0][0].k, stride_d, seq_off.ctypes.data, read_seq0.ctypes.data, n_reads, ctypes.byref(nw)))
        zero = torch.full((nw.value,), -1, dtype=torch.int32, device=dev)
        rep = torch.zeros((n_reads, self.n_colors + 1), dtype=torch.int32, device=dev)
        nk = torch.zeros(n_reads, dtype=torch.int32, device=dev)
        st = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
        args = (vp(d_bases.data_ptr()), seq_off.ctypes.data, n_seqs, read_seq0.ctypes.data, n_reads, stride_d)
        self._to_ctx(dev)
        for ix, _ in self.stripes:
            check(self.lib.cid_readid_stripe_zero(self.ctx.h, ix.h, *args, vp(zero.data_ptr()), vp(nk.data_ptr()), vp(st.data_ptr())))
        rank = dist.get_rank() if _active() else 0
        if _active():
            self._to_torch(dev)
            zero, _ = reduce_perfect_facts(zero, torch.zeros(1, dtype=torch.int64, device=dev))
            zero = zero.contiguous()
            self._to_ctx(dev)
        for i, (ix, base) in enumerate(self.stripes):
            check(self.lib.cid_readid_stripe_count(self.ctx.h, ix.h, *args, start_sample, base, self.n_colors, 1 if (rank == 0 and i == 0) else 0,
                                                   vp(zero.data_ptr()), vp(rep.data_ptr()), vp(nk.data_ptr()), vp(st.data_ptr())))
        self._to_torch(dev)
        if _active():
            all_reduce_sum(rep)
        return rep, nk, st
