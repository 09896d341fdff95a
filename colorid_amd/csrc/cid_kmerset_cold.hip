// The k-mer set's cold paths (see cid_kmerset_obj.hpp): everything here is built on rocPRIM — whose dispatch instantiates every primitive
// for each of its twelve target architectures, some thousand kernels in this unit's code object — and none of it lies on the path of a
// query of reads against an index (window codes -> the set's own sort -> run-length count -> search: cid_kmerset.hip, cid_partition.hpp,
// cid_scan.hpp).  The runtime loads a unit's code object on the first launch of one of its kernels: a process that never comes here
// never pays for it.  What is here: byte-string sets (k > 32: kmer.rs:87-125 on strings), LSD sorts for inputs the MSD partition does
// not take (badly skewed codes, runs beyond a workgroup's LDS), rocPRIM's merge of a later batch (CID_KMERSET_COLD_MERGE=1: the A/B copy of cid_merge.hpp),
// reordering a finished set for an index, the merged ranges of a multi-GPU set, round 1's sort-based long-read path (byte-string keys).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"
#include "cid_kmerset_obj.hpp"
#include "cid_devbuf.hpp"

namespace cid {

__global__ void k_flag_saturated(const uint32_t *counts, const uint64_t *n, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *n && counts[i] == 0xFFFFFFFFu) atomicOr(flag, 1);
}
// sort key for index locality: the 128-byte line of the k-mer's first row (bucket_bits == 0), or the slice of the index it falls
// in when the index is cut into 2^bucket_bits equal slices
__global__ void k_row0_line(const uint64_t *codes, uint32_t k, ModMagic mm, uint32_t line_shift, uint32_t bucket_bits, uint32_t *keys, uint32_t *idx,
                            uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t lsb = rev_fields(codes[i], k);
    uint32_t row0 = 0;
    xxh3_seeds_from(CodeReader{lsb}, k, 1, HashSel::of(mm), [&](uint32_t, uint64_t h) { row0 = (uint32_t)mod_m(h, mm); });
    keys[i] = bucket_bits ? (uint32_t)(((uint64_t)row0 << bucket_bits) / mm.m) : row0 >> line_shift;
    if (idx) idx[i] = (uint32_t)i;
}
// the same key for the k-mers of a byte-string set (k > 32): k ASCII bytes each, hashed as they are
struct BytesReader {
    const uint8_t *b;
    __device__ __forceinline__ uint32_t rd8(uint32_t o) const { return b[o]; }
    __device__ __forceinline__ uint32_t rd32(uint32_t o) const { return rd8(o) | (rd8(o + 1) << 8) | (rd8(o + 2) << 16) | (rd8(o + 3) << 24); }
    __device__ __forceinline__ uint64_t rd64(uint32_t o) const { return (uint64_t)rd32(o) | ((uint64_t)rd32(o + 4) << 32); }
};
__global__ void k_row0_line_ascii(const uint8_t *ascii, uint32_t k, ModMagic mm, uint32_t line_shift, uint32_t bucket_bits, uint32_t *keys, uint32_t *idx,
                                  uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t row0 = 0;
    xxh3_seeds_from(BytesReader{ascii + i * k}, k, 1, HashSel::of(mm), [&](uint32_t, uint64_t h) { row0 = (uint32_t)mod_m(h, mm); });
    keys[i] = bucket_bits ? (uint32_t)(((uint64_t)row0 << bucket_bits) / mm.m) : row0 >> line_shift;
    idx[i] = (uint32_t)i;
}
__global__ void k_permute_rows(const uint8_t *rows_in, const uint32_t *counts_in, const uint32_t *idx, uint32_t k, uint8_t *rows_out, uint32_t *counts_out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = idx[i];
    for (uint32_t t = 0; t < k; ++t) rows_out[i * k + t] = rows_in[j * k + t];
    counts_out[i] = counts_in[j];
}

}  // namespace cid

namespace {
struct KeyCodeLess {   // (row0_key, code) pairs, as a targeted set is ordered
    __host__ __device__ bool operator()(const rocprim::tuple<uint32_t, uint64_t> &a, const rocprim::tuple<uint32_t, uint64_t> &b) const {
        const uint32_t ka = rocprim::get<0>(a), kb = rocprim::get<0>(b);
        return ka < kb || (ka == kb && rocprim::get<1>(a) < rocprim::get<1>(b));
    }
};
}  // namespace

// A finalized code set's contents replaced by the merge of `total` (code, count) pairs in any order (device arrays in ks's ctx; the
// caller keeps owning them): sort by code, equal codes' counts added (cid_group_kmerset: the ranges a rank receives from the others).
int cid::kmerset_assign_merged(cid_kmerset *ks, const uint64_t *d_codes_in, const uint32_t *d_counts_in, size_t total) {
    if (!ks || ks->general || !ks->finalized) return fail(CID_ERR_STATE, "kmerset_assign_merged: a finalized 2-bit-code set is needed");
    if (total >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "%zu k-mers in one rank's share of the set (limit 2^32 - 1): use more GPUs", total);
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    DevBuf<uint64_t> uniq(ks->ctx), kout(ks->ctx), d_count(ks->ctx);
    DevBuf<uint32_t> agg(ks->ctx), vout(ks->ctx);
    int rc;
    if ((rc = uniq.alloc(total)) || (rc = agg.alloc(total)) || (rc = kout.alloc(total)) || (rc = vout.alloc(total)) || (rc = d_count.alloc(1))) return rc;
    uint64_t n_runs = 0;
    if (total) {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_codes_in, kout.p, d_counts_in, vout.p, total, 0u, ks->end_bit, st));
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tmp_bytes))) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, d_codes_in, kout.p, d_counts_in, vout.p, total, 0u, ks->end_bit, st));
        size_t tmp2 = 0;
        HIP_TRY(rocprim::reduce_by_key(nullptr, tmp2, kout.p, vout.p, total, uniq.p, agg.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
        DevBuf<uint8_t> t2(ks->ctx);
        if ((rc = t2.alloc(tmp2))) return rc;
        HIP_TRY(rocprim::reduce_by_key(t2.p, tmp2, kout.p, vout.p, total, uniq.p, agg.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
        HIP_TRY(hipMemsetAsync(ks->d_flags + 1, 0, 4, st));
        hipLaunchKernelGGL(cid::k_flag_saturated, dim3(grid_for_n(total)), dim3(256), 0, st, agg.p, d_count.p, ks->d_flags + 1);
        int sat = 0;
        HIP_TRY(hipMemcpyAsync(&sat, ks->d_flags + 1, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&n_runs, d_count.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (sat) return fail(CID_ERR_UNSUPPORTED,
                             "a k-mer occurs more than 2^32 - 2 times: beyond the u32 multiplicities of the GPU k-mer set (count on the host)");
    }
    if (ks->codes) cid::ctx_free(ks->ctx, ks->codes);
    if (ks->counts) cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = uniq.release();
    ks->counts = agg.release();
    ks->n = n_runs;
    return CID_OK;
}


// ------------------------------------------------------------------------------------------------ long reads (read_id)
// Per-read distinct k-mers in first-occurrence order for reads whose k-mer set does not fit a wave's LDS:
// window codes -> one stable radix sort by code (windows are laid out read by read, so inside a run of equal codes
// the entries of one read are adjacent and ascending) -> first-occurrence flags -> exclusive scan -> ordered lists.
namespace cid {

__global__ void k_iota_u32(uint32_t *p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_first_flags(const uint64_t *sorted_codes, const uint32_t *sorted_idx, const uint64_t *wstart, uint32_t n_reads,
                              uint64_t sentinel, uint32_t *flags, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = sorted_codes[i];
    const uint32_t w = sorted_idx[i];
    bool first = code != sentinel;
    if (first && i > 0 && sorted_codes[i - 1] == code)
        first = read_of_window(wstart, n_reads, sorted_idx[i - 1]) != read_of_window(wstart, n_reads, w);
    flags[w] = first ? 1u : 0u;
}
__global__ void k_scatter_list(const uint64_t *codes, const uint32_t *flags, const uint32_t *pos, uint64_t *list, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) list[pos[i]] = codes[i];
}
__global__ void k_list_starts(const uint64_t *wstart, const uint32_t *pos, uint64_t *list_start, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) list_start[r] = pos[wstart[r]];
}

// ---- general keys: k > 32 or lower-case bases (their case is kept), so a key is a byte string.
// One thread per window: validity, canonical orientation and (for .mxi) the minimizer, all on raw bytes as the reference
// compares them.  A key is described by where its bytes sit in `bases`: entry = offset << 1 | reverse-complement flag
// (minimizers are upper-cased afterwards, kmer.rs:381); its sort image is 4 bits per base (2-bit base | lower-case << 2),
// 16 bases per word, word-major arrays; a window without a key gets all-ones words (no base encodes to 0xF).
__device__ __forceinline__ uint32_t key_byte(const uint8_t *bases, uint64_t entry, uint32_t klen, bool upper, uint32_t t) {
    const uint64_t off = entry >> 1;
    uint32_t b = (entry & 1ull) ? switch_base_dev(bases[off + klen - 1 - t]) : bases[off + t];
    if (upper && b >= 'a' && b <= 'z') b -= 32u;
    return b;
}
__global__ __launch_bounds__(256) void k_general_keys(const uint8_t *bases, const Segment *segs, uint32_t n_segs, uint64_t W, uint32_t k,
                                                      uint32_t msz, uint32_t n_words, uint64_t *keyw, uint64_t *entry, uint32_t upper_keys = 0) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    uint32_t lo = 0, hi = n_segs;   // the segment holding window w: largest s with segs[s].out_off <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].out_off <= w) lo = mid; else hi = mid;
    }
    const uint64_t pos = segs[lo].base_off + (w - segs[lo].out_off) * segs[lo].stride;
    const uint8_t *b = bases + pos;
    bool ok = true;
    for (uint32_t t = 0; t < k; ++t) ok = ok && good_base_dev(b[t]);
    if (!ok) {
        entry[w] = ~0ull;
        for (uint32_t j = 0; j < n_words; ++j) keyw[(uint64_t)j * W + w] = ~0ull;
        return;
    }
    uint32_t rc = 1;   // palindromes take the reverse-complement branch (the same string)
    for (uint32_t t = 0; t < k; ++t) {
        const uint32_t f = b[t], r = switch_base_dev(b[k - 1 - t]);
        if (f != r) { rc = f < r ? 0u : 1u; break; }
    }
    uint64_t e = (pos << 1) | rc;
    uint32_t klen = k;
    if (msz) {   // find_minimizer (kmer.rs:971-986) over the canonical string: candidates (i, reverse-complement)
        const uint64_t canon = e;
        auto cand_byte = [&](uint32_t cand, uint32_t t) -> uint32_t {
            const uint32_t i = cand & 0xFFFFu;
            return (cand >> 16) ? (uint32_t)switch_base_dev((uint8_t)key_byte(bases, canon, k, false, i + msz - 1 - t))
                                : key_byte(bases, canon, k, false, i + t);
        };
        auto less = [&](uint32_t x, uint32_t y) -> bool {
            for (uint32_t t = 0; t < msz; ++t) {
                const uint32_t bx = cand_byte(x, t), by = cand_byte(y, t);
                if (bx != by) return bx < by;
            }
            return false;
        };
        uint32_t best = 0;
        for (uint32_t i = 1; i + msz <= k; ++i) {
            if (less(i, best)) best = i;
            if (less(i | (1u << 16), best)) best = i | (1u << 16);
        }
        // the minimizer as a stretch of `bases`: canonical byte j is b[j] (rc = 0) or comp(b[k-1-j]) (rc = 1)
        const uint32_t i = best & 0xFFFFu, mrc = best >> 16;
        const uint64_t off = rc ? pos + k - i - msz : pos + i;
        e = (off << 1) | (rc ^ mrc);
        klen = msz;
    }
    entry[w] = e;
    for (uint32_t j = 0; j < n_words; ++j) {
        uint64_t word = 0;
        for (uint32_t t = 16 * j; t < 16 * j + 16 && t < klen; ++t) {
            const uint32_t c = key_byte(bases, e, klen, msz != 0 || upper_keys != 0, t);
            word |= (uint64_t)(((c >> 1) & 3u) | ((c >> 3) & 4u)) << (4u * (t & 15u));
        }
        keyw[(uint64_t)j * W + w] = word;
    }
}
__global__ void k_gather_u64(const uint64_t *src, const uint32_t *idx, uint64_t *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_first_flags_general(const uint64_t *keyw, uint32_t n_words, const uint64_t *entry, const uint32_t *sorted_idx,
                                      const uint64_t *wstart, uint32_t n_reads, uint32_t *flags, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    const uint32_t w = sorted_idx[i];
    bool first = entry[w] != ~0ull;
    if (first && i > 0) {
        const uint32_t pw = sorted_idx[i - 1];
        bool same = true;
        for (uint32_t j = 0; j < n_words && same; ++j) same = keyw[(uint64_t)j * W + w] == keyw[(uint64_t)j * W + pw];
        if (same) first = read_of_window(wstart, n_reads, pw) != read_of_window(wstart, n_reads, w);
    }
    flags[w] = first ? 1u : 0u;
}

// ---- byte-string k-mer sets (k > 32): run boundaries over the whole sorted window list
__global__ void k_first_flags_set(const uint64_t *keyw, uint32_t n_words, const uint64_t *entry, const uint32_t *sorted_idx, uint32_t *flags,
                                  uint32_t *valid, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > W) return;
    if (i == W) { flags[i] = 0; valid[i] = 0; return; }   // slot W receives the totals
    const uint32_t w = sorted_idx[i];
    const bool ok = entry[w] != ~0ull;
    bool first = ok;
    if (first && i > 0) {
        const uint32_t pw = sorted_idx[i - 1];
        bool same = true;
        for (uint32_t j = 0; j < n_words && same; ++j) same = keyw[(uint64_t)j * W + w] == keyw[(uint64_t)j * W + pw];
        first = !same;
    }
    flags[i] = first ? 1u : 0u;
    valid[i] = ok ? 1u : 0u;
}
// run j starts at sorted position starts[j]; starts[n_runs] = number of valid windows (they sort before the invalid ones)
__global__ void k_set_starts(const uint32_t *flags, const uint32_t *pos, const uint32_t *sorted_idx, const uint64_t *entry, uint32_t *starts,
                             uint64_t *run_entry, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W && flags[i]) { starts[pos[i]] = (uint32_t)i; run_entry[pos[i]] = entry[sorted_idx[i]]; }
}
__global__ void k_run_counts(const uint32_t *starts, uint32_t n_runs, uint32_t n_valid, uint32_t *counts) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_runs) counts[j] = (j + 1 < n_runs ? starts[j + 1] : n_valid) - starts[j];
}
__global__ void k_entries_to_ascii(const uint8_t *bases, const uint64_t *run_entry, uint32_t k, uint32_t upper, uint8_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = run_entry[i];
    for (uint32_t t = 0; t < k; ++t) out[i * k + t] = (uint8_t)key_byte(bases, e, k, upper != 0, t);
}
__global__ void k_compact_rows(const uint8_t *rows_in, const uint32_t *counts_in, const uint32_t *keep, const uint32_t *pos, uint32_t k,
                               uint8_t *rows_out, uint32_t *counts_out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint64_t o = pos[i];
    for (uint32_t t = 0; t < k; ++t) rows_out[o * k + t] = rows_in[i * k + t];
    counts_out[o] = counts_in[i];
}
__global__ void k_keep_gt(const uint32_t *counts, uint64_t t, uint32_t *keep, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    keep[i] = (i < n && counts[i] > t) ? 1u : 0u;
}

// d_bases resident; host seq_off / read_seq0.  Writes report / n_kmers / status to DEVICE arrays.
// start-up: this unit's code object — rocPRIM's few thousand kernels, 10 MB, ~40 ms to load — ahead of the first call that needs it
// (cid_warmup(CID_WARM_COLD): read_id's sorting path for long reads with k > 32 or soft-masked reads of several hash buckets)
__global__ void k_merge_status(const uint8_t *mine, uint8_t *status, uint64_t n, int only_routed) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && (!only_routed || mine[r] != 2)) status[r] = mine[r];
}
hipError_t warm_cold() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_merge_status));
}

int readid_long_sorted(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                       size_t n_reads, uint32_t stride_d, uint32_t start_sample, const uint8_t *route, bool clear_wide, uint32_t *d_report,
                       uint32_t *d_n_kmers, uint8_t *d_status, const StripePass &sp, bool merge_status) {
    const uint32_t k = index_k(ix);
    hipStream_t st = ctx_stream(c);
    const uint32_t msz = index_m_size(ix);           // > 0: the sets hold minimizers of length msz
    const uint32_t key_len = msz ? msz : k;
    bool general = k > 32;                           // byte-string keys; also taken when a lower-case base shows up
    const uint64_t sentinel_k = k < 32 ? (1ull << (2 * k)) : ~0ull;
    const uint64_t sentinel = key_len < 32 ? (1ull << (2 * key_len)) : ~0ull;
    const unsigned end_bit = key_len < 32 ? 2 * key_len + 1 : 64;
    // windows are numbered read by read, mate by mate
    std::vector<uint64_t> wstart(n_reads + 1, 0);
    std::vector<uint8_t> status(n_reads, 0);
    std::vector<Segment> segs;
    const uint32_t seg_win = kSegWindows / stride_d ? kSegWindows / stride_d : 1;
    uint64_t W = 0;
    for (size_t r = 0; r < n_reads; ++r) {
        wstart[r] = W;
        if (route && !route[r]) { status[r] = 2; continue; }
        const uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
        if (s1 == s0 || seq_off[s0 + 1] - seq_off[s0] < k) { status[r] = 1; continue; }   // too_short (first mate only)
        for (uint64_t s = s0; s < s1; ++s) {
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len < k) continue;
            const uint64_t nw = (len - k) / stride_d + 1;
            for (uint64_t w0 = 0; w0 < nw; w0 += seg_win) {
                const uint32_t m = (uint32_t)(nw - w0 < seg_win ? nw - w0 : seg_win);
                segs.push_back(Segment{seq_off[s] + w0 * stride_d, W, m, stride_d});
                W += m;
            }
        }
    }
    wstart[n_reads] = W;
    if (W >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    const size_t C1 = (size_t)index_n_colors(ix) + 1;
    // the statuses this pass works with are its own; the caller's array takes them at the end — all of them, or (merge_status: a few reads
    // of a batch redone here, cid_readlong.hip) only those of the routed reads
    DevBuf<uint8_t> d_st(c);
    if (int rc0 = d_st.alloc(n_reads)) return rc0;
    HIP_TRY(hipMemcpyAsync(d_st.p, status.data(), n_reads, hipMemcpyHostToDevice, st));
    DevBuf<uint64_t> d_wstart(c), d_codes(c), d_sorted(c), d_list(c), d_lstart(c);
    DevBuf<uint32_t> d_idx(c), d_sidx(c), d_flags(c), d_pos(c);
    DevBuf<Segment> d_segs(c);
    DevBuf<int> d_lower(c);
    int rc;
    if ((rc = d_wstart.alloc(n_reads + 1)) || (rc = d_codes.alloc(W + 1)) || (rc = d_sorted.alloc(W + 1)) || (rc = d_idx.alloc(W + 1)) ||
        (rc = d_sidx.alloc(W + 1)) || (rc = d_flags.alloc(W + 1)) || (rc = d_pos.alloc(W + 1)) || (rc = d_segs.alloc(segs.size())) ||
        (rc = d_lstart.alloc(n_reads + 1)) || (rc = d_lower.alloc(4))) return rc;
    HIP_TRY(hipMemcpyAsync(d_wstart.p, wstart.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_lower.p, 0, 16, st));
    HIP_TRY(hipMemsetAsync(d_flags.p, 0, (W + 1) * 4, st));
    if (W) {
        HIP_TRY(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice, st));
        if (!general) {
            constexpr uint32_t kBytes = kSegWindows + 32 + 96;
            const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
            unsigned grid = (unsigned)((segs.size() + 3) / 4);
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, d_bases, d_segs.p, (uint32_t)segs.size(), k, 1, sentinel_k,
                               d_codes.p, d_lower.p, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr, KeyFor{},
                               (const uint32_t *)nullptr, (uint8_t *)nullptr);
            int lower = 0;
            HIP_TRY(hipMemcpyAsync(&lower, d_lower.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            general = lower != 0;   // case-preserving k-mers cannot be packed in 2 bits per base
        }
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for_n(W)), dim3(256), 0, st, d_idx.p, (uint64_t)W);
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_codes.p, d_sorted.p, d_idx.p, d_sidx.p, W, 0u, 64u, st));
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb))) return rc;
        if (!general) {
            if (msz) hipLaunchKernelGGL(k_codes_to_minimizers, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, (uint64_t)W, k, msz, sentinel_k, sentinel);
            HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, d_codes.p, d_sorted.p, d_idx.p, d_sidx.p, W, 0u, end_bit, st));
            hipLaunchKernelGGL(k_first_flags, dim3(grid_for_n(W)), dim3(256), 0, st, d_sorted.p, d_sidx.p, d_wstart.p, (uint32_t)n_reads,
                               sentinel, d_flags.p, (uint64_t)W);
        } else {
            // stable LSD radix sort over the key words; d_codes ends up holding the entries the search kernel reads
            const uint32_t n_words = (key_len + 15) / 16;
            DevBuf<uint64_t> d_keyw(c), d_gath(c);
            if ((rc = d_keyw.alloc((size_t)n_words * W)) || (rc = d_gath.alloc(W))) return rc;
            hipLaunchKernelGGL(k_general_keys, dim3(grid_for_n(W)), dim3(256), 0, st, d_bases, d_segs.p, (uint32_t)segs.size(), (uint64_t)W, k, msz,
                               n_words, d_keyw.p, d_codes.p);
            uint32_t *cur = d_idx.p, *nxt = d_sidx.p;
            for (uint32_t j = 0; j < n_words; ++j) {
                hipLaunchKernelGGL(k_gather_u64, dim3(grid_for_n(W)), dim3(256), 0, st, d_keyw.p + (size_t)j * W, cur, d_gath.p, (uint64_t)W);
                HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, d_gath.p, d_sorted.p, cur, nxt, W, 0u, 64u, st));
                std::swap(cur, nxt);
            }
            hipLaunchKernelGGL(k_first_flags_general, dim3(grid_for_n(W)), dim3(256), 0, st, d_keyw.p, n_words, d_codes.p, cur, d_wstart.p,
                               (uint32_t)n_reads, d_flags.p, (uint64_t)W);
            HIP_TRY(hipStreamSynchronize(st));   // d_keyw / d_gath go out of scope
        }
    }
    size_t tb2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, d_flags.p, d_pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    DevBuf<uint8_t> tmp2(c);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, d_flags.p, d_pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    uint32_t D = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&D, d_pos.p + W, 4, hipMemcpyDeviceToHost));
    if ((rc = d_list.alloc(D))) return rc;
    if (W) hipLaunchKernelGGL(k_scatter_list, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, d_flags.p, d_pos.p, d_list.p, (uint64_t)W);
    hipLaunchKernelGGL(k_list_starts, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, d_wstart.p, d_pos.p, d_lstart.p,
                       (uint32_t)n_reads);
    ReadIdListParams p{};
    p.mat = index_matrix(ix); p.rs = index_rs(ix); p.w64 = (index_n_colors(ix) + 63) / 64; p.n_colors = index_n_colors(ix);
    p.n_hash = index_n_hash(ix); p.k = key_len; p.mod = index_mod(ix);
    p.list_codes = d_list.p; p.list_start = d_lstart.p; p.n_reads = n_reads; p.start_sample = start_sample;
    p.bases = general ? d_bases : nullptr; p.upper = msz != 0;
    p.hist_pad = p.rs > 128 ? 4u * p.rs : (uint32_t)((C1 + 3) & ~(size_t)3);
    if (p.rs > 128 && clear_wide && !sp.on()) HIP_TRY(hipMemsetAsync(d_report, 0, n_reads * C1 * 4, st));   // wide rows count in place
    p.zero_acc = sp.zero_acc; p.zero_in = sp.zero_in; p.zero_start = sp.zero_start;   // a colour stripe's pass: the caller zeroed the report
    p.colour_base = sp.colour_base; p.report_width = sp.report_width; p.write_nohits = sp.write_nohits;
    p.wave_bytes = (uint32_t)((4ull * kWave * p.n_hash + 4ull * p.hist_pad + 15) & ~15ull);
    if ((size_t)(kBlock / kWave) * p.wave_bytes > 160u * 1024u) return fail(CID_ERR_UNSUPPORTED, "LDS need exceeds 160 KiB");
    p.report = d_report; p.n_kmers = d_n_kmers; p.status = d_st.p;
    uint64_t grid = (n_reads + 3) / 4;
    if (grid > 4096) grid = 4096;
    HIP_TRY(launch_readid_list(p, (int)grid, st));
    hipLaunchKernelGGL(k_merge_status, dim3(grid_for_n(n_reads)), dim3(256), 0, st, d_st.p, d_status, (uint64_t)n_reads, merge_status ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return CID_OK;
}

}  // namespace cid

static int finalize_general(cid_kmerset *ks) {
    using namespace cid;
    cid_ctx *c = ks->ctx;
    hipStream_t st = ctx_stream(c);
    const uint64_t W = ks->g_windows;
    const uint32_t k = ks->k;
    ks->n = 0;
    if (W == 0) return CID_OK;
    if (W >= (1ull << 32) - 1) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 - 2 k-mer windows in one byte-string k-mer set");
    const uint32_t n_words = (k + 15) / 16;
    const uint32_t upper = ks->g_mode == 0 ? 1u : 0u;   // kmerize_vector upper-cases after the canonical choice (kmer.rs:104-117); fastq keeps case
    DevBuf<Segment> d_segs(c);
    DevBuf<uint64_t> keyw(c), entry(c), gath(c), sorted(c), run_entry(c);
    DevBuf<uint32_t> idx(c), idx2(c), flags(c), valid(c), pos(c), vpos(c), starts(c);
    int rc;
    if ((rc = d_segs.alloc(ks->g_segs.size())) || (rc = keyw.alloc((size_t)n_words * W)) || (rc = entry.alloc(W)) || (rc = gath.alloc(W)) ||
        (rc = sorted.alloc(W)) || (rc = idx.alloc(W)) || (rc = idx2.alloc(W)) || (rc = flags.alloc(W + 1)) || (rc = valid.alloc(W + 1)) ||
        (rc = pos.alloc(W + 1)) || (rc = vpos.alloc(W + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(d_segs.p, ks->g_segs.data(), ks->g_segs.size() * sizeof(Segment), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_general_keys, dim3(grid_for_n(W)), dim3(256), 0, st, ks->g_bases, d_segs.p, (uint32_t)ks->g_segs.size(), W, k, 0u, n_words,
                       keyw.p, entry.p, upper);
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for_n(W)), dim3(256), 0, st, idx.p, W);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, gath.p, sorted.p, idx.p, idx2.p, W, 0u, 64u, st));
    DevBuf<uint8_t> tmp(c);
    if ((rc = tmp.alloc(tb))) return rc;
    uint32_t *cur = idx.p, *nxt = idx2.p;
    for (uint32_t j = 0; j < n_words; ++j) {   // stable LSD passes, least significant word (the key's LAST bases) first
        const uint32_t word = n_words - 1 - j;
        hipLaunchKernelGGL(k_gather_u64, dim3(grid_for_n(W)), dim3(256), 0, st, keyw.p + (size_t)word * W, cur, gath.p, W);
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, gath.p, sorted.p, cur, nxt, W, 0u, 64u, st));
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL(k_first_flags_set, dim3(grid_for_n(W + 1)), dim3(256), 0, st, keyw.p, n_words, entry.p, cur, flags.p, valid.p, W);
    size_t tb2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, flags.p, pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    DevBuf<uint8_t> tmp2(c);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, flags.p, pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, valid.p, vpos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    uint32_t n_runs = 0, n_valid = 0;
    HIP_TRY(hipMemcpyAsync(&n_runs, pos.p + W, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_valid, vpos.p + W, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_runs) {
        DevBuf<uint32_t> counts(c);
        DevBuf<uint8_t> ascii(c);
        if ((rc = starts.alloc(n_runs)) || (rc = run_entry.alloc(n_runs)) || (rc = counts.alloc(n_runs)) || (rc = ascii.alloc((size_t)n_runs * k))) return rc;
        hipLaunchKernelGGL(k_set_starts, dim3(grid_for_n(W)), dim3(256), 0, st, flags.p, pos.p, cur, entry.p, starts.p, run_entry.p, W);
        hipLaunchKernelGGL(k_run_counts, dim3(grid_for_n(n_runs)), dim3(256), 0, st, starts.p, n_runs, n_valid, counts.p);
        hipLaunchKernelGGL(k_entries_to_ascii, dim3(grid_for_n(n_runs)), dim3(256), 0, st, ks->g_bases, run_entry.p, k, upper, ascii.p, (uint64_t)n_runs);
        HIP_TRY(hipStreamSynchronize(st));
        ks->ascii = ascii.release();
        ks->counts = counts.release();
    }
    ks->n = n_runs;
    HIP_TRY(hipStreamSynchronize(st));
    return CID_OK;
}


namespace cid {

int cold_sort_keys_u64(cid_ctx *c, hipStream_t st, const uint64_t *in, uint64_t *out, size_t n, unsigned b0, unsigned b1) {
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, in, out, n, b0, b1, st));
    DevBuf<uint8_t> tmp(c);
    int rc = tmp.alloc(tb);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_keys(tmp.p, tb, in, out, n, b0, b1, st));
    return CID_OK;   // (the scratch goes back to the ctx's block cache: later work on the same stream is ordered behind the sort)
}
int cold_sort_pairs_u64_u32(cid_ctx *c, hipStream_t st, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned b0,
                            unsigned b1) {
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, b0, b1, st));
    DevBuf<uint8_t> tmp(c);
    int rc = tmp.alloc(tb);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, kin, kout, vin, vout, n, b0, b1, st));
    return CID_OK;
}
int cold_sort_pairs_u32_u64(cid_ctx *c, hipStream_t st, const uint32_t *kin, uint32_t *kout, const uint64_t *vin, uint64_t *vout, size_t n, unsigned b0,
                            unsigned b1) {
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, b0, b1, st));
    DevBuf<uint8_t> tmp(c);
    int rc = tmp.alloc(tb);
    if (rc) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, kin, kout, vin, vout, n, b0, b1, st));
    return CID_OK;
}

int cold_run_length_u64(cid_ctx *c, hipStream_t st, const uint64_t *sorted, size_t n, uint64_t *uniq, uint32_t *runs, uint64_t *d_n_runs) {
    size_t tb = 0;
    HIP_TRY(rocprim::run_length_encode(nullptr, tb, sorted, n, uniq, runs, d_n_runs, st));
    DevBuf<uint8_t> tmp(c);
    int rc = tmp.alloc(tb);
    if (rc) return rc;
    HIP_TRY(rocprim::run_length_encode(tmp.p, tb, sorted, n, uniq, runs, d_n_runs, st));
    return CID_OK;
}

int kmerset_merge_batch(cid_kmerset *ks, const uint64_t *uniq_p, const uint32_t *agg_p, uint64_t n_runs) {
    hipStream_t st = cid::ctx_stream(ks->ctx);
    struct { const uint64_t *p; } uniq{uniq_p};
    struct { const uint32_t *p; } agg{agg_p};
    DevBuf<uint64_t> d_count(ks->ctx);
    int rc;
    if ((rc = d_count.alloc(1))) return rc;
    const size_t total = ks->n + n_runs;
    DevBuf<uint64_t> mk(ks->ctx), ok(ks->ctx);
    DevBuf<uint32_t> mv(ks->ctx), ov(ks->ctx);
    if ((rc = mk.alloc(total)) || (rc = mv.alloc(total)) || (rc = ok.alloc(total)) || (rc = ov.alloc(total))) return rc;
    size_t tb = 0;
    if (ks->targeted) {   // both lists are in (row0_key, code) order: merged on that pair (the keys are recomputed from the codes, not kept)
        DevBuf<uint32_t> ka(ks->ctx), kb(ks->ctx), kout(ks->ctx);
        if ((rc = ka.alloc(ks->n)) || (rc = kb.alloc(n_runs)) || (rc = kout.alloc(total))) return rc;
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->codes, ks->k, ks->key_for, ka.p, (uint64_t)ks->n);
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(n_runs)), dim3(256), 0, st, uniq.p, ks->k, ks->key_for, kb.p, (uint64_t)n_runs);
        HIP_TRY(hipGetLastError());
        auto in_a = rocprim::make_zip_iterator(rocprim::make_tuple(ka.p, ks->codes));
        auto in_b = rocprim::make_zip_iterator(rocprim::make_tuple(kb.p, uniq.p));
        auto out_k = rocprim::make_zip_iterator(rocprim::make_tuple(kout.p, mk.p));
        HIP_TRY(rocprim::merge(nullptr, tb, in_a, in_b, out_k, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, KeyCodeLess(), st));
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::merge(tmp.p, tb, in_a, in_b, out_k, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, KeyCodeLess(), st));
        HIP_TRY(hipStreamSynchronize(st));   // (ka / kb / kout are released here)
    } else {
    HIP_TRY(rocprim::merge(nullptr, tb, ks->codes, uniq.p, mk.p, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, rocprim::less<uint64_t>(), st));
    {
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::merge(tmp.p, tb, ks->codes, uniq.p, mk.p, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, rocprim::less<uint64_t>(), st));
    }
    }
    size_t tmp2 = 0;
    HIP_TRY(rocprim::reduce_by_key(nullptr, tmp2, mk.p, mv.p, total, ok.p, ov.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
    DevBuf<uint8_t> t2(ks->ctx);
    if ((rc = t2.alloc(tmp2))) return rc;
    HIP_TRY(rocprim::reduce_by_key(t2.p, tmp2, mk.p, mv.p, total, ok.p, ov.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
    // multiplicities are u32 (the reference: usize): a sum that saturated cannot be reported faithfully
    HIP_TRY(hipMemsetAsync(ks->d_flags + 1, 0, 4, st));
    hipLaunchKernelGGL(cid::k_flag_saturated, dim3(grid_for_n(total)), dim3(256), 0, st, ov.p, d_count.p, ks->d_flags + 1);
    int sat = 0;
    uint64_t n_merged = 0;
    HIP_TRY(hipMemcpyAsync(&sat, ks->d_flags + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_merged, d_count.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sat) return fail(CID_ERR_UNSUPPORTED,
                         "a k-mer occurs more than 2^32 - 2 times: beyond the u32 multiplicities of the GPU k-mer set (count on the host)");
    cid::ctx_free(ks->ctx, ks->codes);
    cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = ok.release();
    ks->counts = ov.release();
    ks->n = n_merged;
    return CID_OK;
}

int kmerset_clean_general(cid_kmerset *ks, uint64_t t) {
    hipStream_t st = cid::ctx_stream(ks->ctx);
        DevBuf<uint32_t> keep(ks->ctx), pos(ks->ctx), oc(ks->ctx);
        DevBuf<uint8_t> orows(ks->ctx), tmp(ks->ctx);
        int rc;
        if ((rc = keep.alloc(ks->n + 1)) || (rc = pos.alloc(ks->n + 1))) return rc;
        hipLaunchKernelGGL(cid::k_keep_gt, dim3(grid_for_n(ks->n + 1)), dim3(256), 0, st, ks->counts, t, keep.p, (uint64_t)ks->n);
        size_t tb = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, tb, keep.p, pos.p, 0u, ks->n + 1, rocprim::plus<uint32_t>(), st));
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, keep.p, pos.p, 0u, ks->n + 1, rocprim::plus<uint32_t>(), st));
        uint32_t kept = 0;
        HIP_TRY(hipMemcpyAsync(&kept, pos.p + ks->n, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = orows.alloc((size_t)kept * ks->k)) || (rc = oc.alloc(kept))) return rc;
        hipLaunchKernelGGL(cid::k_compact_rows, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->counts, keep.p, pos.p, ks->k, orows.p, oc.p,
                           (uint64_t)ks->n);
        HIP_TRY(hipStreamSynchronize(st));
        cid::ctx_free(ks->ctx, ks->ascii); cid::ctx_free(ks->ctx, ks->counts);
        ks->ascii = orows.release(); ks->counts = oc.release(); ks->n = kept;
        return CID_OK;
}

}  // namespace cid

int cid::kmerset_finalize_general(cid_kmerset *ks) { return finalize_general(ks); }

extern "C" {

// Device arrays in, device arrays out (asynchronous on the ctx stream): the n k-mers (2-bit codes + multiplicities) grouped by the
// 128-byte index line of their first row (cid_ctx_tune "order_bits" > 0: by that many leading bits of its position instead).  The
// codes and the multiplicities each ride through their own stable radix sort on that key (same permutation), restricted to the
// key's significant bits: no index array, no random gather.
int cid_order_codes_for_index_dev(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, const uint32_t *d_counts, size_t n,
                                  uint64_t *d_codes_out, uint32_t *d_counts_out) {
    if (!c || !ix || (n && (!d_codes || !d_codes_out)) || (d_counts && !d_counts_out)) return fail(CID_ERR_INVALID, "null argument");
    if (n == 0) return CID_OK;
    if (n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mers");
    if (cid::index_k(ix) > 32) return fail(CID_ERR_UNSUPPORTED, "2-bit codes need k_size <= 32");
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    hipStream_t st = cid::ctx_stream(c);
    const uint32_t rs = cid::index_rs(ix);
    uint32_t line_shift = 0;
    while ((rs << line_shift) < 16) ++line_shift;   // rows per 128-byte line = 16 / rs
    const uint32_t bucket_bits = (uint32_t)cid::ctx_order_bits(c);
    const uint64_t max_key = bucket_bits ? ((1ull << bucket_bits) - 1) : ((cid::index_mod(ix).m - 1) >> line_shift);
    unsigned end_bit = 1;
    while (end_bit < 32 && (max_key >> end_bit)) ++end_bit;
    DevBuf<uint32_t> keys(c), keys2(c);
    int rc;
    if ((rc = keys.alloc(n)) || (rc = keys2.alloc(n))) return rc;
    hipLaunchKernelGGL(cid::k_row0_line, dim3(grid_for_n(n)), dim3(256), 0, st, d_codes, cid::index_k(ix), cid::index_mod(ix), line_shift, bucket_bits,
                       keys.p, (uint32_t *)nullptr, (uint64_t)n);
    size_t tb = 0, tb2 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, d_codes, d_codes_out, n, 0u, end_bit, st));
    if (d_counts) HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, keys.p, keys2.p, d_counts, d_counts_out, n, 0u, end_bit, st));
    DevBuf<uint8_t> tmp(c);
    if ((rc = tmp.alloc(tb > tb2 ? tb : tb2))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys.p, keys2.p, d_codes, d_codes_out, n, 0u, end_bit, st));
    if (d_counts) HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb2, keys.p, keys2.p, d_counts, d_counts_out, n, 0u, end_bit, st));
    return CID_OK;   // the scratch goes back to the ctx's block cache; later work on the same stream is ordered behind these kernels
}

int cid_kmerset_order_for_index(cid_kmerset *ks, const cid_index *ix) {
    if (!ks || !ix) return fail(CID_ERR_INVALID, "null argument");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->n == 0) return CID_OK;
    if (cid::index_k(ix) != ks->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", ks->k, cid::index_k(ix));
    if (ks->general) {   // byte strings: the same key from the ASCII k-mer, the rows permuted
        if (ks->n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mers");
        cid_ctx *c = ks->ctx;
        HIP_TRY(hipSetDevice(cid::ctx_device(c)));
        hipStream_t st = cid::ctx_stream(c);
        const uint32_t rs = cid::index_rs(ix);
        uint32_t line_shift = 0;
        while ((rs << line_shift) < 16) ++line_shift;
        const uint32_t bucket_bits = (uint32_t)cid::ctx_order_bits(c);
        const uint64_t max_key = bucket_bits ? ((1ull << bucket_bits) - 1) : ((cid::index_mod(ix).m - 1) >> line_shift);
        unsigned end_bit = 1;
        while (end_bit < 32 && (max_key >> end_bit)) ++end_bit;
        DevBuf<uint32_t> keys(c), keys2(c), idx(c), idx2(c), cnt(c);
        DevBuf<uint8_t> rows(c), tmp(c);
        int rc;
        if ((rc = keys.alloc(ks->n)) || (rc = keys2.alloc(ks->n)) || (rc = idx.alloc(ks->n)) || (rc = idx2.alloc(ks->n)) || (rc = cnt.alloc(ks->n)) ||
            (rc = rows.alloc(ks->n * ks->k)))
            return rc;
        hipLaunchKernelGGL(cid::k_row0_line_ascii, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->k, cid::index_mod(ix), line_shift, bucket_bits, keys.p,
                           idx.p, (uint64_t)ks->n);
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, idx.p, idx2.p, ks->n, 0u, end_bit, st));
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys.p, keys2.p, idx.p, idx2.p, ks->n, 0u, end_bit, st));
        hipLaunchKernelGGL(cid::k_permute_rows, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->counts, idx2.p, ks->k, rows.p, cnt.p, (uint64_t)ks->n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        cid::ctx_free(c, ks->ascii); cid::ctx_free(c, ks->counts);
        ks->ascii = rows.release(); ks->counts = cnt.release();
        return CID_OK;
    }
    DevBuf<uint32_t> on(ks->ctx);
    DevBuf<uint64_t> oc(ks->ctx);
    int rc;
    if ((rc = oc.alloc(ks->n)) || (rc = on.alloc(ks->n))) return rc;
    if ((rc = cid_order_codes_for_index_dev(ks->ctx, ix, ks->codes, ks->counts, ks->n, oc.p, on.p))) return rc;
    HIP_TRY(hipStreamSynchronize(cid::ctx_stream(ks->ctx)));
    cid::ctx_free(ks->ctx, ks->codes); cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = oc.release(); ks->counts = on.release();
    return CID_OK;
}

}  // extern "C"
