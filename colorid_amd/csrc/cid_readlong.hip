// read_id for reads whose k-mer set does not fit a wave's LDS (src/read_id_mt_pe.rs:282-363 takes any read length; the per-read
// set: src/kmer.rs:221-243; the ordered search: read_id_mt_pe.rs:66-165).  Round 1 sorted every window of the batch globally
// (rocPRIM radix sort of (code, window) pairs, 6 passes over 1.8 GB) only to learn, per read, which windows hold the FIRST occurrence
// of their k-mer.  That is a per-read question, and a read's windows fit a workgroup's LDS once they are cut by a hash of the code:
//
//   k_extract_codes      windows -> canonical 2-bit codes (cid_windows.hpp; minimizer indices: -> minimizer codes)
//   k_long_first_flags   work item = (read, bucket b of P): the read's windows whose mixed code falls in bucket b go through ONE LDS
//                        hash table of 4-byte slots (window index << 10 | 10-bit tag; the code itself is read back from the code
//                        array only when a tag matches), atomicMin keeps the smallest window per code; the winners set their bit
//                        in a bitmap over the batch's windows.  P = windows / 16 384, so a pass fills half of a 32 768-slot table;
//                        a pass that overflows anyway is redone on sub-buckets (a further hash bit per level).
//   scan (cid_scan.hpp)  exclusive prefix of the bitmap words' popcounts: rank(w) of any window without a second pass
//   k_readid_slices      the in-order search, one wave per slice of a read, straight from the bitmap and the code array: the flagged
//                        windows in window order ARE the read's k-mers in first-occurrence order (cid_readid.hip); k_readid_combine
//                        adds a read's slices up
//   (k_long_scatter      the same k-mers as lists, for k_readid_list: rows wider than 1 KiB and colour-stripe passes)
//
// A read's windows start at a multiple of 32 in the batch's numbering (the code array has unused gaps): its bitmap words are its own.
// No host round trip between the kernels: nothing is sized by a count only the device knows.
// Byte-string keys (k > 32, or a lower-case base: its case is kept), rows wider than 1 KiB and colour-stripe
// passes keep round 1's path (cid_kmerset.hip: readid_long_sorted) — with this file's lists where the keys pack.
#include <cstring>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"
#include "cid_objects.hpp"
#include "cid_windows.hpp"
#include "cid_scan.hpp"
#include "cid_devbuf.hpp"

namespace cid {

constexpr uint32_t kLongIdxBits = 22, kLongTagBits = 10, kLongEmpty = 0xFFFFFFFFu;
constexpr uint32_t kLongMaxWin = (1u << kLongIdxBits) - 2;       // windows of one read the 4-byte slots can number
constexpr uint32_t kLongSlotsBig = 32768, kLongBlockBig = 1024, kLongBmBig = 4096;   // 128 + 16 KiB of LDS, one workgroup of 16 waves per CU
constexpr uint32_t kLongSlotsSmall = 8192, kLongBlockSmall = 256, kLongBmSmall = 128; // 32.5 KiB: reads of up to kLongSmallWin windows, four workgroups per CU
constexpr uint32_t kLongSmallWin = 4096;
constexpr uint32_t kLongFill = kLongSlotsBig / 2;                 // distinct k-mers a pass over a big table is planned for
constexpr uint32_t kLongMaxLevel = 7;
constexpr uint32_t kSliceWindows = 4096;                          // windows per slice of the search

struct LongItem { uint32_t read, bucket, n_buckets, deal; };   // deal: 1 + index of the read's LongDeal, 0 = the pass reads the code array itself
// A read of several buckets, dealt: its windows are cut into chunks of kDealChunk, every chunk's (code, window) pairs of bucket b lie in
// segment (b, chunk) of `cap` places — pairs[pair_base + (b * n_chunks + chunk) * cap ..], their number in counts[count_base + b * n_chunks + chunk].
struct LongDeal { uint64_t pair_base; uint32_t count_base, n_chunks, cap, read; };
constexpr uint32_t kDealChunk = 16384, kDealFromBuckets = 3;   // (two buckets: 20 kb reads 8.8 ms undealt, 9.2 dealt; three: 40 kb reads 9.6 -> 9.0)
constexpr uint32_t kLongMaxChunks = 256;   // kLongMaxWin / kDealChunk

__device__ __forceinline__ uint64_t long_mix(uint64_t x) {   // a bijection of the 64-bit codes: distinct codes never share all their bits
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// The pre-pass of long reads (more than kDealFromBuckets buckets): every bucket's workgroup used to re-read and re-hash ALL the read's
// windows to find its own — 100 kb reads 11.1 ms, 1 Mb reads 26.9 ms per 150 Mbases against 8.0 at 10 kb.  Here a workgroup takes one
// chunk of a read and deals its (code, window) pairs to the buckets' segments once (LDS counters hand out the places); a bucket's pass
// then reads its own pairs only.  A segment has room for the mean + 25 % + 96 (the mixed codes spread evenly: four standard deviations
// are 12 % at 16 384 / 7 per segment, less at more buckets' smaller means only in absolute terms — hence the + 96); one that
// overflows anyway raises flags[1] and the batch is redone on the sorting path.
constexpr uint32_t kDealBlock = 1024, kDealStageFrom = 24;   // buckets from which a chunk's pairs are grouped in LDS before they are written
__global__ __launch_bounds__(kDealBlock) void k_long_deal(const uint64_t *codes, const uint64_t *wstart, const uint64_t *wend, const LongDeal *deals,
                                                           const uint32_t *chunk_deal, const uint32_t *chunk_no, uint32_t n_chunks_all, uint64_t sentinel,
                                                           uint64_t *pair_code, uint32_t *pair_idx, uint32_t *counts, int *flags) {
    // A chunk's windows are first grouped by bucket in LDS (their numbers only: 64 KiB), then every bucket's pairs leave as one stretch:
    // written straight from the window loop a wave's 64 pairs went to forty different segments, 8 and 4 bytes at a time — 2.5 ms per 150 M
    // windows of 1 Mb reads, the price of 64-byte memory transactions for 12 bytes.
    __shared__ uint32_t s_cnt[256], s_start[257], s_cur[256];
    __shared__ uint32_t s_idx[kDealChunk];
    for (uint32_t ci = blockIdx.x; ci < n_chunks_all; ci += gridDim.x) {
        const LongDeal d = deals[chunk_deal[ci]];
        const uint32_t j = chunk_no[ci];
        const uint64_t w0 = wstart[d.read];
        const uint32_t nw = (uint32_t)(wend[d.read] - w0);
        const uint32_t P = (nw + kLongFill - 1) / kLongFill;   // (as the host counted them: <= 256)
        const uint32_t a = j * kDealChunk, b = a + kDealChunk < nw ? a + kDealChunk : nw;
        if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        if (P < kDealStageFrom) {   // (uniform) few buckets: a wave's pairs fall into few segments as they are — 100 kb reads 9.4 ms so, 9.9 staged
            for (uint32_t w = a + threadIdx.x; w < b; w += kDealBlock) {
                const uint64_t code = codes[w0 + w];
                if (code >= sentinel) continue;
                const uint32_t bk = (uint32_t)(((long_mix(code) >> 32) * P) >> 32);
                const uint32_t at = atomicAdd(&s_cnt[bk], 1u);
                if (at < d.cap) {
                    const uint64_t o = d.pair_base + ((uint64_t)bk * d.n_chunks + j) * d.cap + at;
                    pair_code[o] = code;
                    pair_idx[o] = w;
                } else atomicOr(&flags[1], 1);
            }
            __syncthreads();
            if (threadIdx.x < P) counts[d.count_base + threadIdx.x * d.n_chunks + j] = s_cnt[threadIdx.x] < d.cap ? s_cnt[threadIdx.x] : d.cap;
            __syncthreads();
            continue;
        }
        for (uint32_t w = a + threadIdx.x; w < b; w += kDealBlock) {
            const uint64_t code = codes[w0 + w];
            if (code >= sentinel) continue;
            atomicAdd(&s_cnt[(uint32_t)(((long_mix(code) >> 32) * P) >> 32)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {   // exclusive prefix over the (<= 256) bucket counts: four per lane
            uint32_t c4[4], sum = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { c4[q] = s_cnt[threadIdx.x * 4 + q]; sum += c4[q]; }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o, 64);
                if ((int)threadIdx.x >= o) inc += up;
            }
            uint32_t base = inc - sum;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { s_start[threadIdx.x * 4 + q] = base; s_cur[threadIdx.x * 4 + q] = base; base += c4[q]; }
            if (threadIdx.x == 63) s_start[256] = base;
        }
        __syncthreads();
        for (uint32_t w = a + threadIdx.x; w < b; w += kDealBlock) {
            const uint64_t code = codes[w0 + w];
            if (code >= sentinel) continue;
            s_idx[atomicAdd(&s_cur[(uint32_t)(((long_mix(code) >> 32) * P) >> 32)], 1u)] = w;
        }
        __syncthreads();
        const uint32_t total = s_start[256];
        bool over = false;
        for (uint32_t p = threadIdx.x; p < total; p += kDealBlock) {
            uint32_t lo = 0, hi = 256;   // the bucket of place p: s_start[lo] <= p < s_start[lo + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_start[mid] <= p) lo = mid; else hi = mid;
            }
            const uint32_t at = p - s_start[lo];
            if (at < d.cap) {
                const uint64_t o = d.pair_base + ((uint64_t)lo * d.n_chunks + j) * d.cap + at;
                const uint32_t w = s_idx[p];
                pair_code[o] = codes[w0 + w];
                pair_idx[o] = w;
            } else over = true;
        }
        if (over) atomicOr(&flags[1], 1);
        if (threadIdx.x < P) counts[d.count_base + threadIdx.x * d.n_chunks + j] = s_cnt[threadIdx.x] < d.cap ? s_cnt[threadIdx.x] : d.cap;
        __syncthreads();
    }
}

// One workgroup per work item at a time; XCD x walks the x-th eighth of the items so that the passes over one read's codes meet in
// one L2 (workgroups are dealt to the XCDs in turn).  gridDim.x is a multiple of 8.
__global__ void k_long_first_flags(const uint64_t *codes, const uint64_t *wstart, const uint64_t *wend, const LongItem *items, uint32_t n_items, uint64_t sentinel,
                                   uint32_t slots, uint32_t bm_words, uint32_t *bitmap, int *flags, const LongDeal *deals, const uint64_t *pair_code,
                                   const uint32_t *pair_idx, const uint32_t *deal_counts) {
    extern __shared__ uint32_t table[];   // slots, then bm_words: the stretch of the read's bitmap being put together
    __shared__ int s_over;
    __shared__ uint32_t s_pref[kLongMaxChunks + 1];   // a dealt bucket: pairs in the segments before chunk j
    uint32_t *bm = table + slots;
    const uint32_t max_slots = slots, bm_bits = bm_words * 32u;
    const uint32_t chunk = (n_items + 7u) / 8u;
    for (uint32_t it = blockIdx.x >> 3; it < chunk; it += gridDim.x >> 3) {
        const uint32_t item = (blockIdx.x & 7u) * chunk + it;
        if (item >= n_items) break;
        const LongItem im = items[item];
        const uint64_t w0 = wstart[im.read];
        const uint32_t nw = (uint32_t)(wend[im.read] - w0);   // (wstart[read + 1] lies beyond the padding)
        const uint64_t *rc = codes + w0;
        // a short read takes a corner of the table: clearing and sweeping it is what a pass costs beyond its inserts
        slots = 1024;
        while (slots < max_slots && slots < 2u * nw) slots <<= 1;
        const uint32_t mask = slots - 1;
        uint32_t level = 0;
        for (uint32_t sub = 0; sub < (1u << level); ++sub) {
            for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) table[s] = kLongEmpty;
            if (threadIdx.x == 0) s_over = 0;
            __syncthreads();
            auto insert = [&](uint64_t code, uint32_t w, uint64_t h) {
                const uint32_t tag = (uint32_t)(h >> 15) & ((1u << kLongTagBits) - 1u);
                const uint32_t mine = (w << kLongTagBits) | tag;
                uint32_t pos = (uint32_t)h & mask;
                for (uint32_t probes = 0;; ++probes) {
                    uint32_t cur = __hip_atomic_load(&table[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == kLongEmpty) {
                        cur = atomicCAS(&table[pos], kLongEmpty, mine);
                        if (cur == kLongEmpty) break;
                    }
                    if ((cur & ((1u << kLongTagBits) - 1u)) == tag && rc[cur >> kLongTagBits] == code) {   // the slot is this k-mer's
                        atomicMin(&table[pos], mine);
                        break;
                    }
                    pos = (pos + 1) & mask;
                    if (probes >= slots / 4) { s_over = 1; break; }   // a crowded table: the pass is redone on sub-buckets
                }
            };
            if (im.deal) {   // the bucket's own pairs: its segments (one per chunk of the read, a few hundred pairs each) walked as ONE index space —
                // chunk after chunk, a segment kept a quarter of the workgroup busy and every chunk paid a memory round trip for its count
                // (1 Mb reads: 61 chunks, 187 us per bucket against 23 us for a whole 10 kb read)
                const LongDeal d = deals[im.deal - 1];
                if (sub == 0 || level) {   // (the prefix is the same in every pass; the table's clearing barrier above separates the passes)
                    for (uint32_t j = threadIdx.x; j < d.n_chunks; j += blockDim.x) s_pref[j + 1] = deal_counts[d.count_base + im.bucket * d.n_chunks + j];
                    if (threadIdx.x == 0) s_pref[0] = 0;
                    __syncthreads();
                    if (threadIdx.x == 0)
                        for (uint32_t j = 0; j < d.n_chunks; ++j) s_pref[j + 1] += s_pref[j];
                    __syncthreads();
                }
                const uint32_t total = s_pref[d.n_chunks];
                const uint64_t o0 = d.pair_base + (uint64_t)im.bucket * d.n_chunks * d.cap;
                for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
                    uint32_t lo = 0, hi = d.n_chunks;   // the chunk whose segment holds pair t: s_pref[lo] <= t < s_pref[lo + 1]
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= t) lo = mid; else hi = mid;
                    }
                    const uint64_t o = o0 + (uint64_t)lo * d.cap + (t - s_pref[lo]);
                    const uint64_t code = pair_code[o];
                    const uint64_t h = long_mix(code);
                    if (level && ((uint32_t)(h >> 25) & ((1u << level) - 1u)) != sub) continue;
                    insert(code, pair_idx[o], h);
                }
            } else
            for (uint32_t w = threadIdx.x; w < nw; w += blockDim.x) {
                const uint64_t code = rc[w];
                if (code >= sentinel) continue;   // (no k-mer in this window)
                const uint64_t h = long_mix(code);
                if (im.n_buckets > 1 && (uint32_t)(((h >> 32) * im.n_buckets) >> 32) != im.bucket) continue;
                if (level && ((uint32_t)(h >> 25) & ((1u << level) - 1u)) != sub) continue;
                insert(code, w, h);
            }
            __syncthreads();
            if (s_over) {   // (workgroup-uniform)
                if (level == kLongMaxLevel) {
                    if (threadIdx.x == 0) atomicOr(&flags[1], 1);   // the host redoes the batch on the sorting path
                    break;
                }
                ++level;
                sub = ~0u;   // from the first sub-bucket of the finer level (bits set so far stay right: they are first occurrences)
                __syncthreads();
                continue;
            }
            // The winners' bits: put together in LDS, a stretch of bm_bits windows at a time, and written out as whole words (a read's
            // windows start at a multiple of 32, so its words are its own).  One global atomicOr per BIT took 4.4 of this kernel's
            // 5.4 ms on 150 Mbases of 10 kb reads: the atomics of a read all fall into its dozen of 128-byte lines.
            const bool own_words = im.n_buckets == 1 && level == 0;   // else other passes add to the same words
            for (uint32_t c0 = 0; c0 < nw; c0 += bm_bits) {
                for (uint32_t i = threadIdx.x; i < bm_words; i += blockDim.x) bm[i] = 0;
                __syncthreads();
                for (uint32_t s = threadIdx.x; s < slots; s += blockDim.x) {
                    const uint32_t cur = table[s];
                    if (cur != kLongEmpty) {
                        const uint32_t w = (cur >> kLongTagBits) - c0;
                        if (w < bm_bits) atomicOr(&bm[w >> 5], 1u << (w & 31u));
                    }
                }
                __syncthreads();
                const uint32_t words = (nw - c0 + 31u) / 32u < bm_words ? (nw - c0 + 31u) / 32u : bm_words;
                uint32_t *out = bitmap + ((w0 + c0) >> 5);
                for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
                    const uint32_t v = bm[i];
                    if (own_words) out[i] = v;
                    else if (v) atomicOr(&out[i], v);
                }
                __syncthreads();
            }
        }
    }
}

struct PopcIn {
    const uint32_t *bitmap;
    __device__ uint64_t operator()(uint64_t i) const { return (uint64_t)__popc(bitmap[i]); }
};
struct PrefixOut {
    uint32_t *prefix;
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t) const { prefix[i] = (uint32_t)excl; }
};

__device__ __forceinline__ uint32_t long_rank(const uint32_t *bitmap, const uint32_t *prefix, uint64_t w) {
    return prefix[w >> 5] + (uint32_t)__popc(bitmap[w >> 5] & ((1u << (w & 31u)) - 1u));
}
__global__ void k_long_scatter(const uint64_t *codes, const uint32_t *bitmap, const uint32_t *prefix, uint64_t *list, uint64_t W) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const uint32_t word = bitmap[w >> 5];
    if ((word >> (w & 31u)) & 1u) list[prefix[w >> 5] + (uint32_t)__popc(word & ((1u << (w & 31u)) - 1u))] = codes[w];
}
__global__ void k_long_list_starts(const uint64_t *wstart, const uint32_t *bitmap, const uint32_t *prefix, uint64_t *list_start, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) list_start[r] = long_rank(bitmap, prefix, wstart[r]);
}
// rows of the reads this path answers without a search (too short: the first mate has no window)
__global__ void k_long_short_rows(const uint8_t *status, uint32_t n_reads, uint32_t C, uint32_t *report, uint32_t *n_kmers) {
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_reads || status[r] != 1) return;
    for (uint32_t c = threadIdx.x & 63u; c <= C; c += 64u) report[(uint64_t)r * (C + 1) + c] = 0;
    if ((threadIdx.x & 63u) == 0) n_kmers[r] = 0;
}

int readid_long(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, const uint64_t *read_seq0, size_t n_reads,
                uint32_t stride_d, uint32_t start_sample, const uint8_t *route, bool clear_wide, uint32_t *d_report, uint32_t *d_n_kmers,
                uint8_t *d_status, const StripePass &sp) {
    const uint32_t k = index_k(ix);
    if (k > 32 || !c->tune.readid_long_lds)
        return readid_long_sorted(c, ix, d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample, route, clear_wide, d_report, d_n_kmers, d_status, sp);
    hipStream_t st = ctx_stream(c);
    const uint32_t msz = index_m_size(ix);
    const uint32_t key_len = msz ? msz : k;
    const uint64_t sentinel_k = k < 32 ? (1ull << (2 * k)) : ~0ull;
    const uint64_t sentinel = key_len < 32 ? (1ull << (2 * key_len)) : ~0ull;
    const uint32_t C = index_n_colors(ix), rs = index_rs(ix), n_hash = index_n_hash(ix);
    const bool own_search = rs <= 128 && !sp.on();   // k_readid_slices; else the lists feed k_readid_list
    const bool cut = start_sample <= 64;              // a later slice gathers the first S k-mers again
    // windows are numbered read by read, mate by mate
    std::vector<uint64_t> wstart(n_reads + 1, 0), wend(n_reads, 0);
    std::vector<uint8_t> status(n_reads, 0);
    std::vector<Segment> segs;
    std::vector<LongItem> items_small, items_big;
    std::vector<LongDeal> deals;
    std::vector<uint32_t> chunk_deal, chunk_no;   // the chunks of the dealt reads: which deal, which chunk of it
    uint64_t n_pairs = 0;
    uint32_t n_deal_counts = 0;
    std::vector<ReadSlice> slices;
    std::vector<ReadCombine> combs;
    const uint32_t seg_win = kSegWindows / stride_d ? kSegWindows / stride_d : 1;
    uint64_t W = 0;
    for (size_t r = 0; r < n_reads; ++r) {
        wstart[r] = W;
        if (route && !route[r]) { status[r] = 2; continue; }
        W = (W + 31) & ~(uint64_t)31;   // (its bitmap words are its own)
        wstart[r] = W;
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
        const uint64_t win = W - wstart[r];
        wend[r] = W;
        if (win > kLongMaxWin)   // (a 4 Mb read: the slots number 2^22 windows)
            return readid_long_sorted(c, ix, d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample, route, clear_wide, d_report, d_n_kmers,
                                      d_status, sp);
        if (win <= kLongSmallWin) items_small.push_back(LongItem{(uint32_t)r, 0u, 1u, 0u});
        else {
            const uint32_t P = (uint32_t)((win + kLongFill - 1) / kLongFill);
            uint32_t deal = 0;
            if (P >= kDealFromBuckets && c->tune.readid_long_deal) {
                const uint32_t nc = (uint32_t)((win + kDealChunk - 1) / kDealChunk);
                const uint32_t cap = kDealChunk / P + kDealChunk / P / 4 + 96;
                deals.push_back(LongDeal{n_pairs, n_deal_counts, nc, cap, (uint32_t)r});
                deal = (uint32_t)deals.size();
                n_pairs += (uint64_t)P * nc * cap;
                n_deal_counts += P * nc;
                for (uint32_t j = 0; j < nc; ++j) { chunk_deal.push_back(deal - 1); chunk_no.push_back(j); }
            }
            for (uint32_t b = 0; b < P; ++b) items_big.push_back(LongItem{(uint32_t)r, b, P, deal});
        }
        if (own_search) {
            const uint32_t n_sl = cut ? (uint32_t)((win + kSliceWindows - 1) / kSliceWindows) : 1u;
            if (n_sl > 1) combs.push_back(ReadCombine{(uint32_t)r, (uint32_t)slices.size(), n_sl, 0u});
            for (uint32_t j = 0; j < n_sl; ++j) {
                const uint64_t a = wstart[r] + (uint64_t)j * kSliceWindows;
                const uint64_t b = n_sl == 1 ? W : (a + kSliceWindows < W ? a + kSliceWindows : W);
                slices.push_back(ReadSlice{(uint32_t)r, (uint32_t)a, (uint32_t)b, j | (n_sl > 1 ? 0x80000000u : 0u)});
            }
        }
    }
    wstart[n_reads] = W;
    if (W >= (1ull << 32) - 64) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    if (n_reads >= (1ull << 31)) return fail(CID_ERR_UNSUPPORTED, "more than 2^31 reads in one batch");
    const size_t C1 = (size_t)C + 1;
    const uint64_t n_words = W / 32 + 1;   // (rank(W) reads the word after the last window's)
    // the host-made arrays travel as ONE block, before the first kernel, through the ctx's pinned arena when they fit: a copy out of
    // pageable memory makes the runtime wait for the stream, and one issued between two kernels stalls the launch of the second
    struct Part { const void *src; size_t bytes, off; };
    Part parts[10] = {{wstart.data(), (n_reads + 1) * 8, 0}, {wend.data(), n_reads * 8, 0}, {segs.data(), segs.size() * sizeof(Segment), 0},
                      {items_small.data(), items_small.size() * sizeof(LongItem), 0}, {items_big.data(), items_big.size() * sizeof(LongItem), 0},
                      {slices.data(), slices.size() * sizeof(ReadSlice), 0}, {combs.data(), combs.size() * sizeof(ReadCombine), 0},
                      {deals.data(), deals.size() * sizeof(LongDeal), 0}, {chunk_deal.data(), chunk_deal.size() * 4, 0}, {chunk_no.data(), chunk_no.size() * 4, 0}};
    size_t meta_bytes = (n_reads + 15) & ~(size_t)15;   // the status bytes lead the block
    for (Part &pt : parts) { pt.off = meta_bytes; meta_bytes += (pt.bytes + 15) & ~(size_t)15; }
    DevBuf<uint64_t> d_codes(c), d_list(c), d_lstart(c), d_scan(c);
    DevBuf<uint32_t> d_bitmap(c), d_prefix(c), d_partial(c), d_pair_idx(c), d_deal_counts(c);
    DevBuf<uint64_t> d_pair_code(c);
    DevBuf<uint8_t> d_meta(c);
    DevBuf<int> d_flags(c);
    int rc;
    if ((rc = d_meta.alloc(meta_bytes + 16)) || (rc = d_codes.alloc(W + 1)) || (rc = d_scan.alloc(scan_state_words(n_words))) ||
        (rc = d_bitmap.alloc(n_words)) || (rc = d_prefix.alloc(n_words)) ||
        (rc = d_partial.alloc(combs.empty() ? 1 : slices.size() * (C1 + 1))) || (rc = d_flags.alloc(4)) ||
        (rc = d_pair_code.alloc(n_pairs)) || (rc = d_pair_idx.alloc(n_pairs)) || (rc = d_deal_counts.alloc(n_deal_counts)))
        return rc;
    const uint64_t *d_wstart = reinterpret_cast<const uint64_t *>(d_meta.p + parts[0].off), *d_wend = reinterpret_cast<const uint64_t *>(d_meta.p + parts[1].off);
    const Segment *d_segs = reinterpret_cast<const Segment *>(d_meta.p + parts[2].off);
    const LongItem *d_items_small = reinterpret_cast<const LongItem *>(d_meta.p + parts[3].off), *d_items_big = reinterpret_cast<const LongItem *>(d_meta.p + parts[4].off);
    const ReadSlice *d_slices = reinterpret_cast<const ReadSlice *>(d_meta.p + parts[5].off);
    const ReadCombine *d_combs = reinterpret_cast<const ReadCombine *>(d_meta.p + parts[6].off);
    const LongDeal *d_deals = reinterpret_cast<const LongDeal *>(d_meta.p + parts[7].off);
    const uint32_t *d_chunk_deal = reinterpret_cast<const uint32_t *>(d_meta.p + parts[8].off), *d_chunk_no = reinterpret_cast<const uint32_t *>(d_meta.p + parts[9].off);
    if (uint8_t *pin = pin_reserve(c, meta_bytes + 64)) {
        HIP_TRY(hipStreamSynchronize(st));   // (the arena may still feed an earlier copy)
        memcpy(pin, status.data(), n_reads);
        for (const Part &pt : parts) if (pt.bytes) memcpy(pin + pt.off, pt.src, pt.bytes);
        HIP_TRY(hipMemcpyAsync(d_meta.p, pin, meta_bytes, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(d_meta.p, status.data(), n_reads, hipMemcpyHostToDevice, st));
        for (const Part &pt : parts) if (pt.bytes) HIP_TRY(hipMemcpyAsync(d_meta.p + pt.off, pt.src, pt.bytes, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(d_status, d_meta.p, n_reads, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(d_flags.p, 0, 16, st));
    HIP_TRY(hipMemsetAsync(d_bitmap.p, 0, n_words * 4, st));
    int h_flags[4] = {0, 0, 0, 0};
    if (W) {
        constexpr uint32_t kBytes = kSegWindows + 32 + 96;
        const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
        unsigned grid = (unsigned)((segs.size() + 3) / 4);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, d_bases, d_segs, (uint32_t)segs.size(), k, 1, sentinel_k, d_codes.p,
                           d_flags.p, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr, KeyFor{});
        if (msz) hipLaunchKernelGGL(k_codes_to_minimizers, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, (uint64_t)W, k, msz, sentinel_k, sentinel);
        const unsigned n_cu = (unsigned)ctx_n_cu(c);
        if (!chunk_deal.empty()) {
            unsigned g = (unsigned)chunk_deal.size();
            if (g > n_cu * 8u) g = n_cu * 8u;
            hipLaunchKernelGGL(k_long_deal, dim3(g), dim3(kDealBlock), 0, st, d_codes.p, d_wstart, d_wend, d_deals, d_chunk_deal, d_chunk_no, (uint32_t)chunk_deal.size(),
                               sentinel, d_pair_code.p, d_pair_idx.p, d_deal_counts.p, d_flags.p);
        }
        if (!items_small.empty()) {
            unsigned g = n_cu * 4u;   // four 32-KiB workgroups per CU
            g = (g + 7u) & ~7u;
            hipLaunchKernelGGL(k_long_first_flags, dim3(g), dim3(kLongBlockSmall), (kLongSlotsSmall + kLongBmSmall) * 4, st, d_codes.p, d_wstart, d_wend, d_items_small,
                               (uint32_t)items_small.size(), sentinel, kLongSlotsSmall, kLongBmSmall, d_bitmap.p, d_flags.p, d_deals, d_pair_code.p, d_pair_idx.p,
                               d_deal_counts.p);
        }
        if (!items_big.empty()) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_long_first_flags), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)((kLongSlotsBig + kLongBmBig) * 4)));
            unsigned g = (n_cu + 7u) & ~7u;
            hipLaunchKernelGGL(k_long_first_flags, dim3(g), dim3(kLongBlockBig), (kLongSlotsBig + kLongBmBig) * 4, st, d_codes.p, d_wstart, d_wend,
                               d_items_big, (uint32_t)items_big.size(), sentinel, kLongSlotsBig, kLongBmBig, d_bitmap.p, d_flags.p, d_deals, d_pair_code.p,
                               d_pair_idx.p, d_deal_counts.p);
        }
        HIP_TRY(hipGetLastError());
        if (!own_search) {   // wide rows and stripe passes add into rows in place: what would send the batch to the sorting path (see the
            HIP_TRY(hipMemcpyAsync(h_flags, d_flags.p, 8, hipMemcpyDeviceToHost, st));   // end of this function) must be known before anything is counted
            HIP_TRY(hipStreamSynchronize(st));
            if (h_flags[0] || h_flags[1])
                return readid_long_sorted(c, ix, d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample, route, clear_wide, d_report, d_n_kmers,
                                          d_status, sp);
        }
    }
    HIP_TRY(scan_launch(PopcIn{d_bitmap.p}, PrefixOut{d_prefix.p}, n_words, d_scan.p, st));
    if (!own_search) {   // k_readid_list walks lists
        if ((rc = d_list.alloc(W + 1)) || (rc = d_lstart.alloc(n_reads + 1))) return rc;
        if (W) hipLaunchKernelGGL(k_long_scatter, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, d_bitmap.p, d_prefix.p, d_list.p, (uint64_t)W);
        hipLaunchKernelGGL(k_long_list_starts, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, d_wstart, d_bitmap.p, d_prefix.p, d_lstart.p,
                           (uint32_t)n_reads);
        HIP_TRY(hipGetLastError());
    }
    const uint32_t hist_pad = rs > 128 ? 4u * rs : (uint32_t)((C1 + 3) & ~(size_t)3);
    const uint32_t wave_bytes = (uint32_t)((4ull * kWave * n_hash + 4ull * hist_pad + 15) & ~15ull);
    if ((size_t)(kBlock / kWave) * wave_bytes > 160u * 1024u) return fail(CID_ERR_UNSUPPORTED, "LDS need exceeds 160 KiB");
    if (own_search) {
        ReadIdSliceParams p{};
        p.mat = index_matrix(ix); p.rs = rs; p.w64 = (C + 63) / 64; p.n_colors = C; p.n_hash = n_hash; p.k = key_len; p.mod = index_mod(ix);
        p.codes = d_codes.p; p.wstart = d_wstart; p.wend = d_wend; p.bitmap = d_bitmap.p; p.word_prefix = d_prefix.p;
        p.slices = d_slices; p.n_slices = (uint32_t)slices.size(); p.start_sample = start_sample;
        p.hist_pad = hist_pad; p.wave_bytes = wave_bytes;
        p.report = d_report; p.n_kmers = d_n_kmers; p.partial = d_partial.p;
        uint64_t grid = (slices.size() + 3) / 4;
        const uint64_t cap = (uint64_t)ctx_n_cu(c) * 32;
        if (grid > cap) grid = cap;
        HIP_TRY(launch_readid_slices(p, (int)grid, st));
        HIP_TRY(launch_readid_combine(d_combs, (uint32_t)combs.size(), d_partial.p, C, d_report, st));
        hipLaunchKernelGGL(k_long_short_rows, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, d_status, (uint32_t)n_reads, C, d_report, d_n_kmers);
        HIP_TRY(hipGetLastError());
    } else {
        ReadIdListParams p{};
        p.mat = index_matrix(ix); p.rs = rs; p.w64 = (C + 63) / 64; p.n_colors = C; p.n_hash = n_hash; p.k = key_len; p.mod = index_mod(ix);
        p.list_codes = d_list.p; p.list_start = d_lstart.p; p.n_reads = n_reads; p.start_sample = start_sample;
        p.bases = nullptr; p.upper = msz != 0;
        p.hist_pad = hist_pad; p.wave_bytes = wave_bytes;
        if (rs > 128 && clear_wide && !sp.on()) HIP_TRY(hipMemsetAsync(d_report, 0, n_reads * C1 * 4, st));   // wide rows count in place
        p.zero_acc = sp.zero_acc; p.zero_in = sp.zero_in; p.zero_start = sp.zero_start;   // a colour stripe's pass: the caller zeroed the report
        p.colour_base = sp.colour_base; p.report_width = sp.report_width; p.write_nohits = sp.write_nohits;
        p.report = d_report; p.n_kmers = d_n_kmers; p.status = d_status;
        uint64_t grid = (n_reads + 3) / 4;
        if (grid > 4096) grid = 4096;
        HIP_TRY(launch_readid_list(p, (int)grid, st));
    }
    // The one wait of the call, at its end: the host arrays above leave scope, and two facts only the kernels know decide whether the
    // batch has to be redone on the sorting path — a lower-case base among the long reads, or a hash table that overflowed seven levels deep.
    HIP_TRY(hipMemcpyAsync(h_flags, d_flags.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_flags[0] || h_flags[1])
        return readid_long_sorted(c, ix, d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample, route, clear_wide, d_report, d_n_kmers, d_status, sp);
    return CID_OK;
}

}  // namespace cid
