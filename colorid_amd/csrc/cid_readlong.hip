// read_id for reads whose k-mer set does not fit a wave's LDS (src/read_id_mt_pe.rs:282-363 takes any read length; the per-read
// set: src/kmer.rs:221-243; the ordered search: read_id_mt_pe.rs:66-165).  Round 1 sorted every window of the batch globally
// (rocPRIM radix sort of (code, window) pairs, 6 passes over 1.8 GB) only to learn, per read, which windows hold the FIRST occurrence
// of their k-mer.  That is a per-read question, and a read's windows fit a workgroup's LDS once they are cut by a hash of the code:
//
//   k_long_route         (callers with mixed batches) which reads are this path's: at least `long_from` bases
//   k_long_plan / _emit  the plan of the batch, on the device (round 6): classes, window numbering, every work list below — from
//                        seq_off / read_seq0 in HBM; the host reads back 64 bytes of totals to size the lists
//   k_long_fused         reads whose bases fit the LDS beside the table: bases -> 2-bit fields in LDS -> rolling canonical codes -> LDS hash
//                        table of 4-byte slots (window index << 10 | 10-bit tag), atomicMin keeps the smallest window per code -> the
//                        winners' bits as whole words of the first-occurrence bitmap; the codes leave for k_readid_slices.  Four
//                        shapes: a wave per read (<= 1 024 windows: 2 048 slots), 256 threads (<= 4 096: 8 192 slots), 1 024 threads
//                        (<= 16 384 windows, every 10 kb read: 32 768 slots), and that table filled once per hash bucket for reads of
//                        up to 49 152 windows (two or three passes over the read's bases in LDS)
//   k_extract_codes      (longer reads, strides that stretch a read beyond the fused kernel's LDS) windows -> canonical 2-bit codes
//   k_long_deal          reads of three buckets and more: their (code, window) pairs dealt to the buckets' segments once, and a bit
//                        of the bitmap for every window that holds a k-mer
//   k_long_first_flags   work item = (read, bucket b of P): the read's windows whose mixed code falls in bucket b through ONE table;
//                        P = windows / 16 384; a pass that overflows anyway is redone on sub-buckets (a further hash bit per level).
//                        A dealt bucket's pass takes the LATER occurrences' bits out of the bitmap; the others put the first ones' in
//   scan (cid_scan.hpp)  exclusive prefix of the bitmap words' popcounts: rank(w) of any window without a second pass
//   k_readid_slices      the in-order search, one wave per slice of a read, straight from the bitmap and the code array: the flagged
//                        windows in window order ARE the read's k-mers in first-occurrence order (cid_readid.hip); k_readid_combine
//                        adds a read's slices up
//   (k_long_scatter      the same k-mers as lists, for k_readid_list: rows wider than 1 KiB and colour-stripe passes)
//
// A read's windows start at a multiple of 32 in the batch's numbering (the code array has unused gaps): its bitmap words are its own.
// PER READ (round 6; until round 5 per batch): a read with a lower-case base (its case is kept: byte-string keys), of
// more than 2^22 - 2 windows, or whose table crowded seven levels deep is marked in redo[] and goes — alone — through round 1's sorting
// path (cid_kmerset_cold.hip: readid_long_sorted), which also keeps k > 32.
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
constexpr uint32_t kLongSlotsBig = 32768, kLongBlockBig = 1024, kLongBmBig = 6144;   // 128 + 24 KiB of LDS, one workgroup of 16 waves per CU
constexpr uint32_t kDealQueue = kLongBmBig / (kLongBlockBig / 64);                    // a wave's queue of a dealt bucket's pass (the bitmap stretch is idle there)
constexpr uint32_t kLongSlotsSmall = 8192, kLongBlockSmall = 256, kLongBmSmall = 128; // 32.5 KiB: reads of up to kLongSmallWin windows, four workgroups per CU
constexpr uint32_t kLongSmallWin = 4096;
constexpr uint32_t kLongFill = kLongSlotsBig / 2;                 // distinct k-mers a pass over a big table is planned for
constexpr uint32_t kLongMaxLevel = 7;
constexpr uint32_t kSliceWindows = 4096;                          // windows per slice of the search

// One pass of k_long_first_flags: bucket `bucket` of `n_buckets` of read `read`.  deal: 1 + index of the read's LongDeal, 0 = the pass reads the
// code array itself.  Everything the pass needs to start lies in the item (the read's windows, the dealt bucket's segments): looked up
// one after the other — item, window range, deal, counts — a bucket's pass began with four trips to memory and nothing else to do.
struct LongItem { uint32_t read, bucket, n_buckets, deal, nw, n_chunks, cap, count_base; uint64_t w0, pair_base; };
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

// k_long_fused's table hash: the slot and the tag of a code.  Nothing rests on it being injective (a matching tag is followed by a comparison
// of the codes), so two 32-bit multiplies do where long_mix spends two 64-bit ones.
__device__ __forceinline__ uint32_t long_mix32(uint64_t code) {
    uint32_t x = (uint32_t)code ^ ((uint32_t)(code >> 32) * 0x9E3779B1u);
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

// The pre-pass of long reads (kDealFromBuckets buckets and more): every bucket's workgroup used to re-read and re-hash ALL the read's
// windows to find its own — 100 kb reads 11.1 ms, 1 Mb reads 26.9 ms per 150 Mbases against 8.0 at 10 kb.  Here a workgroup takes one
// chunk of a read and deals its (code, window) pairs to the buckets' segments once (LDS counters hand out the places); a bucket's pass
// then reads its own pairs only.  A segment has room for the mean + 25 % + 96 (the mixed codes spread evenly: four standard deviations
// are 12 % at 16 384 / 7 per segment, less at more buckets' smaller means only in absolute terms — hence the + 96); one that
// overflows anyway marks its read (redo[read], flags[1]): that read alone is redone on the sorting path.
// The chunk's words of the read's bitmap are written here as well, a bit for every window that holds a k-mer: the buckets' passes
// (k_long_first_flags) take the bits of the later occurrences OUT instead of putting the first occurrences' in.
constexpr uint32_t kDealBlock = 1024;
__global__ __launch_bounds__(kDealBlock) void k_long_deal(const uint64_t *codes, const uint64_t *wstart, const uint64_t *wend, const LongDeal *deals,
                                                           const uint32_t *chunk_deal, const uint32_t *chunk_no, uint32_t n_chunks_all, uint64_t sentinel,
                                                           uint64_t *pair_code, uint32_t *pair_idx, uint32_t *counts, int *flags, uint8_t *redo,
                                                           uint32_t *bitmap) {
    __shared__ uint32_t s_cnt[256];
    for (uint32_t ci = blockIdx.x; ci < n_chunks_all; ci += gridDim.x) {
        const LongDeal d = deals[chunk_deal[ci]];
        const uint32_t j = chunk_no[ci];
        const uint64_t w0 = wstart[d.read];
        const uint32_t nw = (uint32_t)(wend[d.read] - w0);
        const uint32_t P = (nw + kLongFill - 1) / kLongFill;   // (as the host counted them: <= 256)
        const uint32_t a = j * kDealChunk, b = a + kDealChunk < nw ? a + kDealChunk : nw;
        if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        // A thread's sixteen windows are ASKED FOR together (every load unconditional, from an address that exists), then counted together,
        // then written: one window per trip — load, hash, count, write — was sixteen trips to memory in a row per chunk (and forty-eight
        // in the form that grouped a chunk's pairs by bucket in LDS before writing them, which reads of 24 buckets and more took until
        // round 6: 1.62 ms per 150 Mbases of 1 Mb reads; grouped with its loads batched 1.45; this, 1.16 — a wave's 64 pairs fall into
        // forty segments, but the writes are on their way together).
        constexpr uint32_t kPer = kDealChunk / kDealBlock, kNoBucket = 0xFFFFu;
        uint64_t c[kPer];
        uint32_t bk[kPer], at[kPer];   // the window's bucket (kNoBucket: no k-mer there) and its place in the bucket's segment
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint32_t w = a + i * kDealBlock + threadIdx.x;
            c[i] = codes[w0 + (w < b ? w : a)];
        }
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint32_t w = a + i * kDealBlock + threadIdx.x;
            const bool valid = w < b && c[i] < sentinel;
            const uint64_t vm = __ballot(valid);   // (a is a multiple of 64 and so is every wave's first window)
            if ((threadIdx.x & 31u) == 0 && w < b) bitmap[(w0 + w) >> 5] = (uint32_t)(vm >> (threadIdx.x & 32u));
            bk[i] = valid ? (uint32_t)(((long_mix(c[i]) >> 32) * P) >> 32) : kNoBucket;
        }
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) at[i] = bk[i] != kNoBucket ? atomicAdd(&s_cnt[bk[i]], 1u) : 0u;
        bool over = false;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            if (bk[i] == kNoBucket) continue;
            if (at[i] < d.cap) {
                const uint64_t o = d.pair_base + ((uint64_t)bk[i] * d.n_chunks + j) * d.cap + at[i];
                pair_code[o] = c[i];
                pair_idx[o] = a + i * kDealBlock + threadIdx.x;
            } else over = true;
        }
        __syncthreads();
        if (over) { redo[d.read] = 1; atomicOr(&flags[1], 1); }
        if (threadIdx.x < P) counts[d.count_base + threadIdx.x * d.n_chunks + j] = s_cnt[threadIdx.x] < d.cap ? s_cnt[threadIdx.x] : d.cap;
        __syncthreads();
    }
}

#ifdef CID_LONG_PROF
__device__ unsigned long long g_ff_prof[8];   // k_long_first_flags: cycles per phase, summed over the workgroups (thread 0's clock)
#define FF_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_ff_prof[i], t_ - ff_t); ff_t = t_; } } while (0)
#else
#define FF_MARK(i) do { } while (0)
#endif
// One workgroup per work item at a time; XCD x walks the x-th eighth of the items so that the passes over one read's codes meet in
// one L2 (workgroups are dealt to the XCDs in turn).  gridDim.x is a multiple of 8.
__global__ void k_long_first_flags(const uint64_t *codes, const uint64_t *wstart, const uint64_t *wend, const LongItem *items, uint32_t n_items, uint64_t sentinel,
                                   uint32_t slots, uint32_t bm_words, uint32_t *bitmap, int *flags, const LongDeal *deals, const uint64_t *pair_code,
                                   const uint32_t *pair_idx, const uint32_t *deal_counts, uint8_t *redo) {
    (void)wstart; (void)wend; (void)deals;   // (the items carry what these held)
    extern __shared__ __align__(16) uint32_t table[];   // slots, then bm_words: the stretch of the read's bitmap being put together
    __shared__ int s_over;
    __shared__ uint32_t s_pref2[2][kLongMaxChunks + 1];   // a dealt bucket: pairs in its segment of chunk j (this item's, the next one's)
    __shared__ uint32_t s_wsum[kLongBlockBig / 64];
    uint32_t *bm = table + slots;
    const uint32_t max_slots = slots, bm_bits = bm_words * 32u;
    const uint32_t chunk = (n_items + 7u) / 8u;
#ifdef CID_LONG_PROF
    unsigned long long ff_t = __builtin_readcyclecounter();
#endif
    // The items of this workgroup, three in hand: the one being worked on, the next (whose counts are asked for now and stored at this
    // item's end) and the one after (asked for now).  A bucket's pass is forty-odd microseconds of which the trips to memory for its own
    // description were a third, with one workgroup to a CU and nothing to run meanwhile.
    const uint32_t it_step = gridDim.x >> 3, it_base = (blockIdx.x & 7u) * chunk;
    auto item_at = [&](uint32_t it, LongItem &o) -> bool {   // (workgroup-uniform)
        if (it >= chunk || it_base + it >= n_items) return false;
        o = items[it_base + it];
        return true;
    };
    uint32_t it = blockIdx.x >> 3, pb = 0;
    LongItem im{}, nxt{}, nn{};
    bool have_im = item_at(it, im), have_nxt = item_at(it + it_step, nxt);
    if (have_im && im.deal) for (uint32_t j = threadIdx.x; j < im.n_chunks; j += blockDim.x) s_pref2[0][j] = deal_counts[im.count_base + im.bucket * im.n_chunks + j];
    bool have_nn = false;
    for (; have_im; im = nxt, nxt = nn, have_im = have_nxt, have_nxt = have_nn, it += it_step, pb ^= 1u) {
        have_nn = item_at(it + 2u * it_step, nn);
        uint32_t cnt_next = 0;   // (n_chunks <= 256 <= blockDim.x)
        if (have_nxt && nxt.deal && threadIdx.x < nxt.n_chunks) cnt_next = deal_counts[nxt.count_base + nxt.bucket * nxt.n_chunks + threadIdx.x];
        uint32_t *s_pref = s_pref2[pb];
        const uint64_t w0 = im.w0;
        const uint32_t nw = im.nw;
        const uint64_t *rc = codes + w0;
        // a short read takes a corner of the table: clearing and sweeping it is what a pass costs beyond its inserts
        slots = 1024;
        while (slots < max_slots && slots < 2u * nw) slots <<= 1;
        const uint32_t mask = slots - 1;
        uint32_t level = 0;
        for (uint32_t sub = 0; sub < (1u << level); ++sub) {
            for (uint32_t s = threadIdx.x; s < slots / 4u; s += blockDim.x)   // (16 bytes a store: slots is a power of two >= 1 024)
                reinterpret_cast<uint4 *>(table)[s] = uint4{kLongEmpty, kLongEmpty, kLongEmpty, kLongEmpty};
            if (threadIdx.x == 0) s_over = 0;
            __syncthreads();
            FF_MARK(0);
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
            if (im.deal) {   // the bucket's own pairs: segment j (one per chunk of the read, `cap` places) holds s_pref[j] of them
                FF_MARK(1);
                // This pass is bound by the INSTRUCTIONS it issues (SQ counters, 100 kb reads: 3 070 vector and 2 800 scalar instructions per
                // wave and bucket, the vector unit busy for more than half of the kernel's cycles at four waves to a SIMD; without its inserts
                // the kernel took 0.54 of its 1.45 ms), so the common k-mer's way through it is made short:
                //  * a WAVE takes a segment (a part of one when the read has fewer segments than the workgroup has waves), its lanes the pairs
                //    that are there — no place without a pair is visited, no division finds a place's segment;
                //  * the next 256 pairs are asked for before the current 256 are inserted, every load unconditional (loads under a condition
                //    are loads the compiler waits for one by one — and for the next ones with them);
                //  * the table is buckets of four slots filled from the front.  ONE 16-byte read shows a k-mer its bucket; no equal tag there
                //    and a free slot: one compare-and-swap claims it, a lane's four k-mers in step and without a branch (a claim that is not
                //    wanted compares with a word no slot ever holds).  Everything else — an equal tag, a full bucket, a claim lost to another
                //    lane: one k-mer in twenty — is put aside in the wave's queue and inserted the general way, 64 lanes at a time: written
                //    where it happens, the rare case costs every k-mer of the wave its instructions;
                //  * a slot holds the PLACE of its k-mer's pair (16 bits: a bucket's places are fewer than 45 056) and 16 bits of tag: with the
                //    window (22 bits) and 10 bits of tag one occupied slot in a thousand sent its wave to memory to compare the codes.  Which of
                //    two equal k-mers came first is asked of their windows in memory when they meet — repeats only — and the later one's
                //    bit leaves the bitmap k_long_deal wrote: found exactly once, whichever lane finds it.
                const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
                const uint32_t nc = im.n_chunks, cap = im.cap;
                const uint32_t parts = nc < n_waves ? n_waves / nc : 1u, units = nc * parts;
                const uint64_t *pc = pair_code + im.pair_base + (uint64_t)im.bucket * nc * cap;
                const uint32_t *pi = pair_idx + im.pair_base + (uint64_t)im.bucket * nc * cap;
                const uint32_t bmask = (slots >> 2) - 1u, tmask = 0xFFFFu;
                const uint64_t lt_mask = (1ull << lane) - 1ull;
                uint32_t *queue = bm + wave * kDealQueue;
                uint32_t q_n = 0;
                auto general = [&](uint32_t x) {   // place x's k-mer into the table, whatever it meets
                    const uint64_t code = pc[x];
                    const uint32_t w = pi[x];
                    const uint32_t h = long_mix32(code), m = (x << 16) | ((h >> 13) & tmask);
                    uint32_t bk = h & bmask;
                    for (uint32_t steps = 0;; ++steps) {
                        const uint4 v = reinterpret_cast<const uint4 *>(table)[bk];
                        const uint32_t e[4] = {v.x, v.y, v.z, v.w};
                        uint32_t used = 0;
                        bool done = false;
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) {
                            if (done || e[q] == kLongEmpty) continue;
                            ++used;
                            if ((e[q] ^ m) & tmask) continue;
                            uint32_t cur = e[q];
                            if (pc[cur >> 16] != code) continue;
                            uint32_t later;   // the slot is this k-mer's: the earlier window of the two keeps it
                            for (;;) {
                                const uint32_t wo = pi[cur >> 16];
                                if (wo < w) { later = w; break; }
                                const uint32_t got = atomicCAS(&table[4u * bk + q], cur, m);
                                if (got == cur) { later = wo; break; }
                                cur = got;   // (another pair of this k-mer took the slot meanwhile)
                            }
                            const uint64_t lw = w0 + later;
                            atomicAnd(&bitmap[lw >> 5], ~(1u << (lw & 31u)));
                            done = true;
                        }
                        if (done) break;
                        if (used < 4) { if (atomicCAS(&table[4u * bk + used], kLongEmpty, m) == kLongEmpty) break; }   // (else: the bucket once more)
                        else bk = (bk + 1u) & bmask;
                        if (steps >= 256u) { s_over = 1; break; }   // a crowded table: the pass is redone on sub-buckets
                    }
                };
                auto drain = [&]() {
                    wave_lds_fence();
                    for (uint32_t b = 0; b < q_n; b += 64u) if (b + lane < q_n) general(queue[b + lane]);
                    wave_lds_fence();
                    q_n = 0;
                };
                // the wave's stretches of 256 pairs, one after the other: (seg0 + i .. seg0 + end) in the bucket's places
                uint32_t unit = wave, seg0 = 0, i = 0, end = 0;
                auto open_unit = [&]() -> bool {
                    for (; unit < units; unit += n_waves) {
                        const uint32_t j = parts > 1u ? unit % nc : unit, part = parts > 1u ? unit / nc : 0u;
                        const uint32_t c = s_pref[j], per = ((c + parts - 1u) / parts + 63u) & ~63u;
                        i = part * per;
                        end = c < i + per ? c : i + per;
                        seg0 = j * cap;
                        if (i < end) return true;
                    }
                    return false;
                };
                uint64_t code_n[4];
                uint32_t w_n[4], x_n[4];
                bool have_n[4];
                auto fetch = [&]() {
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t at = i + lane + 64u * u;
                        have_n[u] = at < end;
                        x_n[u] = seg0 + (have_n[u] ? at : i);
                        code_n[u] = pc[x_n[u]];
                        w_n[u] = pi[x_n[u]];
                    }
                };
                bool more = open_unit();
                if (more) fetch();
                while (more) {
                    uint32_t bk4[4], mine4[4], x4[4], act = 0;
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t h = long_mix32(code_n[u]);
                        x4[u] = x_n[u];
                        bk4[u] = h & bmask;
                        mine4[u] = (x_n[u] << 16) | ((h >> 13) & tmask);
                        if (have_n[u] && (!level || ((h >> 25) & ((1u << level) - 1u)) == sub)) act |= 1u << u;
                    }
                    i += 256u;
                    if (i >= end) { unit += n_waves; more = open_unit(); }
                    if (more) fetch();
                    FF_MARK(2);
                    if (q_n > kDealQueue - 256u) drain();   // (wave-uniform)
                    uint4 v4[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) v4[u] = reinterpret_cast<const uint4 *>(table)[bk4[u]];
                    uint32_t at4[4], cmp4[4], cas4[4], want = 0;
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint4 v = v4[u];
                        const uint32_t m = mine4[u];
                        const bool n0 = v.x != kLongEmpty, n1 = v.y != kLongEmpty, n2 = v.z != kLongEmpty, n3 = v.w != kLongEmpty;
                        const uint32_t used = (uint32_t)n0 + (uint32_t)n1 + (uint32_t)n2 + (uint32_t)n3;
                        const bool tag = (n0 && !((v.x ^ m) & tmask)) || (n1 && !((v.y ^ m) & tmask)) || (n2 && !((v.z ^ m) & tmask)) || (n3 && !((v.w ^ m) & tmask));
                        const bool claim = ((act >> u) & 1u) && !tag && used < 4u;
                        at4[u] = 4u * bk4[u] + (claim ? used : 0u);
                        cmp4[u] = claim ? kLongEmpty : kLongEmpty - 1u;
                        if (claim) want |= 1u << u;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) cas4[u] = atomicCAS(&table[at4[u]], cmp4[u], mine4[u]);
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const bool aside = ((act >> u) & 1u) && !(((want >> u) & 1u) && cas4[u] == kLongEmpty);
                        const uint64_t am = __ballot(aside);
                        if (aside) queue[q_n + (uint32_t)__popcll(am & lt_mask)] = x4[u];
                        q_n += (uint32_t)__popcll(am);
                    }
                    FF_MARK(3);
                }
                if (q_n) drain();
                FF_MARK(5);
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
            FF_MARK(4);
            if (s_over) {   // (workgroup-uniform)
                if (level == kLongMaxLevel) {
                    if (threadIdx.x == 0) { redo[im.read] = 1; atomicOr(&flags[1], 1); }   // the host redoes this read on the sorting path
                    break;
                }
                ++level;
                sub = ~0u;   // from the first sub-bucket of the finer level (bits set so far stay right: they are first occurrences)
                __syncthreads();
                continue;
            }
            if (im.deal) continue;   // (workgroup-uniform) nothing to add: the later occurrences' bits went as they were met
            // The winners' bits: put together in LDS, a stretch of bm_bits windows at a time, and written out as whole words (a read's
            // windows start at a multiple of 32, so its words are its own).  One global atomicOr per BIT took 4.4 of this kernel's
            // 5.4 ms on 150 Mbases of 10 kb reads: the atomics of a read all fall into its dozen of 128-byte lines.
            const bool own_words = im.n_buckets == 1 && level == 0;   // else other passes add to the same words
            // A read of several stretches (a megabase: eight of them at 131 072 windows a stretch, the table swept once for each): the table's
            // entries are first pushed together into its lower half — they are at most half its slots — and the upper half joins the bitmap's
            // stretch: 655 360 windows a stretch, and a sweep reads the entries there are, not the slots (1 Mb reads: 3.45 -> ms).
            uint32_t *bmx = bm;
            uint32_t bmx_words = bm_words, n_entries = slots;
            if (nw > bm_bits && slots == max_slots && slots == 32u * blockDim.x) {   // (workgroup-uniform)
                uint32_t mine[32], cnt = 0;
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) {
                    const uint4 v = reinterpret_cast<const uint4 *>(table)[threadIdx.x * 8u + q];
                    mine[4 * q] = v.x; mine[4 * q + 1] = v.y; mine[4 * q + 2] = v.z; mine[4 * q + 3] = v.w;
                }
#pragma unroll
                for (uint32_t q = 0; q < 32; ++q) cnt += mine[q] != kLongEmpty ? 1u : 0u;
                uint32_t inc = cnt;
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += up;
                }
                if (lane == 63) s_wsum[wave] = inc;
                __syncthreads();   // (every thread holds its slots: the table may be written)
                uint32_t base = 0, total = 0;
                for (uint32_t t = 0; t < blockDim.x / 64u; ++t) { if ((int)t < wave) base += s_wsum[t]; total += s_wsum[t]; }
                if (total <= slots / 2u) {   // (workgroup-uniform)
                    uint32_t at = base + inc - cnt;
#pragma unroll
                    for (uint32_t q = 0; q < 32; ++q) if (mine[q] != kLongEmpty) table[at++] = mine[q];
                    bmx = table + slots / 2u;
                    bmx_words = slots / 2u + bm_words;
                    n_entries = total;
                }
                __syncthreads();
            }
            const uint32_t bmx_bits = bmx_words * 32u;
            for (uint32_t c0 = 0; c0 < nw; c0 += bmx_bits) {
                for (uint32_t i = threadIdx.x; i < bmx_words; i += blockDim.x) bmx[i] = 0;
                __syncthreads();
                for (uint32_t s = threadIdx.x; s < n_entries; s += blockDim.x) {
                    const uint32_t cur = table[s];
                    if (cur != kLongEmpty) {
                        const uint32_t w = (cur >> kLongTagBits) - c0;
                        if (w < bmx_bits) atomicOr(&bmx[w >> 5], 1u << (w & 31u));
                    }
                }
                __syncthreads();
                const uint32_t words = (nw - c0 + 31u) / 32u < bmx_words ? (nw - c0 + 31u) / 32u : bmx_words;
                uint32_t *out = bitmap + ((w0 + c0) >> 5);
                for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
                    const uint32_t v = bmx[i];
                    if (own_words) out[i] = v;
                    else if (v) atomicOr(&out[i], v);
                }
                __syncthreads();
            }
        }
        if (have_nxt && nxt.deal && threadIdx.x < nxt.n_chunks) s_pref2[pb ^ 1u][threadIdx.x] = cnt_next;   // (read after the next table's clearing barrier)
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


// ------------------------------------------------------------------------------------------------
// The plan of a batch, made on the device (round 6): which reads this path takes, where their windows are numbered, and every work list
// of the kernels below — from seq_off / read_seq0 in HBM.  Until round 5 the host walked the offsets and filled ten std::vectors per call
// (0.6 ms of a 8.2 ms call on 15 000 reads of 10 kb) and the device FASTQ front end, whose offsets never exist on the host, had to
// refuse long reads altogether.
//   k_long_plan   one thread per read: its class, its windows; ONE pass of seven decoupled look-back scans (cid_scan.hpp's, a word per
//                 tile and field) gives every read its place in every list; the last tile leaves the totals
//   (the host reads the totals — 64 bytes, the one wait before the kernels — and sizes the lists)
//   k_long_emit   one wave per read: its entries of the lists
enum LongClass : uint8_t {
    kClsOther = 0,      // not routed here (status 2)
    kClsShort = 1,      // the first mate has no window (status 1: too_short)
    kClsFusedTiny = 2,  // one hash table; codes and table in ONE kernel (k_long_fused): a wave / 2 048 slots (reads of up to 1 024 windows)
    kClsFusedSmall = 3, //   ... 256 threads / 8 192 slots
    kClsFusedBig = 4,   //   ... 1 024 threads / 32 768 slots
    kClsFusedMulti = 5, //   ... the same table filled up to three times, a third of the read's k-mers each time (reads of up to 49 152 windows)
    kClsItemsSmall = 6, // k_extract_codes + k_long_first_flags: reads whose bases do not fit the fused kernel's LDS (strides), many mates
    kClsItemsBig = 7,   //   ... and every read of several buckets
    kClsSorted = 8      // more windows than a 4-byte slot numbers: the sorting path (redo[r] is set from the start)
};
constexpr uint32_t kFuseSeqs = 4;                 // sequences (mates) of a read the fused kernel keeps a table of
// Reads of up to 1 024 windows (the shortest this path takes, 700 bases up) by a WAVE each: a workgroup of 256 threads spent such a read's
// few microseconds mostly at its barriers, four reads in flight per CU; a wave needs none, and sixteen of them fit (8.8 KiB of LDS each).
constexpr uint32_t kLongSlotsTiny = 2048, kLongBlockTiny = 64, kLongTinyWin = 1024, kFusePosTiny = 1536, kPlanTinyShift = 40;
constexpr uint32_t kFusePosSmall = 6144, kFusePosBig = 20480;   // positions (bases, each mate padded to its 16-byte pieces) of a read in its LDS
// Reads of 16 385 .. 49 152 windows (what a long-read sequencer's run mostly consists of) used to leave the fused kernel for k_extract_codes +
// k_long_first_flags, two and three buckets each re-reading the read's codes: 9.2 ms per 150 Mbases at 30 kb against 7.0 at 10 kb.  Their
// bases fit the LDS beside the table all the same (3 bits a base), so the fused kernel takes them in PASSES: the table is filled with the
// k-mers of one hash bucket, its winners' bits join the read's bitmap in LDS, and again for the next bucket.
constexpr uint32_t kFusePassesMax = 3, kFuseWinMulti = kFusePassesMax * kLongFill, kFusePosMulti = 51200, kFusePiecesMulti = 4;
constexpr uint32_t kPlanFields = 7, kPlanPer = 4, kPlanTile = kScanBlock * kPlanPer;
struct LongPlanParams {
    const uint64_t *seq_off, *read_seq0;
    const uint8_t *route;     // NULL: every read
    const uint8_t *bases;     // (only its address: the 16-byte pieces of the fused kernel are pieces of the address space)
    uint64_t n_reads;
    uint32_t k, stride, seg_win;
    uint32_t fuse, cut, own_search, deal, multi, tiny;
    // out
    uint64_t *wstart, *wend;   // [n_reads + 1], [n_reads]
    uint32_t *win;             // [n_reads]
    uint8_t *cls, *redo, *status;
    uint64_t *pre;             // [kPlanFields][n_reads + 1]: exclusive prefixes
    uint64_t *state;           // [kPlanFields][tiles + 2] look-back words (zeroed)
    uint64_t *totals;          // [kPlanFields + 1]
    int *flags;
};
constexpr uint32_t kPlanMultiShift = 46;   // (places: fewer than 2^34 of them; reads of the third list: fewer than 2^18)
struct LongRow { uint32_t win, segs, cls; };
__device__ __forceinline__ LongRow long_row(const LongPlanParams &p, uint64_t r) {
    LongRow o{0u, 0u, kClsOther};
    if (p.route && p.route[r] != 1) return o;
    const uint64_t s0 = p.read_seq0[r], s1 = p.read_seq0[r + 1];
    if (s1 <= s0 || p.seq_off[s0 + 1] - p.seq_off[s0] < p.k) { o.cls = kClsShort; return o; }
    uint64_t win = 0, segs = 0, pos = 0;
    uint32_t nk = 0;
    for (uint64_t s = s0; s < s1; ++s) {
        const uint64_t a = p.seq_off[s], len = p.seq_off[s + 1] - a;
        if (len < p.k) continue;
        const uint64_t nw = (len - p.k) / p.stride + 1;
        win += nw;
        segs += (nw + p.seg_win - 1) / p.seg_win;
        pos += (len + ((uint64_t)(uintptr_t)(p.bases + a) & 15u) + 31u) & ~(uint64_t)31;
        ++nk;
    }
    if (win > kLongMaxWin) { o.cls = kClsSorted; return o; }
    o.win = (uint32_t)win;
    o.segs = (uint32_t)segs;
    const bool can_fuse = p.fuse && nk <= kFuseSeqs;
    if (can_fuse && p.tiny && win <= kLongTinyWin && pos <= kFusePosTiny) o.cls = kClsFusedTiny;
    else if (can_fuse && win <= kLongSmallWin && pos <= kFusePosSmall) o.cls = kClsFusedSmall;
    else if (can_fuse && win <= kLongFill && pos <= kFusePosBig) o.cls = kClsFusedBig;
    else if (can_fuse && p.multi && win <= kFuseWinMulti && pos <= kFusePosMulti) o.cls = kClsFusedMulti;
    else o.cls = win <= kLongSmallWin ? kClsItemsSmall : kClsItemsBig;
    if (o.cls >= kClsFusedTiny && o.cls <= kClsFusedMulti) o.segs = 0;
    return o;
}
// what read r adds to each of the seven lists
struct LongDealShape { uint32_t P, nc, cap; bool deal; };
__device__ __forceinline__ LongDealShape long_deal_shape(uint32_t win, uint32_t cls, uint32_t deal_on) {
    LongDealShape d{0u, 0u, 0u, false};
    if (cls != kClsItemsBig) return d;
    d.P = (win + kLongFill - 1) / kLongFill;
    if (d.P >= kDealFromBuckets && deal_on) {
        d.deal = true;
        d.nc = (win + kDealChunk - 1) / kDealChunk;
        d.cap = kDealChunk / d.P + kDealChunk / d.P / 4 + 96;
    }
    return d;
}
__device__ __forceinline__ void long_fields(const LongPlanParams &p, const LongRow &row, uint64_t v[kPlanFields]) {
    const bool windows = row.cls >= kClsFusedTiny && row.cls <= kClsItemsBig;
    const LongDealShape d = long_deal_shape(row.win, row.cls, p.deal);
    const uint32_t n_sl = !windows || !p.own_search ? 0u : (p.cut ? (row.win + kSliceWindows - 1) / kSliceWindows : 1u);
    v[0] = (windows ? ((uint64_t)row.win + 31u) & ~(uint64_t)31 : 0ull) | (row.cls == kClsFusedTiny ? 1ull << kPlanTinyShift : 0ull);   // windows (a read's start at a multiple of 32; fewer than 2^33 in all) | the fused kernel's list of the shortest reads
    v[1] = (row.cls == kClsFusedSmall ? 1ull : 0ull) | (row.cls == kClsFusedBig ? 1ull << 32 : 0ull);    // the fused kernel's two lists
    v[2] = (row.cls == kClsItemsSmall ? 1ull : 0ull) | ((uint64_t)d.P << 32);                            // k_long_first_flags' items
    v[3] = (uint64_t)n_sl | (n_sl > 1 ? 1ull << 32 : 0ull);                                              // slices | reads of several slices
    v[4] = (uint64_t)row.segs | (d.deal ? 1ull << 32 : 0ull);                                            // k_extract_codes' segments | deals
    v[5] = (uint64_t)d.nc | ((uint64_t)d.P * d.nc) << 32;                                                // chunks of dealt reads | their segment counters
    v[6] = (uint64_t)d.P * d.nc * d.cap | (row.cls == kClsFusedMulti ? 1ull << kPlanMultiShift : 0ull);   // places of dealt pairs | the fused kernel's third list
}
__global__ __launch_bounds__(kScanBlock) void k_long_plan(LongPlanParams p) {
    const uint64_t n = p.n_reads + 1;   // (element n_reads: empty — it receives the totals' positions, wstart[n_reads] among them)
    const uint64_t tiles = (n + kPlanTile - 1) / kPlanTile;
    const uint64_t tile = scan_ticket(p.state, tiles);
    if (tile >= tiles) return;
    const uint64_t i0 = tile * kPlanTile + (uint64_t)threadIdx.x * kPlanPer;
    LongRow row[kPlanPer];
    uint64_t v[kPlanPer][kPlanFields];
#pragma unroll
    for (uint32_t j = 0; j < kPlanPer; ++j) {
        row[j] = i0 + j < p.n_reads ? long_row(p, i0 + j) : LongRow{0u, 0u, kClsOther};
        long_fields(p, row[j], v[j]);
    }
#pragma unroll
    for (uint32_t f = 0; f < kPlanFields; ++f) {
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPlanPer; ++j) mine += v[j][f];
        uint64_t tile_sum;
        const uint64_t local = scan_block_exclusive(mine, &tile_sum);
        const uint64_t tile_excl = scan_lookback_block(p.state + (size_t)f * (tiles + 2), tile, tiles, tile_sum);
        uint64_t run = tile_excl + local;
#pragma unroll
        for (uint32_t j = 0; j < kPlanPer; ++j) {
            if (i0 + j < n) p.pre[(size_t)f * n + i0 + j] = run;
            if (f == 0 && i0 + j < n) {
                const uint64_t w_at = run & ((1ull << kPlanTinyShift) - 1ull);
                p.wstart[i0 + j] = w_at;
                if (i0 + j < p.n_reads) p.wend[i0 + j] = w_at + row[j].win;
            }
            run += v[j][f];
        }
        if (tile == tiles - 1 && threadIdx.x == 0) p.totals[f] = tile_excl + tile_sum;
    }
    bool any_sorted = false;
#pragma unroll
    for (uint32_t j = 0; j < kPlanPer; ++j) {
        if (i0 + j >= p.n_reads) continue;
        const uint32_t cls = row[j].cls;
        p.win[i0 + j] = row[j].win;
        p.cls[i0 + j] = (uint8_t)cls;
        p.redo[i0 + j] = cls == kClsSorted ? 1 : 0;
        p.status[i0 + j] = cls == kClsShort ? 1 : (cls == kClsOther || cls == kClsSorted ? 2 : 0);
        any_sorted = any_sorted || cls == kClsSorted;
    }
    if (any_sorted) atomicOr(&p.flags[2], 1);
}

// What k_long_fused wants to know of a read, in ONE record (read through list[] -> wstart / wend -> read_seq0 -> seq_off, the workgroup
// waited for four dependent round trips per read before its first base arrived): its windows, and per mate the 16-byte-aligned address its
// pieces start at, the position of its first base among the read's staged positions, its length, its first window, its first piece.
struct FuseItem {
    uint32_t read, nw, n_seq, n_pieces;
    uint64_t w0, pad;
    uint64_t addr[kFuseSeqs];
    uint32_t first[kFuseSeqs], len[kFuseSeqs], wbase[kFuseSeqs], piece0[kFuseSeqs];
};
struct LongLists {
    FuseItem *fused_tiny, *fused_small, *fused_big, *fused_multi;   // (one array, in this order)
    LongItem *items_small, *items_big;
    ReadSlice *slices;
    ReadCombine *combs;
    Segment *segs;
    uint32_t *seg_read;
    LongDeal *deals;
    uint32_t *chunk_deal, *chunk_no;
};
// One WAVE per read: the lists of a megabase read are hundreds of entries each (490 segments, 244 slices, 61 items and chunks), written by
// one thread they took 160 us per 150 such reads.
// LANES = 64: a wave per read, for the reads of k_long_first_flags (their lists are long).  LANES = 1: a thread per read, for the reads of the fused
// kernel — one record and a dozen slices at most: 187 500 reads of 800 bases kept a wave each busy with one lane, 0.2 ms of the call's 7.9.
template <uint32_t LANES>
__global__ __launch_bounds__(256) void k_long_emit(LongPlanParams p, LongLists L) {
    const uint64_t r = LANES == 64u ? (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = LANES == 64u ? threadIdx.x & 63u : 0u;
    if (r >= p.n_reads) return;
    const uint32_t cls = p.cls[r];
    if (cls < kClsFusedTiny || cls > kClsItemsBig) return;
    if ((cls <= kClsFusedMulti) != (LANES == 1u)) return;   // (the other launch's read)
    const uint64_t n = p.n_reads + 1;
    const uint32_t win = p.win[r];
    const uint64_t w0 = p.wstart[r];
    auto pre = [&](uint32_t f) { return p.pre[(size_t)f * n + r]; };
    if (cls >= kClsFusedTiny && cls <= kClsFusedMulti && lane == 0) {
        FuseItem it{};
        it.read = (uint32_t)r; it.nw = win; it.w0 = w0;
        uint32_t ns = 0, wb = 0, piece = 0;
        const uint64_t q0 = p.read_seq0[r], q1 = p.read_seq0[r + 1];
        for (uint64_t s = q0; s < q1 && ns < kFuseSeqs; ++s) {
            const uint64_t a = p.seq_off[s], len = p.seq_off[s + 1] - a;
            if (len < p.k) continue;
            const uint64_t addr = (uint64_t)(uintptr_t)(p.bases + a);
            const uint32_t o = (uint32_t)(addr & 15u);
            it.addr[ns] = addr - o;
            it.first[ns] = piece * 16u + o;
            it.len[ns] = (uint32_t)len;
            it.wbase[ns] = wb;
            it.piece0[ns] = piece;
            wb += (uint32_t)((len - p.k) / p.stride) + 1u;
            piece += (((uint32_t)len + o + 31u) & ~31u) / 16u;
            ++ns;
        }
        it.n_seq = ns; it.n_pieces = piece;
        if (cls == kClsFusedTiny) L.fused_tiny[(uint32_t)(pre(0) >> kPlanTinyShift)] = it;
        else if (cls == kClsFusedSmall) L.fused_small[(uint32_t)pre(1)] = it;
        else if (cls == kClsFusedBig) L.fused_big[(uint32_t)(pre(1) >> 32)] = it;
        else L.fused_multi[(uint32_t)(pre(6) >> kPlanMultiShift)] = it;
    }
    const LongDealShape d = long_deal_shape(win, cls, p.deal);
    if (cls == kClsItemsSmall && lane == 0) L.items_small[(uint32_t)pre(2)] = LongItem{(uint32_t)r, 0u, 1u, 0u, win, 0u, 0u, 0u, w0, 0ull};
    if (cls == kClsItemsBig) {
        uint32_t deal = 0, count_base = 0;
        uint64_t pair_base = 0;
        if (d.deal) {
            const uint32_t di = (uint32_t)(pre(4) >> 32), c0 = (uint32_t)pre(5);
            pair_base = pre(6) & ((1ull << kPlanMultiShift) - 1ull);
            count_base = (uint32_t)(pre(5) >> 32);
            if (lane == 0) L.deals[di] = LongDeal{pair_base, count_base, d.nc, d.cap, (uint32_t)r};
            deal = di + 1;
            for (uint32_t j = lane; j < d.nc; j += LANES) { L.chunk_deal[c0 + j] = di; L.chunk_no[c0 + j] = j; }
        }
        const uint32_t i0 = (uint32_t)(pre(2) >> 32);
        for (uint32_t b = lane; b < d.P; b += LANES) L.items_big[i0 + b] = LongItem{(uint32_t)r, b, d.P, deal, win, d.nc, d.cap, count_base, w0, pair_base};
    }
    if (p.own_search) {
        const uint32_t n_sl = p.cut ? (win + kSliceWindows - 1) / kSliceWindows : 1u;
        const uint32_t s0 = (uint32_t)pre(3);
        if (n_sl > 1 && lane == 0) L.combs[(uint32_t)(pre(3) >> 32)] = ReadCombine{(uint32_t)r, s0, n_sl, 0u};
        const uint64_t W1 = w0 + win;
        for (uint32_t j = lane; j < n_sl; j += LANES) {
            const uint64_t a = w0 + (uint64_t)j * kSliceWindows;
            const uint64_t b = n_sl == 1 ? W1 : (a + kSliceWindows < W1 ? a + kSliceWindows : W1);
            L.slices[s0 + j] = ReadSlice{(uint32_t)r, (uint32_t)a, (uint32_t)b, j | (n_sl > 1 ? 0x80000000u : 0u)};
        }
    }
    if (cls == kClsItemsSmall || cls == kClsItemsBig) {   // the segments k_extract_codes walks: windows numbered mate by mate
        uint32_t sg = (uint32_t)pre(4);
        uint64_t W = w0;
        const uint64_t q0 = p.read_seq0[r], q1 = p.read_seq0[r + 1];
        for (uint64_t s = q0; s < q1; ++s) {
            const uint64_t a = p.seq_off[s], len = p.seq_off[s + 1] - a;
            if (len < p.k) continue;
            const uint64_t nw = (len - p.k) / p.stride + 1;
            const uint64_t n_sg = (nw + p.seg_win - 1) / p.seg_win;
            for (uint64_t g = lane; g < n_sg; g += LANES) {
                const uint64_t x = g * p.seg_win;
                const uint32_t m = (uint32_t)(nw - x < p.seg_win ? nw - x : p.seg_win);
                L.segs[sg + g] = Segment{a + x * p.stride, W + x, m, p.stride};
                L.seg_read[sg + g] = (uint32_t)r;
            }
            sg += (uint32_t)n_sg;
            W += nw;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_long_fused: reads of ONE hash table (at most 16 384 windows: every 10 kb read) — windows, codes, table and bitmap in one workgroup, straight
// from the bases.  Round 5 ran k_extract_codes (1.2 GB of codes written per 150 Mbases) and then k_long_first_flags, whose insert loop
// waited for a global load per window (a round trip per iteration, 16 waves per CU to hide it: 23 us per 10 kb read, 1.37 ms per
// 150 Mbases) and read the code array back whenever two tags met.  Here a read's bases are staged once as 2-bit fields + bad-base bits
// in LDS (16-byte pieces of the address space, so every load is aligned whatever the read's offset), a window's code is a few shifts
// away whenever it is wanted — for the insert, for the tag check of another window, for the code array k_readid_slices walks — and a
// thread remembers the slot each of its windows ended in: the winners are the windows whose slot still holds their number, their bits
// leave as ballots (64 consecutive windows = two whole words of the read's own bitmap stretch; no LDS bitmap, no sweep of the table).
// A read with a lower-case base (its case would have to be kept: SURVEY App. B Q2) or a table that crowds anyway marks redo[read].
struct LongFuseParams {
    const FuseItem *items;
    uint32_t n_list, k, msz, stride, max_slots, pos_cap, bm_words;   // bm_words: the read's bitmap in LDS (its windows / 32)
    uint64_t sentinel;       // of the keys (k-mers, or minimizers)
    uint64_t *codes;
    uint32_t *bitmap;
    uint8_t *redo;
    int *flags;
    uint32_t *lower_list, *n_lower;   // non-NULL: a read with a lower-case base is listed here (item_base + item | big << 31) for k_long_bytes
    uint32_t big, item_base;
    unsigned long long *prof;   // CID_LONG_PROF builds: [gridDim.x][8] cycles per phase (thread 0's clock)
};
#ifdef CID_LONG_PROF
#define LONG_PROF_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define LONG_PROF_MARK(i) do { } while (0)
#endif
// this thread's (at most PIECES) 16-byte pieces of a read's bases; a piece of padding behind a mate is never loaded
template <uint32_t BLOCK, uint32_t PIECES>
__device__ __forceinline__ void fuse_load_pieces(const FuseItem *it, uint4 (&pc)[PIECES]) {
    const uint32_t ns = it->n_seq, n_pieces = it->n_pieces;
#pragma unroll
    for (uint32_t j = 0; j < PIECES; ++j) {
        const uint32_t x = threadIdx.x + j * BLOCK;
        pc[j] = uint4{0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu};   // 'N'
        if (x >= n_pieces) continue;
        uint32_t q = 0;
#pragma unroll
        for (uint32_t t = 1; t < kFuseSeqs; ++t) q += (t < ns && x >= it->piece0[t]) ? 1u : 0u;
        if (x * 16u < it->first[q] + it->len[q]) pc[j] = *reinterpret_cast<const uint4 *>(it->addr[q] + (uint64_t)(x - it->piece0[q]) * 16u);
    }
}
// MULTI: reads of up to kFuseWinMulti windows, their k-mers dealt to ceil(windows / kLongFill) hash buckets and the table filled once per
// bucket — every pass walks all the read's windows (the rolling codes are the cheap part) and inserts its own bucket's.
template <uint32_t BLOCK, uint32_t ITERS, bool MULTI = false>
__global__ __launch_bounds__(BLOCK, 4) void k_long_fused(LongFuseParams p) {
    constexpr uint32_t PIECES = MULTI ? kFusePiecesMulti : 2u;
    extern __shared__ __align__(16) uint32_t table[];   // max_slots, then the 2-bit fields (pos_cap / 16 + 4 words), the bad-base bits (pos_cap / 32 + 4 words), the bitmap
    __shared__ uint32_t s_first[kFuseSeqs], s_wbase[kFuseSeqs + 1], s_piece0[kFuseSeqs + 1];
    __shared__ uint32_t s_len[kFuseSeqs];
    __shared__ int s_over;
    uint32_t *s_pack = table + p.max_slots;
    uint32_t *s_bad = s_pack + p.pos_cap / 16 + 4;
    uint32_t *s_bm = s_bad + p.pos_cap / 32 + 4;   // p.bm_words: the read's first-occurrence bits
    const uint32_t pack_words = p.pos_cap / 16 + 4, bad_words = p.pos_cap / 32 + 4;
    const uint32_t k = p.k, stride = p.stride;
    const uint32_t chunk = (p.n_list + 7u) / 8u;   // (XCD x walks the x-th eighth of the list)
    const uint32_t item0 = (blockIdx.x & 7u) * chunk, item1 = item0 + chunk < p.n_list ? item0 + chunk : p.n_list;
    const uint32_t step = gridDim.x >> 3;
    uint32_t item = item0 + (blockIdx.x >> 3);
    uint4 pc[PIECES];
#ifdef CID_LONG_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
    if (item < item1) fuse_load_pieces<BLOCK, PIECES>(p.items + item, pc);
    for (; item < item1; item += step) {
        LONG_PROF_MARK(7);
        const FuseItem *rec = p.items + item;
        const uint32_t read = rec->read, nw = rec->nw, ns = rec->n_seq, n_pieces = rec->n_pieces;
        const uint64_t w0 = rec->w0;
        uint32_t slots = 1024;
        while (slots < p.max_slots && slots < 2u * nw) slots <<= 1;
        const uint32_t mask = slots - 1;
        if (threadIdx.x < kFuseSeqs) {   // the read's mates: where their positions, windows and pieces start
            const uint32_t t = threadIdx.x;
            s_first[t] = rec->first[t]; s_len[t] = rec->len[t];
            s_wbase[t] = t < ns ? rec->wbase[t] : nw;
            s_piece0[t] = t < ns ? rec->piece0[t] : n_pieces;
            if (t == 0) { s_wbase[kFuseSeqs] = nw; s_piece0[kFuseSeqs] = n_pieces; s_over = 0; }
        }
        for (uint32_t s = threadIdx.x; s < slots / 4u; s += BLOCK)   // (16 bytes a store: slots is a power of two >= 1 024)
            reinterpret_cast<uint4 *>(table)[s] = uint4{kLongEmpty, kLongEmpty, kLongEmpty, kLongEmpty};
        for (uint32_t s = threadIdx.x; s < bad_words; s += BLOCK) s_bad[s] = 0xFFFFFFFFu;
        for (uint32_t s = threadIdx.x; s < pack_words; s += BLOCK) s_pack[s] = 0u;
        for (uint32_t s = threadIdx.x; s < p.bm_words; s += BLOCK) s_bm[s] = 0u;
        __syncthreads();
        LONG_PROF_MARK(0);   // record + clears
        bool lower = false;
#pragma unroll
        for (uint32_t j = 0; j < PIECES; ++j) {
            const uint32_t x = threadIdx.x + j * BLOCK;
            if (x >= n_pieces) continue;
            uint32_t q = 0;
            while (q + 1 < ns && x >= s_piece0[q + 1]) ++q;
            const uint32_t pos = x * 16u;                                   // position of this piece's first byte
            const uint32_t lo = s_first[q], hi = s_first[q] + s_len[q];      // the mate's bases: positions [lo, hi)
            if (pos >= hi) continue;                                         // (padding behind the mate: its bits stay "bad")
            const uint32_t words[4] = {pc[j].x, pc[j].y, pc[j].z, pc[j].w};
            uint32_t code = 0, bad = 0, low = 0;
#pragma unroll
            for (uint32_t t = 0; t < 16; ++t) {
                const uint32_t b = (words[t >> 2] >> (8u * (t & 3u))) & 0xFFu;
                const uint32_t c2 = (b >> 1) & 3u;
                code |= (c2 ^ (c2 >> 1)) << (2 * t);
                const bool inside = pos + t >= lo && pos + t < hi;
                const bool good = inside && good_base_dev(b);
                bad |= (good ? 0u : 1u) << t;
                low |= (good ? (b >> 5) & 1u : 0u) << t;
            }
            s_pack[x] = code;
            reinterpret_cast<uint16_t *>(s_bad)[x] = (uint16_t)bad;
            lower = lower || low != 0;
        }
        // the NEXT read's bases are asked for now: they arrive while this read's table is filled
        if (item + step < item1) fuse_load_pieces<BLOCK, PIECES>(p.items + item + step, pc);
        LONG_PROF_MARK(1);   // staging (own part)
        if (__syncthreads_or(lower ? 1 : 0)) {   // (workgroup-uniform) a byte-string path takes this read: k_long_bytes, or the sorting path
            if (threadIdx.x == 0) {
                p.redo[read] = 1;
                if (p.lower_list) p.lower_list[atomicAdd(p.n_lower, 1u)] = (p.item_base + item) | (p.big << 31);
                else atomicOr(&p.flags[0], 1);
            }
            __syncthreads();
            continue;
        }
        auto code_at = [&](uint32_t w, bool &valid) -> uint64_t {   // the key of the read's window w
            uint32_t q = 0;
            while (q + 1 < ns && w >= s_wbase[q + 1]) ++q;
            const uint32_t pos = s_first[q] + (w - s_wbase[q]) * stride;
            if (bits_at_dev(s_bad, pos, k) != 0) { valid = false; return p.sentinel; }
            const uint64_t lsb = bits_at_dev(s_pack, 2u * pos, 2u * k);
            uint64_t msb;
            canonical_code(lsb, k, &msb);
            valid = true;
            return p.msz ? minimizer_code(msb, k, p.msz) : msb;
        };
        LONG_PROF_MARK(2);   // barrier behind the staging
        // A thread takes w_per CONSECUTIVE windows: the forward and the reverse-complement code of a window follow from the window before by
        // one base each (the strided assignment paid two unaligned 64-bit extractions and a field reversal per window: ~150 instructions a
        // window, 24 us per 10 kb read with sixteen waves on a CU — the kernel was instruction-bound).
        const uint32_t w_per = (nw + BLOCK - 1) / BLOCK;   // <= ITERS (MULTI: <= kFusePassesMax * ITERS)
        const uint32_t wa = threadIdx.x * w_per;
        const uint32_t n_pass = MULTI ? (nw + kLongFill - 1) / kLongFill : 1u;
        bool crowded = false;
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
        if (MULTI && pass) {   // (the winners of the pass before are in s_bm; the barrier behind its sweep lies before this)
            for (uint32_t s = threadIdx.x; s < slots / 4u; s += BLOCK)
                reinterpret_cast<uint4 *>(table)[s] = uint4{kLongEmpty, kLongEmpty, kLongEmpty, kLongEmpty};
            __syncthreads();
        }
        const uint64_t kmask = code_mask(k);
        uint32_t q = 0, next_base = 0, pos = 0, good_run = 0;
        uint64_t fwd = 0, rcv = 0;
        auto prime = [&](uint32_t w) {   // the rolling state at window w: both codes and the run of good bases ending at its last base
            q = 0;
            while (q + 1 < ns && w >= s_wbase[q + 1]) ++q;
            next_base = s_wbase[q + 1];
            pos = s_first[q] + (w - s_wbase[q]) * stride;
            const uint64_t lsb = bits_at_dev(s_pack, 2u * pos, 2u * k);
            fwd = rev_fields(lsb, k);
            rcv = ~lsb & kmask;
            const uint64_t badbits = bits_at_dev(s_bad, pos, k);
            good_run = badbits ? (uint32_t)__clzll((long long)badbits) - (64u - k) : k;   // the bases behind the last bad one
        };
        if (wa < nw) prime(wa);
        // (a ROLLED loop: unrolled sixteen times the kernel is 60 KB of code, the sixteen waves of a workgroup run all over it and the
        // instruction cache — 64 KB for two CUs — misses: 8.2 ms per 150 Mbases instead of 7.0)
#pragma unroll 1
        for (uint32_t i = 0; i < w_per; ++i) {
            const uint32_t w = wa + i;
            if (w >= nw) continue;
            if (i) {
                if (w == next_base || stride != 1) prime(w);   // the next mate; strides: every window from the bases
                else {                                          // one base further
                    ++pos;
                    const uint32_t pn = pos + k - 1;
                    const uint32_t c = (s_pack[pn >> 4] >> (2u * (pn & 15u))) & 3u;
                    const uint32_t bad = (reinterpret_cast<const uint16_t *>(s_bad)[pn >> 4] >> (pn & 15u)) & 1u;
                    fwd = ((fwd << 2) | c) & kmask;
                    rcv = (rcv >> 2) | ((uint64_t)(3u - c) << (2u * (k - 1u)));
                    good_run = bad ? 0u : good_run + 1u;
                }
            }
            uint64_t code = p.sentinel;
            const bool valid = good_run >= k;
            if (valid) {
                code = fwd < rcv ? fwd : rcv;   // (equal: the same string)
                if (p.msz) code = minimizer_code(code, k, p.msz);
            }
            if (!MULTI || pass == 0) p.codes[w0 + w] = code;
            if (!valid) continue;
            const uint32_t h = long_mix32(code);
            if (MULTI && (((h >> 15) & 127u) * n_pass) >> 7 != pass) continue;   // (bits between the slot's and the tag's)
            const uint32_t tag = h >> (32u - kLongTagBits);
            const uint32_t mine = (w << kLongTagBits) | tag;
            uint32_t at = h & mask;
            for (uint32_t probes = 0;; ++probes) {
                const uint32_t cur = atomicCAS(&table[at], kLongEmpty, mine);   // (at a load below a half most windows win their home slot at once)
                if (cur == kLongEmpty) break;
                if ((cur & ((1u << kLongTagBits) - 1u)) == tag) {
                    bool v2;
                    if (code_at(cur >> kLongTagBits, v2) == code) {   // the slot is this k-mer's
                        atomicMin(&table[at], mine);
                        break;
                    }
                }
                at = (at + 1) & mask;
                if (probes >= slots / 4) { s_over = 1; break; }
            }
        }
        LONG_PROF_MARK(3);   // windows + inserts (thread 0's own)
        __syncthreads();
        LONG_PROF_MARK(4);   // waiting for the others
        if (s_over) {   // (workgroup-uniform; never seen at a load of a half)
            if (threadIdx.x == 0) { p.redo[read] = 1; atomicOr(&p.flags[1], 1); }
            __syncthreads();
            crowded = true;
            break;
        }
        // the winners — the smallest window of every k-mer, which is what its slot holds — as bits of the read's own bitmap words
        for (uint32_t sl = threadIdx.x; sl < slots / 4u; sl += BLOCK) {
            const uint4 c4 = reinterpret_cast<const uint4 *>(table)[sl];
            const uint32_t cur[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
                if (cur[t] != kLongEmpty) atomicOr(&s_bm[cur[t] >> (kLongTagBits + 5u)], 1u << ((cur[t] >> kLongTagBits) & 31u));
        }
        __syncthreads();
        }   // (passes)
        if (crowded) continue;
        uint32_t *out = p.bitmap + (w0 >> 5);
        for (uint32_t i = threadIdx.x; i < (nw + 31u) / 32u; i += BLOCK) out[i] = s_bm[i];
        __syncthreads();
        LONG_PROF_MARK(5);   // winners + bitmap
    }
#ifdef CID_LONG_PROF
    if (threadIdx.x == 0 && p.prof) for (int i = 0; i < 8; ++i) p.prof[(size_t)blockIdx.x * 8 + i] = prof_acc[i];
#endif
}

// k_long_bytes: the reads k_long_fused found a lower-case base in (soft-masked sequence).  The reference keeps a base's case
// (kmer.rs:221-243 never upper-cases: SURVEY App. B Q2), so such a read's k-mers are byte strings: the canonical choice compares the
// bytes of the window and of its reverse complement (upper case sorts before lower case), "acgt" and "ACGT" are different k-mers, and
// the search hashes the bytes as they are.  Until round 5 one such base sent the whole batch through round 1's global rocPRIM sort (2.5 x
// the time, and a 10 MB code object loaded on first use: 40 ms of a 35 ms CLI classification); now the read — alone — is redone here:
// the bases as three planes in LDS (2-bit fields, bad-base bits, case bits), a k-mer's identity = (canonical code, its case bits in
// canonical order), the same table, the winners' bits into the read's bitmap words and at each winner's window a (byte offset << 1 | reverse
// complement) entry in the code array, which k_readid_slices' BYTES instantiation searches slice by slice like any other read.
struct LongBytesParams {
    const FuseItem *items_small, *items_big;
    const uint32_t *lower, *n_lower;
    const uint8_t *bases;
    uint32_t k, stride;
    uint64_t *codes;
    uint32_t *bitmap;
    uint8_t *bytes_read;
    uint8_t *redo;
    int *flags;
};
constexpr uint32_t kBytesBlock = 1024;
__global__ __launch_bounds__(kBytesBlock) void k_long_bytes(LongBytesParams p) {
    extern __shared__ __align__(16) uint32_t table[];   // kLongSlotsBig, then the three planes of kFusePosBig positions, then the bitmap
    __shared__ uint32_t s_first[kFuseSeqs], s_wbase[kFuseSeqs + 1], s_piece0[kFuseSeqs + 1], s_len[kFuseSeqs];
    __shared__ int s_over;
    constexpr uint32_t pack_words = kFusePosBig / 16 + 4, bit_words = kFusePosBig / 32 + 4, bm_words = kLongFill / 32;
    uint32_t *s_pack = table + kLongSlotsBig, *s_bad = s_pack + pack_words, *s_low = s_bad + bit_words, *s_bm = s_low + bit_words;
    const uint32_t k = p.k, stride = p.stride;
    const uint32_t n_lower = *p.n_lower;
    const uint64_t kmask = code_mask(k);
    const uint32_t lowmask = k >= 32 ? ~0u : ((1u << k) - 1u);
    for (uint32_t e = blockIdx.x; e < n_lower; e += gridDim.x) {
        const uint32_t le = p.lower[e];
        const FuseItem *rec = ((le >> 31) ? p.items_big : p.items_small) + (le & 0x7FFFFFFFu);
        const uint32_t read = rec->read, nw = rec->nw, ns = rec->n_seq, n_pieces = rec->n_pieces;
        const uint64_t w0 = rec->w0;
        uint32_t slots = 1024;
        while (slots < kLongSlotsBig && slots < 2u * nw) slots <<= 1;
        const uint32_t mask = slots - 1;
        if (threadIdx.x < kFuseSeqs) {
            const uint32_t t = threadIdx.x;
            s_first[t] = rec->first[t]; s_len[t] = rec->len[t];
            s_wbase[t] = t < ns ? rec->wbase[t] : nw;
            s_piece0[t] = t < ns ? rec->piece0[t] : n_pieces;
            if (t == 0) { s_wbase[kFuseSeqs] = nw; s_piece0[kFuseSeqs] = n_pieces; s_over = 0; }
        }
        for (uint32_t s = threadIdx.x; s < slots; s += kBytesBlock) table[s] = kLongEmpty;
        for (uint32_t s = threadIdx.x; s < bit_words; s += kBytesBlock) { s_bad[s] = 0xFFFFFFFFu; s_low[s] = 0u; }
        for (uint32_t s = threadIdx.x; s < pack_words; s += kBytesBlock) s_pack[s] = 0u;
        for (uint32_t s = threadIdx.x; s < bm_words; s += kBytesBlock) s_bm[s] = 0u;
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < n_pieces; x += kBytesBlock) {
            uint32_t q = 0;
            while (q + 1 < ns && x >= s_piece0[q + 1]) ++q;
            const uint32_t pos = x * 16u, lo = s_first[q], hi = s_first[q] + s_len[q];
            if (pos >= hi) continue;   // (padding behind the mate)
            const uint4 v = *reinterpret_cast<const uint4 *>(rec->addr[q] + (uint64_t)(x - s_piece0[q]) * 16u);
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
            uint32_t code = 0, bad = 0, low = 0;
#pragma unroll
            for (uint32_t t = 0; t < 16; ++t) {
                const uint32_t b = (words[t >> 2] >> (8u * (t & 3u))) & 0xFFu;
                const uint32_t c2 = (b >> 1) & 3u;
                code |= (c2 ^ (c2 >> 1)) << (2 * t);
                const bool good = pos + t >= lo && pos + t < hi && good_base_dev(b);
                bad |= (good ? 0u : 1u) << t;
                low |= (good ? (b >> 5) & 1u : 0u) << t;
            }
            s_pack[x] = code;
            reinterpret_cast<uint16_t *>(s_bad)[x] = (uint16_t)bad;
            reinterpret_cast<uint16_t *>(s_low)[x] = (uint16_t)low;
        }
        __syncthreads();
        // the k-mer of window w: canonical code (base 0 most significant), its case bits in canonical order (bit j: base j), the strand
        auto ident = [&](uint32_t w, uint64_t &code, uint32_t &cases, bool &fwd, uint32_t &pos_out, uint32_t &q_out) -> bool {
            uint32_t q = 0;
            while (q + 1 < ns && w >= s_wbase[q + 1]) ++q;
            const uint32_t pos = s_first[q] + (w - s_wbase[q]) * stride;
            pos_out = pos; q_out = q;
            if (bits_at_dev(s_bad, pos, k) != 0) return false;
            const uint64_t lsb = bits_at_dev(s_pack, 2u * pos, 2u * k);
            const uint32_t low = (uint32_t)bits_at_dev(s_low, pos, k);
            const uint64_t f_msb = rev_fields(lsb, k), rc_msb = ~lsb & kmask;
            if (low == 0u || low == lowmask) fwd = f_msb < rc_msb;   // one case: byte order is code order (equal: the reverse-complement branch)
            else {                                                    // mixed: the bytes decide, upper case before lower case
                fwd = false;
                for (uint32_t t = 0; t < k; ++t) {
                    const uint32_t fb = (uint32_t)(lsb >> (2u * t)) & 3u, fc = (low >> t) & 1u;
                    const uint32_t rb = 3u - ((uint32_t)(lsb >> (2u * (k - 1u - t))) & 3u), rcs = (low >> (k - 1u - t)) & 1u;
                    const uint32_t fkey = (fc << 2) | fb, rkey = (rcs << 2) | rb;
                    if (fkey != rkey) { fwd = fkey < rkey; break; }
                }
            }
            code = fwd ? f_msb : rc_msb;
            cases = fwd ? low : (__brev(low) >> (32u - k));
            return true;
        };
        for (uint32_t w = threadIdx.x; w < nw; w += kBytesBlock) {
            uint64_t code; uint32_t cases, pos, q; bool fwd;
            if (!ident(w, code, cases, fwd, pos, q)) continue;
            const uint32_t h = long_mix32(code ^ ((uint64_t)cases * 0x9E3779B97F4A7C15ull));
            const uint32_t tag = h >> (32u - kLongTagBits);
            const uint32_t mine = (w << kLongTagBits) | tag;
            uint32_t at = h & mask;
            for (uint32_t probes = 0;; ++probes) {
                const uint32_t cur = atomicCAS(&table[at], kLongEmpty, mine);
                if (cur == kLongEmpty) break;
                if ((cur & ((1u << kLongTagBits) - 1u)) == tag) {
                    uint64_t c2; uint32_t cs2, p2, q2; bool f2;
                    if (ident(cur >> kLongTagBits, c2, cs2, f2, p2, q2) && c2 == code && cs2 == cases) {   // the slot is this k-mer's
                        atomicMin(&table[at], mine);
                        break;
                    }
                }
                at = (at + 1) & mask;
                if (probes >= slots / 4) { s_over = 1; break; }
            }
        }
        __syncthreads();
        if (s_over) {   // (workgroup-uniform) redo[read] stays set: the sorting path
            if (threadIdx.x == 0) atomicOr(&p.flags[1], 1);
            __syncthreads();
            continue;
        }
        for (uint32_t sl = threadIdx.x; sl < slots; sl += kBytesBlock) {
            const uint32_t cur = table[sl];
            if (cur != kLongEmpty) atomicOr(&s_bm[cur >> (kLongTagBits + 5u)], 1u << ((cur >> kLongTagBits) & 31u));
        }
        __syncthreads();
        // the winners' bits into the read's own bitmap words, and at each winner's window its entry for the BYTES search: byte offset of
        // the k-mer in `bases` << 1 | reverse complement
        const uint32_t n_words = (nw + 31u) / 32u;   // <= 512
        if (threadIdx.x < n_words) {
            uint32_t word = s_bm[threadIdx.x];
            p.bitmap[(w0 >> 5) + threadIdx.x] = word;
            while (word) {
                const uint32_t w = threadIdx.x * 32u + (uint32_t)__builtin_ctz(word);
                word &= word - 1u;
                uint64_t code; uint32_t cases, pos, q; bool fwd;
                ident(w, code, cases, fwd, pos, q);
                const uint64_t off = (rec->addr[q] - (uint64_t)(uintptr_t)p.bases) + (uint64_t)(pos - s_piece0[q] * 16u);
                p.codes[w0 + w] = (off << 1) | (fwd ? 0ull : 1ull);
            }
        }
        if (threadIdx.x == 0) { p.bytes_read[read] = 1; p.redo[read] = 0; }
        __syncthreads();
    }
}

// Which reads of a batch take this path (route[r] = 1), which the LDS kernels (0), which neither (3: beyond the maxima a device-pointer
// caller stated — status 3, as k_readid_check_caps marks them).  Long = at least `long_from` bases.  stats: {long reads, LDS-kernel reads,
// the longest of those in bases, in windows}.
__global__ __launch_bounds__(256) void k_long_route(const uint64_t *seq_off, const uint64_t *read_seq0, uint64_t n_reads, uint32_t k, uint32_t stride,
                                                    uint64_t long_from, uint64_t cap_bytes, uint64_t cap_win, uint8_t *route, uint32_t *stats) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t is_long = 0, is_short = 0, sb = 0, sw = 0;
    if (r < n_reads) {
        const uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
        const uint64_t bytes = s1 > s0 ? seq_off[s1] - seq_off[s0] : 0;
        uint64_t win = 0;
        for (uint64_t s = s0; s < s1; ++s) {
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len >= k) win += (len - k) / stride + 1;
        }
        const bool beyond = bytes > cap_bytes || win > cap_win;
        const bool lng = !beyond && bytes >= long_from;
        route[r] = beyond ? 3 : (lng ? 1 : 0);
        is_long = lng ? 1u : 0u;
        is_short = (!beyond && !lng) ? 1u : 0u;
        if (is_short) { sb = (uint32_t)bytes; sw = (uint32_t)win; }   // (below long_from: small)
    }
    const uint32_t nl = (uint32_t)__popcll(__ballot(is_long)), nsh = (uint32_t)__popcll(__ballot(is_short));
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t ob = __shfl_xor(sb, d, 64), ow = __shfl_xor(sw, d, 64);
        sb = ob > sb ? ob : sb;
        sw = ow > sw ? ow : sw;
    }
    if ((threadIdx.x & 63) == 0) {
        if (nl) atomicAdd(&stats[0], nl);
        if (nsh) atomicAdd(&stats[1], nsh);
        if (sb) atomicMax(&stats[2], sb);
        if (sw) atomicMax(&stats[3], sw);
    }
}
// the reads beyond a device-pointer caller's stated maxima: status 3, no k-mers, an empty row (report_width == 0: the caller zeroed the rows)
__global__ __launch_bounds__(256) void k_long_beyond_rows(const uint8_t *route, uint64_t n_reads, uint32_t report_width, uint32_t *report, uint32_t *n_kmers,
                                                          uint8_t *status) {
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_reads || route[r] != 3) return;
    for (uint32_t c = threadIdx.x & 63u; c < report_width; c += 64u) report[r * (uint64_t)report_width + c] = 0;
    if ((threadIdx.x & 63u) == 0) { n_kmers[r] = 0; status[r] = 3; }
}
// list mode (rows counted in place): a read marked for the sorting path is taken out before anything is counted for it
__global__ void k_long_redo_status(const uint8_t *redo, uint8_t *status, uint64_t n_reads) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads && redo[r]) status[r] = 2;
}

// start-up: this unit's code object (the long-read kernels) loaded ahead of the first long read (cid_warmup)
hipError_t warm_readlong() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_long_route));
}

int long_route_launch(cid_ctx *c, const uint64_t *d_seq_off, const uint64_t *d_read_seq0, size_t n_reads, uint32_t k, uint32_t stride_d, uint64_t long_from,
                      uint64_t cap_bytes, uint64_t cap_win, uint8_t *d_route, uint32_t *d_stats) {
    hipStream_t st = ctx_stream(c);
    HIP_TRY(hipMemsetAsync(d_stats, 0, 16, st));
    hipLaunchKernelGGL(k_long_route, dim3(grid_for_n(n_reads)), dim3(256), 0, st, d_seq_off, d_read_seq0, (uint64_t)n_reads, k, stride_d, long_from, cap_bytes, cap_win,
                       d_route, d_stats);
    HIP_TRY(hipGetLastError());
    return CID_OK;
}
int long_beyond_launch(cid_ctx *c, const uint8_t *d_route, size_t n_reads, uint32_t report_width, uint32_t *d_report, uint32_t *d_n_kmers, uint8_t *d_status) {
    hipLaunchKernelGGL(k_long_beyond_rows, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, ctx_stream(c), d_route, (uint64_t)n_reads, report_width, d_report,
                       d_n_kmers, d_status);
    HIP_TRY(hipGetLastError());
    return CID_OK;
}

// The sorting path for the reads listed in h_route (all of them: NULL): it walks the offsets on the host — a device-pointer caller's come down first.
static int long_sorted_for(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0, size_t n_reads,
                           uint32_t stride_d, uint32_t start_sample, const uint8_t *h_route, bool clear_wide, uint32_t *d_report, uint32_t *d_n_kmers,
                           uint8_t *d_status, const StripePass &sp, const uint64_t *h_seq_off, const uint64_t *h_read_seq0, bool merge_status) {
    std::vector<uint64_t> so, r0;
    if (!h_seq_off || !h_read_seq0) {
        hipStream_t st = ctx_stream(c);
        r0.resize(n_reads + 1);
        HIP_TRY(hipMemcpyAsync(r0.data(), d_read_seq0, (n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        so.resize(r0[n_reads] + 1);
        HIP_TRY(hipMemcpyAsync(so.data(), d_seq_off, so.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h_seq_off = so.data(); h_read_seq0 = r0.data();
    }
    return readid_long_sorted(c, ix, d_bases, h_seq_off, h_read_seq0, n_reads, stride_d, start_sample, h_route, clear_wide, d_report, d_n_kmers, d_status, sp,
                              merge_status);
}

int readid_long(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0, size_t n_reads,
                uint32_t stride_d, uint32_t start_sample, const uint8_t *d_route, bool clear_wide, uint32_t *d_report, uint32_t *d_n_kmers,
                uint8_t *d_status, const StripePass &sp, const uint64_t *h_seq_off, const uint64_t *h_read_seq0, const uint32_t *d_route_stats,
                uint32_t *route_stats) {
    const uint32_t k = index_k(ix);
    hipStream_t st = ctx_stream(c);
    if (n_reads >= (1ull << 31)) return fail(CID_ERR_UNSUPPORTED, "more than 2^31 reads in one batch");
    if (k > 32 || !c->tune.readid_long_lds) {   // byte-string keys (or the measurement switch): every routed read through the sorting path
        std::vector<uint8_t> h_route;
        if (d_route) {
            h_route.resize(n_reads);
            HIP_TRY(hipMemcpyAsync(h_route.data(), d_route, n_reads, hipMemcpyDeviceToHost, st));
            if (d_route_stats) HIP_TRY(hipMemcpyAsync(route_stats, d_route_stats, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (uint8_t &b : h_route) b = b == 1 ? 1 : 0;
        }
        return long_sorted_for(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, d_route ? h_route.data() : nullptr, clear_wide, d_report,
                               d_n_kmers, d_status, sp, h_seq_off, h_read_seq0, false);
    }
    const uint32_t msz = index_m_size(ix);
    const uint32_t key_len = msz ? msz : k;
    const uint64_t sentinel_k = k < 32 ? (1ull << (2 * k)) : ~0ull;
    const uint64_t sentinel = key_len < 32 ? (1ull << (2 * key_len)) : ~0ull;
    const uint32_t C = index_n_colors(ix), rs = index_rs(ix), n_hash = index_n_hash(ix);
    const bool own_search = rs <= 128 && !sp.on();   // k_readid_slices; else the lists feed k_readid_list
    const bool cut = start_sample <= 64;              // a later slice gathers the first S k-mers again
    const size_t C1 = (size_t)C + 1;
    const uint64_t n = n_reads + 1;
    const uint64_t tiles = (n + kPlanTile - 1) / kPlanTile;
    // ---- the plan
    DevBuf<uint64_t> d_wstart(c), d_wend(c), d_pre(c), d_state(c), d_totals(c);
    DevBuf<uint32_t> d_win(c);
    DevBuf<uint8_t> d_cls(c), d_redo(c);
    DevBuf<int> d_flags(c);
    int rc;
    if ((rc = d_wstart.alloc(n)) || (rc = d_wend.alloc(n)) || (rc = d_pre.alloc((size_t)kPlanFields * n)) || (rc = d_state.alloc((size_t)kPlanFields * (tiles + 2))) ||
        (rc = d_totals.alloc(kPlanFields + 1)) || (rc = d_win.alloc(n)) || (rc = d_cls.alloc(n)) || (rc = d_redo.alloc(n)) || (rc = d_flags.alloc(4)))
        return rc;
    LongPlanParams pp{};
    pp.seq_off = d_seq_off; pp.read_seq0 = d_read_seq0; pp.route = d_route; pp.bases = d_bases; pp.n_reads = n_reads;
    pp.k = k; pp.stride = stride_d; pp.seg_win = kSegWindows / stride_d ? kSegWindows / stride_d : 1;
    pp.multi = c->tune.readid_long_multi ? 1u : 0u;
    pp.tiny = c->tune.readid_long_tiny ? 1u : 0u;
    pp.fuse = c->tune.readid_long_fuse ? 1u : 0u; pp.cut = cut ? 1u : 0u; pp.own_search = own_search ? 1u : 0u; pp.deal = c->tune.readid_long_deal ? 1u : 0u;
    pp.wstart = d_wstart.p; pp.wend = d_wend.p; pp.win = d_win.p; pp.cls = d_cls.p; pp.redo = d_redo.p; pp.status = d_status;
    pp.pre = d_pre.p; pp.state = d_state.p; pp.totals = d_totals.p; pp.flags = d_flags.p;
    HIP_TRY(hipMemsetAsync(d_state.p, 0, (size_t)kPlanFields * (tiles + 2) * 8, st));
    HIP_TRY(hipMemsetAsync(d_flags.p, 0, 16, st));
    hipLaunchKernelGGL(k_long_plan, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, pp);
    HIP_TRY(hipGetLastError());
    uint64_t t[kPlanFields];
    HIP_TRY(hipMemcpyAsync(t, d_totals.p, sizeof(t), hipMemcpyDeviceToHost, st));
    if (d_route_stats) HIP_TRY(hipMemcpyAsync(route_stats, d_route_stats, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));   // the one wait before the kernels: the lists' sizes
    const uint64_t W = t[0] & ((1ull << kPlanTinyShift) - 1ull), n_pairs = t[6] & ((1ull << kPlanMultiShift) - 1ull);
    const uint32_t n_ft = (uint32_t)(t[0] >> kPlanTinyShift);
    const uint32_t n_fm = (uint32_t)(t[6] >> kPlanMultiShift);
    if (W >= (1ull << 32) - 64) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    const uint32_t n_fs = (uint32_t)t[1], n_fb = (uint32_t)(t[1] >> 32), n_is = (uint32_t)t[2], n_ib = (uint32_t)(t[2] >> 32);
    const uint32_t n_slices = (uint32_t)t[3], n_combs = (uint32_t)(t[3] >> 32), n_segs = (uint32_t)t[4], n_deals = (uint32_t)(t[4] >> 32);
    const uint32_t n_chunks = (uint32_t)t[5], n_deal_counts = (uint32_t)(t[5] >> 32);
    const uint64_t n_words = W / 32 + 1;   // (rank(W) reads the word after the last window's)
    DevBuf<uint64_t> d_codes(c), d_list(c), d_lstart(c), d_scan(c), d_pair_code(c);
    DevBuf<uint32_t> d_bitmap(c), d_prefix(c), d_partial(c), d_pair_idx(c), d_deal_counts(c), d_lists32(c);
    DevBuf<LongItem> d_items(c);
    DevBuf<FuseItem> d_fuse(c);
    DevBuf<ReadSlice> d_slices(c);
    DevBuf<ReadCombine> d_combs(c);
    DevBuf<Segment> d_segs(c);
    DevBuf<LongDeal> d_deals(c);
    if ((rc = d_codes.alloc(W + 1)) || (rc = d_scan.alloc(scan_state_words(n_words))) || (rc = d_bitmap.alloc(n_words)) || (rc = d_prefix.alloc(n_words)) ||
        (rc = d_partial.alloc(n_combs ? (size_t)n_slices * (C1 + 1) : 1)) || (rc = d_pair_code.alloc(n_pairs)) || (rc = d_pair_idx.alloc(n_pairs)) ||
        (rc = d_deal_counts.alloc(n_deal_counts)) || (rc = d_lists32.alloc((size_t)n_segs + 2 * (size_t)n_chunks)) || (rc = d_fuse.alloc((size_t)n_ft + n_fs + n_fb + n_fm)) ||
        (rc = d_items.alloc((size_t)n_is + n_ib)) || (rc = d_slices.alloc(n_slices)) || (rc = d_combs.alloc(n_combs)) || (rc = d_segs.alloc(n_segs)) ||
        (rc = d_deals.alloc(n_deals)))
        return rc;
    // soft-masked reads of the fused classes are redone on the device (k_long_bytes + k_readid_list over byte strings): whole k-mers, rows of
    // at most 1 KiB, no stripe pass — elsewhere they take the sorting path
    const bool bytes_on_device = own_search && !msz && (n_ft + n_fs + n_fb) > 0;
    const uint32_t n_fused = n_ft + n_fs + n_fb;
    DevBuf<uint32_t> d_lower(c);
    DevBuf<uint8_t> d_bytes_read(c);
    if (bytes_on_device) {
        if ((rc = d_lower.alloc((size_t)n_fused + 4)) || (rc = d_bytes_read.alloc(n_reads))) return rc;
        HIP_TRY(hipMemsetAsync(d_lower.p, 0, 16, st));
        HIP_TRY(hipMemsetAsync(d_bytes_read.p, 0, n_reads, st));
    }
    LongLists L{};
    L.fused_tiny = d_fuse.p; L.fused_small = d_fuse.p + n_ft; L.fused_big = L.fused_small + n_fs; L.fused_multi = L.fused_big + n_fb; L.seg_read = d_lists32.p; L.chunk_deal = L.seg_read + n_segs; L.chunk_no = L.chunk_deal + n_chunks;
    L.items_small = d_items.p; L.items_big = d_items.p + n_is;
    L.slices = d_slices.p; L.combs = d_combs.p; L.segs = d_segs.p; L.deals = d_deals.p;
    if (n_ft + n_fs + n_fb + n_fm) hipLaunchKernelGGL(k_long_emit<1u>, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, pp, L);
    if (n_is + n_ib) hipLaunchKernelGGL(k_long_emit<64u>, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, pp, L);
    HIP_TRY(hipMemsetAsync(d_bitmap.p, 0, n_words * 4, st));
    HIP_TRY(hipGetLastError());
    int h_flags[4] = {0, 0, 0, 0};
    const unsigned n_cu = (unsigned)ctx_n_cu(c);
    if (W) {
        if (n_segs) {   // the reads of several buckets, and those whose bases do not fit the fused kernel: codes first, then the tables
            constexpr uint32_t kBytes = kSegWindows + 32 + 96;
            const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
            unsigned grid = (n_segs + 3) / 4;
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, d_bases, (const Segment *)d_segs.p, n_segs, k, 1, sentinel_k, d_codes.p,
                               d_flags.p, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr, KeyFor{},
                               (const uint32_t *)L.seg_read, d_redo.p);
            // (minimizer indices: over the whole numbering — what the fused kernel writes afterwards are minimizer codes already)
            if (msz) hipLaunchKernelGGL(k_codes_to_minimizers, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, (uint64_t)W, k, msz, sentinel_k, sentinel);
            if (n_chunks) {
                unsigned g = n_chunks;
                if (g > n_cu * 8u) g = n_cu * 8u;
                hipLaunchKernelGGL(k_long_deal, dim3(g), dim3(kDealBlock), 0, st, d_codes.p, d_wstart.p, d_wend.p, (const LongDeal *)d_deals.p,
                                   (const uint32_t *)L.chunk_deal, (const uint32_t *)L.chunk_no, n_chunks, sentinel, d_pair_code.p, d_pair_idx.p, d_deal_counts.p,
                                   d_flags.p, d_redo.p, d_bitmap.p);
            }
            if (n_is) {
                unsigned g = n_cu * 4u;   // four 32-KiB workgroups per CU
                g = (g + 7u) & ~7u;
                hipLaunchKernelGGL(k_long_first_flags, dim3(g), dim3(kLongBlockSmall), (kLongSlotsSmall + kLongBmSmall) * 4, st, d_codes.p, d_wstart.p, d_wend.p,
                                   (const LongItem *)L.items_small, n_is, sentinel, kLongSlotsSmall, kLongBmSmall, d_bitmap.p, d_flags.p, (const LongDeal *)d_deals.p,
                                   d_pair_code.p, d_pair_idx.p, d_deal_counts.p, d_redo.p);
            }
            if (n_ib) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_long_first_flags), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)((kLongSlotsBig + kLongBmBig) * 4)));
                unsigned g = (n_cu + 7u) & ~7u;
                hipLaunchKernelGGL(k_long_first_flags, dim3(g), dim3(kLongBlockBig), (kLongSlotsBig + kLongBmBig) * 4, st, d_codes.p, d_wstart.p, d_wend.p,
                                   (const LongItem *)L.items_big, n_ib, sentinel, kLongSlotsBig, kLongBmBig, d_bitmap.p, d_flags.p, (const LongDeal *)d_deals.p,
                                   d_pair_code.p, d_pair_idx.p, d_deal_counts.p, d_redo.p);
#ifdef CID_LONG_PROF
                unsigned long long hp[8];
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_ff_prof), sizeof(hp)));
                fprintf(stderr, "k_long_first_flags<big> %u items, cycles per item: clear %llu, setup %llu, fetch %llu, inserts %llu, drain %llu, barrier %llu\n", n_ib, hp[0] / n_ib,
                        hp[1] / n_ib, hp[2] / n_ib, hp[3] / n_ib, hp[5] / n_ib, hp[4] / n_ib);
                unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ff_prof), z, sizeof(z)));
#endif
            }
        }
        LongFuseParams fp{};
        fp.k = k; fp.msz = msz; fp.stride = stride_d; fp.sentinel = sentinel; fp.codes = d_codes.p; fp.bitmap = d_bitmap.p; fp.redo = d_redo.p; fp.flags = d_flags.p;
        if (bytes_on_device) { fp.lower_list = d_lower.p + 4; fp.n_lower = d_lower.p; }
        if (n_ft) {
            fp.items = L.fused_tiny; fp.n_list = n_ft; fp.big = 0; fp.item_base = 0; fp.max_slots = kLongSlotsTiny; fp.pos_cap = kFusePosTiny; fp.bm_words = kLongTinyWin / 32;
            unsigned g = (n_cu * 16u + 7u) & ~7u;   // sixteen waves to a CU
            hipLaunchKernelGGL((k_long_fused<kLongBlockTiny, kLongTinyWin / kLongBlockTiny>), dim3(g), dim3(kLongBlockTiny),
                               (kLongSlotsTiny + kFusePosTiny / 16 + 4 + kFusePosTiny / 32 + 4 + kLongTinyWin / 32) * 4, st, fp);
        }
        if (n_fs) {
            fp.item_base = n_ft;
            fp.items = L.fused_small; fp.n_list = n_fs; fp.big = 0; fp.max_slots = kLongSlotsSmall; fp.pos_cap = kFusePosSmall; fp.bm_words = kLongSmallWin / 32;
            unsigned g = (n_cu * 4u + 7u) & ~7u;
            hipLaunchKernelGGL((k_long_fused<kLongBlockSmall, kLongSmallWin / kLongBlockSmall>), dim3(g), dim3(kLongBlockSmall),
                               (kLongSlotsSmall + kFusePosSmall / 16 + 4 + kFusePosSmall / 32 + 4 + kLongSmallWin / 32) * 4, st, fp);
        }
#ifdef CID_LONG_PROF
        DevBuf<unsigned long long> d_prof(c);
        if ((rc = d_prof.alloc((size_t)(n_cu + 8) * 8))) return rc;
        HIP_TRY(hipMemsetAsync(d_prof.p, 0, (size_t)(n_cu + 8) * 64, st));
        fp.prof = n_fb ? d_prof.p : nullptr;
#endif
        if (n_fb) {
            fp.item_base = 0;
            fp.items = L.fused_big; fp.n_list = n_fb; fp.big = 1; fp.max_slots = kLongSlotsBig; fp.pos_cap = kFusePosBig; fp.bm_words = kLongFill / 32;
            const int shmem = (int)((kLongSlotsBig + kFusePosBig / 16 + 4 + kFusePosBig / 32 + 4 + kLongFill / 32) * 4);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_long_fused<kLongBlockBig, kLongFill / kLongBlockBig>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
            unsigned g = (n_cu + 7u) & ~7u;
            hipLaunchKernelGGL((k_long_fused<kLongBlockBig, kLongFill / kLongBlockBig>), dim3(g), dim3(kLongBlockBig), shmem, st, fp);
#ifdef CID_LONG_PROF
            std::vector<unsigned long long> hp((size_t)g * 8);
            HIP_TRY(hipMemcpyAsync(hp.data(), d_prof.p, hp.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t b = 0; b < g; ++b) for (int i = 0; i < 8; ++i) sum[i] += hp[b * 8 + i];
            fprintf(stderr, "k_long_fused<big> %u reads, mean cycles per workgroup: clears %llu, staging %llu, barrier %llu, inserts %llu, wait %llu, winners %llu, loop top %llu\n", n_fb,
                    sum[0] / g, sum[1] / g, sum[2] / g, sum[3] / g, sum[4] / g, sum[5] / g, sum[7] / g);
#endif
        }
        if (n_fm) {   // (a lower-case base in one of these reads: the sorting path at the end of the call)
            fp.items = L.fused_multi; fp.n_list = n_fm; fp.big = 1; fp.max_slots = kLongSlotsBig; fp.pos_cap = kFusePosMulti; fp.bm_words = kFuseWinMulti / 32;
            fp.lower_list = nullptr; fp.n_lower = nullptr; fp.prof = nullptr;
            const int shmem = (int)((kLongSlotsBig + kFusePosMulti / 16 + 4 + kFusePosMulti / 32 + 4 + kFuseWinMulti / 32) * 4);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_long_fused<kLongBlockBig, kLongFill / kLongBlockBig, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
            unsigned g = (n_cu + 7u) & ~7u;
            hipLaunchKernelGGL((k_long_fused<kLongBlockBig, kLongFill / kLongBlockBig, true>), dim3(g), dim3(kLongBlockBig), shmem, st, fp);
        }
        if (bytes_on_device) {
            LongBytesParams bp{};
            // (the shortest reads' list lies before the small ones': item_base)
            bp.items_small = L.fused_tiny; bp.items_big = L.fused_big; bp.lower = d_lower.p + 4; bp.n_lower = d_lower.p; bp.bases = d_bases;
            bp.k = k; bp.stride = stride_d; bp.codes = d_codes.p; bp.bitmap = d_bitmap.p; bp.bytes_read = d_bytes_read.p;
            bp.redo = d_redo.p; bp.flags = d_flags.p;
            const int shmem = (int)((kLongSlotsBig + kFusePosBig / 16 + 4 + 2 * (kFusePosBig / 32 + 4) + kLongFill / 32) * 4);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_long_bytes), hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
            unsigned g = n_fused < n_cu ? n_fused : n_cu;
            hipLaunchKernelGGL(k_long_bytes, dim3(g), dim3(kBytesBlock), shmem, st, bp);
        }
        HIP_TRY(hipGetLastError());
    }
    if (!own_search) {   // wide rows and stripe passes add into rows in place: the reads the sorting path will redo are taken out before anything is counted
        HIP_TRY(hipMemcpyAsync(h_flags, d_flags.p, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_flags[0] || h_flags[1] || h_flags[2]) {
            hipLaunchKernelGGL(k_long_redo_status, dim3(grid_for_n(n_reads)), dim3(256), 0, st, d_redo.p, d_status, (uint64_t)n_reads);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(scan_launch(PopcIn{d_bitmap.p}, PrefixOut{d_prefix.p}, n_words, d_scan.p, st));
    const uint32_t hist_pad = rs > 128 ? 4u * rs : (uint32_t)((C1 + 3) & ~(size_t)3);
    const uint32_t wave_bytes = (uint32_t)((4ull * kWave * n_hash + 4ull * hist_pad + 15) & ~15ull);
    if ((size_t)(kBlock / kWave) * wave_bytes > 160u * 1024u) return fail(CID_ERR_UNSUPPORTED, "LDS need exceeds 160 KiB");
    if (own_search) {
        ReadIdSliceParams p{};
        p.mat = index_matrix(ix); p.rs = rs; p.w64 = (C + 63) / 64; p.n_colors = C; p.n_hash = n_hash; p.k = key_len; p.mod = index_mod(ix);
        p.codes = d_codes.p; p.wstart = d_wstart.p; p.wend = d_wend.p; p.bitmap = d_bitmap.p; p.word_prefix = d_prefix.p;
        p.slices = d_slices.p; p.n_slices = n_slices; p.start_sample = start_sample;
        p.hist_pad = hist_pad; p.wave_bytes = wave_bytes;
        p.report = d_report; p.n_kmers = d_n_kmers; p.partial = d_partial.p;
        uint64_t grid = ((uint64_t)n_slices + 3) / 4;
        const uint64_t cap = (uint64_t)ctx_n_cu(c) * 32;
        if (grid > cap) grid = cap;
        if (bytes_on_device) { p.bytes_read = d_bytes_read.p; p.bases = d_bases; }
        HIP_TRY(launch_readid_slices(p, (int)grid, st));
        if (bytes_on_device) HIP_TRY(launch_readid_slices(p, (int)grid, st, true));   // the soft-masked reads' slices (none, as a rule: every wave leaves at once)
        HIP_TRY(launch_readid_combine(d_combs.p, n_combs, d_partial.p, C, d_report, st));
        hipLaunchKernelGGL(k_long_short_rows, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, d_status, (uint32_t)n_reads, C, d_report, d_n_kmers);
        HIP_TRY(hipGetLastError());
    } else {   // k_readid_list walks lists
        if ((rc = d_list.alloc(W + 1)) || (rc = d_lstart.alloc(n_reads + 1))) return rc;
        if (W) hipLaunchKernelGGL(k_long_scatter, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, d_bitmap.p, d_prefix.p, d_list.p, (uint64_t)W);
        hipLaunchKernelGGL(k_long_list_starts, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, d_wstart.p, d_bitmap.p, d_prefix.p, d_lstart.p,
                           (uint32_t)n_reads);
        HIP_TRY(hipGetLastError());
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
    // The wait at the end of the call: the scratch above leaves scope, and three facts only the kernels know send single reads to the sorting
    // path — a lower-case base (its case is kept: byte-string keys), a table that crowded seven levels deep, more windows than a slot numbers.
    HIP_TRY(hipMemcpyAsync(h_flags, d_flags.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_flags[0] || h_flags[1] || h_flags[2]) {
        std::vector<uint8_t> h_redo(n_reads);
        HIP_TRY(hipMemcpyAsync(h_redo.data(), d_redo.p, n_reads, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return long_sorted_for(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, h_redo.data(), false, d_report, d_n_kmers, d_status, sp,
                               h_seq_off, h_read_seq0, true);
    }
    return CID_OK;
}

}  // namespace cid
