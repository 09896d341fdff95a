// The run sort of k-mer sets with many copies of every k-mer (reads of an isolate at 30-100x coverage — the reference's everyday input):
// between k_run_bucket_sort (a key ranks itself against its bucket's members: c comparisons per key in a bucket of c) and k_run_sort
// (LSD radix in LDS: six to eight passes whatever the keys) sits k_run_dedupe_sort, for the runs the bucket kernel finds crowded.
// A bucket's stretch of LDS first serves as an open-addressing table of its DISTINCT keys: a key starts at a hashed place of the stretch,
// walks it (wrapping round) to the first empty slot, which it claims with a 64-bit compare-and-swap, or to its own value, and draws its
// number among the copies from that slot's counter — one or two probes and ONE contended atomic per key where the crowding is copies
// (in code order a true k-mer's bucket also holds its one-off error variants — a handful at 50x, dozens at 500x — which is why the walk
// does not start at the stretch's first slot there; a set built for an index scatters the variants by their first-row keys, and its
// walk does start at the first slot: the filled slots are then a prefix of the stretch and need no list).  Whoever claims a slot appends it
// to the bucket's list of distinct keys; those rank each other, weighted by their counts: that is where each one's copies start; a key's
// place is that plus its number.  The run leaves sorted, copies included,
// like the other run sorts' — the run-length count downstream (cid_rle.hpp) is unchanged.
//   A key that has not found its slot after kDedupeProbes steps (many DIFFERENT keys in one bucket: low-complexity sequence) gives the run
//   up: it is named in hard2 for k_run_sort, as are listed runs beyond this kernel's largest launched capacity (name_larger).
//   1 M reads of a 3 Mb genome (50x, 1 % errors: 33.8 M distinct of 120 M windows): see DESIGN.md §5.
#pragma once
#include "cid_partition.hpp"

namespace cid {

constexpr uint32_t kDedupeProbes = 48, kDedupeNdWords = (kBuckets + 2) / 3, kDedupeNdMax = 1023;

template <int MAXR, bool PAIR>
__global__ __launch_bounds__(kPartBlock) void k_run_dedupe_sort(const uint32_t *keys, const uint64_t *vals, uint64_t *out, const uint32_t *run_off, uint32_t n_runs, PairOrder ord,
                                                                 uint32_t bits, uint32_t min_size, uint32_t max_size, uint32_t name_larger, const uint32_t *list,
                                                                 const uint32_t *list_n, uint32_t *n_hard2, uint32_t *hard2) {
    constexpr uint32_t CAP = kPartBlock * MAXR;
    constexpr unsigned long long EMPTY = ~0ull;   // never a code: codes of k <= 31 stay below 2^62, the k = 32 path does not come here with all ones
    __shared__ uint64_t s_val[CAP];
    __shared__ uint32_t s_key[PAIR ? CAP : 1];
    // s_cnt: first the buckets' member counts; then per slot: its copies (low half) and, at the stretch's i-th place, the slot of the bucket's
    // i-th distinct key (high half); then per slot: the place of its first copy
    __shared__ uint32_t s_cnt[CAP];
    constexpr bool HASHED = !PAIR;               // where a key's walk starts, and how a bucket's distinct keys are found again (see above)
    __shared__ uint32_t s_nd[HASHED ? kDedupeNdWords : 4];   // distinct keys per bucket, three buckets to a word (ten bits each: a bucket of 1023 gives the run up);
                                                 // its first four words carry the scan's wave totals before that
    __shared__ uint16_t s_pre[kBuckets + 1];
    __shared__ uint32_t s_fail;
    uint32_t *s_wave = s_nd;
    static_assert(CAP >= kBuckets, "the bucket counts borrow the slot counters");
    static_assert(CAP <= 4096, "a slot number and a count share a word");
    constexpr uint32_t BPT = kBuckets / kPartBlock;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t bbits = 0, kmask = 0, bshift = 0, bmask = 0;
    if (PAIR) {
        const uint32_t total_bits = ord.kbits + ord.vbits;
        bbits = total_bits < kBucketBits ? total_bits : kBucketBits;
        kmask = ord.kmask();
    } else {
        bshift = bits > kBucketBits ? bits - kBucketBits : 0u;
        bmask = bits >= kBucketBits ? kBuckets - 1u : ((1u << bits) - 1u);
    }
    auto bucket_of = [&](uint32_t k, uint64_t v) -> uint32_t { return PAIR ? ord.bucket(k, v, bbits) : ((uint32_t)(v >> bshift) & bmask); };
    const uint32_t n_todo = list ? *list_n : n_runs;   // list: the runs the bucket kernels named; none: every run (a batch found crowded by sampling)
    for (uint32_t at = blockIdx.x; at < n_todo; at += gridDim.x) {
        const uint32_t run = list ? list[at] : at;
        const uint32_t start = run_off[run], N = run_off[run + 1] - start;
        if (N < min_size) continue;   // (uniform over the block) a smaller instantiation's run, or an empty one
        if (N > max_size) {
            if (name_larger && threadIdx.x == 0) hard2[atomicAdd(n_hard2, 1u)] = run;
            continue;
        }
#pragma unroll
        for (uint32_t j = 0; j < BPT; ++j) s_cnt[threadIdx.x * BPT + j] = 0;
        if (threadIdx.x == 0) s_fail = 0;
        __syncthreads();
        uint64_t val[MAXR];
        uint32_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            key[r] = (PAIR && p < N) ? keys[start + p] & kmask : 0u;
            val[r] = p < N ? vals[start + p] : 0ull;
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) atomicAdd(&s_cnt[bucket_of(key[r], val[r])], 1u);
        __syncthreads();
        {   // exclusive prefix over the bucket counts: a bucket's stretch has room for every member, so also for its distinct ones
            uint32_t c[BPT], sum = 0;
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { c[j] = s_cnt[threadIdx.x * BPT + j]; sum += c[j]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
            if (lane == 63) s_wave[w] = incl;
            __syncthreads();
            uint32_t base = incl - sum;
            for (uint32_t ww = 0; ww < w; ++ww) base += s_wave[ww];
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { s_pre[threadIdx.x * BPT + j] = (uint16_t)base; base += c[j]; }
            if (threadIdx.x == kPartBlock - 1) s_pre[kBuckets] = (uint16_t)base;
        }
        __syncthreads();   // (the wave totals have been read: their words go back to s_nd)
        for (uint32_t p = threadIdx.x; p < CAP; p += kPartBlock) { s_val[p] = EMPTY; s_cnt[p] = 0; }
        if (HASHED)
            for (uint32_t p = threadIdx.x; p < kDedupeNdWords; p += kPartBlock) s_nd[p] = 0;
        __syncthreads();
        uint16_t my[MAXR];     // the slot of this key's value
        uint32_t ord_[MAXR];   // ... and which of the value's copies it is (the order the slot's counter handed out)
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            my[r] = 0;
            ord_[r] = 0;
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) {
                const uint32_t bkt = bucket_of(key[r], val[r]);
                const uint32_t lo = s_pre[bkt], hi = s_pre[bkt + 1], c = hi - lo;
                // where the walk starts: 24 hashed bits scaled to the stretch's length
                uint32_t slot = HASHED ? lo + (uint32_t)(((uint64_t)((uint32_t)((val[r] * 0x9E3779B97F4A7C15ull) >> 40) & 0xFFFFFFu) * c) >> 24) : lo;   // (64-bit: c passes 256)
                bool found = false;
                const uint32_t limit = c < kDedupeProbes ? c : kDedupeProbes;
                if (val[r] != EMPTY)   // (a value that looks like an empty slot — no k <= 31 code does — sends the run to the radix kernel)
                for (uint32_t probes = 0; probes < limit; ++probes) {
                    // a plain read first: the copies that come after a value's first one only read (same-address reads are a broadcast,
                    // same-address atomics queue up; a slot never changes once it holds a value).  Only "it is mine" is taken from the
                    // plain read — anything else, a half-written slot included, is settled by the compare-and-swap.
                    unsigned long long old = *reinterpret_cast<volatile unsigned long long *>(&s_val[slot]);
                    if (old != val[r]) old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_val[slot]), EMPTY, (unsigned long long)val[r]);
                    if (old == EMPTY) {   // claimed: the bucket's next distinct key
                        if (PAIR) s_key[slot] = key[r];
                        if (HASHED) {
                            const uint32_t word = bkt / 3u, sh = (bkt - word * 3u) * 10u;
                            const uint32_t nd = (atomicAdd(&s_nd[word], 1u << sh) >> sh) & 0x3FFu;
                            if (nd >= kDedupeNdMax - 1) { s_fail = 1; break; }   // (the field would run over into its neighbour's)
                            atomicOr(&s_cnt[lo + nd], slot << 16);
                        }
                        found = true;
                        break;
                    }
                    if (old == val[r]) { found = true; break; }
                    if (++slot == hi) slot = lo;   // (not HASHED: the walk has been through the whole stretch by then — limit)
                }
                if (found) { ord_[r] = atomicAdd(&s_cnt[slot], 1u) & 0xFFFFu; my[r] = (uint16_t)slot; }
                else s_fail = 1;
            }
        }
        __syncthreads();
        if (s_fail) {   // (uniform) many different keys in one bucket: the radix kernel's run
            if (threadIdx.x == 0) hard2[atomicAdd(n_hard2, 1u)] = run;
            __syncthreads();
            continue;
        }
        uint32_t first[MAXR];   // slot p holds a distinct key: the place of its first copy = its bucket's start + the copies of the bucket's smaller keys
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            first[r] = 0xFFFFFFFFu;
            if (p < N && s_val[p] != EMPTY) {
                const uint64_t v = s_val[p];
                const uint32_t k = PAIR ? s_key[p] : 0u;
                const uint32_t bkt = bucket_of(k, v);
                const uint32_t lo = s_pre[bkt];
                uint32_t off = 0;
                if (HASHED) {
                    const uint32_t word = bkt / 3u;
                    const uint32_t nd = (s_nd[word] >> ((bkt - word * 3u) * 10u)) & 0x3FFu;
                    for (uint32_t i = 0; i < nd; ++i) {
                        const uint32_t q = s_cnt[lo + i] >> 16;
                        off += s_val[q] < v ? (s_cnt[q] & 0xFFFFu) : 0u;
                    }
                } else {
                    const uint32_t hi = s_pre[bkt + 1];
                    for (uint32_t q = lo; q < hi; ++q) {
                        const uint64_t ov = s_val[q];
                        if (ov == EMPTY) break;   // the filled slots are a prefix of the stretch
                        const bool less = s_key[q] < k || (s_key[q] == k && ov < v);
                        off += less ? (s_cnt[q] & 0xFFFFu) : 0u;
                    }
                }
                first[r] = lo + off;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if (first[r] != 0xFFFFFFFFu) s_cnt[(uint32_t)r * kPartBlock + threadIdx.x] = first[r];
        __syncthreads();
        uint32_t dest[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            dest[r] = 0;
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) dest[r] = s_cnt[my[r]] + ord_[r];
        }
        __syncthreads();   // (the table in s_val has been read for the last time above)
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) s_val[dest[r]] = val[r];
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < N; p += kPartBlock) out[start + p] = s_val[p];
        __syncthreads();
    }
}


// Is this batch one of copies?  kCrowdSample runs, evenly spaced, are counted into their buckets as the run sorts would; out[0] += runs looked
// at (those of 64 .. 2048 keys), out[1] += those whose buckets would cost a key more than crowded_at comparisons on average.  The host then
// sends EVERY run through k_run_dedupe_sort instead of letting the bucket kernels count each crowded run only to hand it on.
constexpr uint32_t kCrowdSample = 128;
template <bool PAIR>
__global__ __launch_bounds__(kPartBlock) void k_run_crowd_sample(const uint32_t *keys, const uint64_t *vals, const uint32_t *run_off, uint32_t n_runs, PairOrder ord,
                                                                  uint32_t bits, uint32_t crowded_at, uint32_t *out) {
    __shared__ uint32_t s_cnt[kBuckets];
    __shared__ uint32_t s_sq[4];
    const uint32_t run = (uint32_t)(((uint64_t)blockIdx.x * n_runs) / gridDim.x);
    const uint32_t start = run_off[run], N = run_off[run + 1] - start;
    if (N < 64 || N > 2048) return;
    constexpr uint32_t BPT = kBuckets / kPartBlock;
    uint32_t bbits = 0, kmask = 0, bshift = 0, bmask = 0;
    if (PAIR) {
        const uint32_t total_bits = ord.kbits + ord.vbits;
        bbits = total_bits < kBucketBits ? total_bits : kBucketBits;
        kmask = ord.kmask();
    } else {
        bshift = bits > kBucketBits ? bits - kBucketBits : 0u;
        bmask = bits >= kBucketBits ? kBuckets - 1u : ((1u << bits) - 1u);
    }
#pragma unroll
    for (uint32_t j = 0; j < BPT; ++j) s_cnt[threadIdx.x * BPT + j] = 0;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < N; p += kPartBlock) {
        const uint64_t v = vals[start + p];
        const uint32_t k = PAIR ? keys[start + p] & kmask : 0u;
        atomicAdd(&s_cnt[PAIR ? ord.bucket(k, v, bbits) : ((uint32_t)(v >> bshift) & bmask)], 1u);
    }
    __syncthreads();
    uint32_t sq = 0;
#pragma unroll
    for (uint32_t j = 0; j < BPT; ++j) { const uint32_t c = s_cnt[threadIdx.x * BPT + j]; sq += c * c; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], 1u);
        if (s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3] > crowded_at * N) atomicAdd(&out[1], 1u);
    }
}

}  // namespace cid
