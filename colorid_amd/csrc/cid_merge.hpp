// Two sorted lists of distinct (k-mer, multiplicity) pairs merged into one, equal k-mers' multiplicities added — how a later batch
// joins a k-mer set that already holds k-mers (kmer.rs:87-125 / :461-510 count into ONE map; cid_kmerset.hip counts a batch at a time).
// Merge path: the plain merge of A and B (n_a + n_b elements) is cut into tiles of kMergeTile outputs; k_merge_partition finds, per tile
// boundary, how many elements of A and of B lie before it (a binary search along the diagonal; ties: A first, and a pair of equal keys is
// never cut apart); k_merge_tiles takes the tiles in ticket order: its two stretches meet in LDS, every thread merges kMergePer outputs of its
// own, equal neighbours are joined (always an A element followed by its B twin), and the tile's number of distinct keys goes through the
// decoupled look-back of cid_scan.hpp, so that every tile knows where its output starts — one pass over the data, no rocPRIM.
// Order: (key32, code) — key32 is the first-row key of a set built for an index (cid_kmerset_set_target_index), or absent (NULL: code order).
#pragma once
#include "cid_scan.hpp"

namespace cid {

constexpr uint32_t kMergePer = 8, kMergeTile = kScanBlock * kMergePer;   // 2048 outputs of the plain merge per tile

struct MergeKey {
    uint32_t k32;
    uint64_t code;
};
__device__ __forceinline__ bool merge_less(const MergeKey &x, const MergeKey &y) { return x.k32 < y.k32 || (x.k32 == y.k32 && x.code < y.code); }
__device__ __forceinline__ bool merge_eq(const MergeKey &x, const MergeKey &y) { return x.k32 == y.k32 && x.code == y.code; }
__device__ __forceinline__ MergeKey merge_key(const uint32_t *k32, const uint64_t *codes, uint64_t i) { return MergeKey{k32 ? k32[i] : 0u, codes[i]}; }

// split[t] = elements of A before output t * kMergeTile of the plain merge (B: t * kMergeTile - split[t], + 1 where a twin was pulled over);
// split_b[t] = elements of B before it.  t = 0 .. n_tiles (the last entry = n_a, n_b).
__global__ void k_merge_partition(const uint32_t *ka, const uint64_t *a, uint64_t n_a, const uint32_t *kb, const uint64_t *b, uint64_t n_b, uint32_t n_tiles,
                                  uint64_t *split_a, uint64_t *split_b) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint64_t d = (uint64_t)t * kMergeTile;
    if (d > n_a + n_b) d = n_a + n_b;
    uint64_t lo = d > n_b ? d - n_b : 0, hi = d < n_a ? d : n_a;   // i = elements of A among the first d: the largest i with A[i-1] <= B[d-i]
    while (lo < hi) {
        const uint64_t i = (lo + hi + 1) >> 1;   // try to take i elements of A: fine iff A[i-1] <= B[d-i] (or B exhausted)
        const uint64_t j = d - i;
        if (j >= n_b || !merge_less(merge_key(kb, b, j), merge_key(ka, a, i - 1))) lo = i; else hi = i - 1;
    }
    uint64_t i = lo, j = d - lo;
    if (i > 0 && j < n_b && merge_eq(merge_key(ka, a, i - 1), merge_key(kb, b, j))) ++j;   // the twin of A's last element stays with it
    split_a[t] = i;
    split_b[t] = j;
}

// state: scan_state_words(n_tiles * kScanTile) is more than enough — the kernel uses n_tiles + 2 words (zeroed by the launcher)
__global__ __launch_bounds__(kScanBlock) void k_merge_tiles(const uint32_t *ka, const uint64_t *a, const uint32_t *ca, const uint32_t *kb, const uint64_t *b,
                                                            const uint32_t *cb, const uint64_t *split_a, const uint64_t *split_b, uint32_t n_tiles, uint64_t *out_codes,
                                                            uint32_t *out_counts, uint64_t *state, int *saturated) {
    constexpr uint32_t CAP = kMergeTile + 2;
    __shared__ uint64_t s_code[CAP];
    __shared__ uint32_t s_k32[CAP], s_cnt[CAP];
    __shared__ uint32_t s_first_dup[kScanBlock + 1];   // count carried by a thread's first element when that is the twin of the thread before's last
    __shared__ uint64_t s_last_code[kScanBlock];
    __shared__ uint32_t s_last_k32[kScanBlock];
    const uint64_t tile = scan_ticket(state, n_tiles);
    if (tile >= n_tiles) return;
    const uint64_t a0 = split_a[tile], a1 = split_a[tile + 1], b0 = split_b[tile], b1 = split_b[tile + 1];
    const uint32_t na = (uint32_t)(a1 - a0), nb = (uint32_t)(b1 - b0), n = na + nb;   // <= kMergeTile + 1
    for (uint32_t i = threadIdx.x; i < n; i += kScanBlock) {   // A's stretch, then B's
        const bool from_a = i < na;
        const uint64_t g = from_a ? a0 + i : b0 + (i - na);
        s_code[i] = from_a ? a[g] : b[g];
        s_k32[i] = from_a ? (ka ? ka[g] : 0u) : (kb ? kb[g] : 0u);
        s_cnt[i] = from_a ? ca[g] : cb[g];
    }
    __syncthreads();
    // this thread's outputs [d0, d1) of the tile's plain merge: its own diagonal search inside LDS
    const uint32_t per = (n + kScanBlock - 1) / kScanBlock;   // <= kMergePer + 1
    const uint32_t d0 = threadIdx.x * per < n ? threadIdx.x * per : n, d1 = d0 + per < n ? d0 + per : n;
    auto key_at = [&](uint32_t i) { return MergeKey{s_k32[i], s_code[i]}; };
    uint32_t lo = d0 > nb ? d0 - nb : 0, hi = d0 < na ? d0 : na;
    while (lo < hi) {
        const uint32_t i = (lo + hi + 1) >> 1, j = d0 - i;
        if (j >= nb || !merge_less(key_at(na + j), key_at(i - 1))) lo = i; else hi = i - 1;
    }
    uint32_t i = lo, j = d0 - lo;
    MergeKey k[kMergePer + 1];
    uint32_t c[kMergePer + 1];
    uint32_t m = 0;
    for (uint32_t o = d0; o < d1; ++o, ++m) {
        const bool take_a = i < na && (j >= nb || !merge_less(key_at(na + j), key_at(i)));
        const uint32_t at = take_a ? i++ : na + j++;
        k[m] = key_at(at);
        c[m] = s_cnt[at];
    }
    // join twins inside the thread (an element equal to the one before it: its count moves there)
    uint32_t heads = 0;
    for (uint32_t q = 0; q < m; ++q) {
        if (q > 0 && merge_eq(k[q], k[q - 1])) {
            uint32_t hq = q - 1;
            while (!((heads >> hq) & 1u)) --hq;   // (the head of the run: twins come in pairs, so this is q - 1)
            const uint32_t s = c[hq] + c[q];
            c[hq] = s < c[q] ? 0xFFFFFFFFu : s;
        } else heads |= 1u << q;
    }
    // ... and across threads: my first element may be the twin of the previous thread's last
    s_last_code[threadIdx.x] = m ? k[m - 1].code : 0ull;
    s_last_k32[threadIdx.x] = m ? k[m - 1].k32 : 0u;
    s_first_dup[threadIdx.x + 1] = 0;
    if (threadIdx.x == 0) s_first_dup[0] = 0;
    __syncthreads();
    if (m && threadIdx.x > 0 && d0 > 0) {
        // the previous non-empty thread is threadIdx.x - 1 (threads are filled in order)
        const MergeKey prev{s_last_k32[threadIdx.x - 1], s_last_code[threadIdx.x - 1]};
        if (merge_eq(k[0], prev)) { heads &= ~1u; s_first_dup[threadIdx.x] = c[0]; }
    }
    __syncthreads();
    if (m) {   // the count my neighbour's first element hands to my last (which is a head of a run of one, or the head its own twin joined)
        const uint32_t extra = s_first_dup[threadIdx.x + 1];
        if (extra) {
            uint32_t hq = m - 1;
            while (!((heads >> hq) & 1u)) { if (hq == 0) break; --hq; }
            const uint32_t s = c[hq] + extra;
            c[hq] = s < extra ? 0xFFFFFFFFu : s;
        }
    }
    uint64_t tile_heads;
    const uint32_t local = (uint32_t)scan_block_exclusive((uint64_t)__popc(heads), &tile_heads);
    const uint64_t tile_excl = scan_lookback_block(state, tile, n_tiles, tile_heads);
    // the tile's distinct keys through LDS, so that the stores are coalesced
    __syncthreads();
    uint32_t q = local;
    bool sat = false;
    for (uint32_t x = 0; x < m; ++x)
        if ((heads >> x) & 1u) { s_code[q] = k[x].code; s_cnt[q] = c[x]; sat = sat || c[x] == 0xFFFFFFFFu; ++q; }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < (uint32_t)tile_heads; x += kScanBlock) {
        out_codes[tile_excl + x] = s_code[x];
        out_counts[tile_excl + x] = s_cnt[x];
    }
    if (sat) atomicOr(saturated, 1);
}

inline uint32_t merge_tiles(uint64_t n_a, uint64_t n_b) { return (uint32_t)((n_a + n_b + kMergeTile - 1) / kMergeTile); }

// out_codes / out_counts: room for n_a + n_b; split: 2 * (tiles + 1) u64; state: tiles + 2 u64; *saturated |= 1 when a sum does not fit u32.
// The number of merged k-mers is state[tiles + 1] afterwards.  Asynchronous on `st`.
inline hipError_t merge_launch(const uint32_t *ka, const uint64_t *a, const uint32_t *ca, uint64_t n_a, const uint32_t *kb, const uint64_t *b, const uint32_t *cb,
                               uint64_t n_b, uint64_t *out_codes, uint32_t *out_counts, uint64_t *split, uint64_t *state, int *saturated, hipStream_t st) {
    const uint32_t tiles = merge_tiles(n_a, n_b);
    hipError_t e = hipMemsetAsync(state, 0, ((size_t)tiles + 2) * 8, st);
    if (e != hipSuccess || tiles == 0) return e;
    hipLaunchKernelGGL(k_merge_partition, dim3((tiles + 1 + 255) / 256), dim3(256), 0, st, ka, a, n_a, kb, b, n_b, tiles, split, split + tiles + 1);
    hipLaunchKernelGGL(k_merge_tiles, dim3(tiles), dim3(kScanBlock), 0, st, ka, a, ca, kb, b, cb, split, split + tiles + 1, tiles, out_codes, out_counts, state, saturated);
    return hipGetLastError();
}

}  // namespace cid
