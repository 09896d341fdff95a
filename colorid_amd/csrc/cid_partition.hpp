// MSD radix partition of u64 keys, one 8-bit digit per pass, for the k-mer set's sort (cid_kmerset.hip: windows -> codes -> sorted,
// run-length counted).  An LSD radix sort moves every key once per digit — 8 passes for the 63 significant bits of a 31-mer code.
// Two MSD passes cut the array into 65 536 runs of ~n / 65 536 keys that share their top 16 bits; each run then fits one workgroup's
// LDS and is finished there (a segmented sort), so a key crosses HBM three times instead of eight.
//
// One pass = the classic three steps, no decoupled look-back (no wave ever waits on another workgroup):
//   k_part_hist    per tile of kPartTile keys: its 256-bin digit histogram -> table[(segment, digit, tile)]
//   exclusive scan over the table (laid out segment-major, digit-major inside a segment, tile-minor): since the segments tile the
//                  array in order, the scan IS every (segment, digit, tile)'s first output position
//   k_part_scatter per tile: keys staged in LDS grouped by digit, written out as one run per digit (~kPartTile / 256 keys each)
// Segments: seg_off[S + 1] (device).  A tile never straddles a segment; tile -> segment by binary search over tile_base[S + 1].
// A pass may use fewer than 8 bits (`bits`, bins = 1 << bits): the last level takes what is left of the planned prefix.
// The partition is not stable inside a digit (LDS atomics hand out the slots) — irrelevant here: keys are sorted to the end afterwards
// and equal keys are indistinguishable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cid {

constexpr uint32_t kPartBits = 8, kPartBins = 1u << kPartBits;
constexpr uint32_t kPartTile = 4096;   // keys per tile: 32 KiB of LDS staging, runs of ~16 keys = 128 bytes per digit (round 6, pairs: 2048 -> 4.43 ms per set, 8192 -> 4.27, 4096 -> 3.96)
constexpr uint32_t kPartBlock = 256;

// tile_base[s] = tiles of the segments before s (segment s has ceil(size / kPartTile) tiles); tile_base[S] = all tiles.  One block.
__global__ __launch_bounds__(kPartBlock) void k_part_tiles(const uint32_t *seg_off, uint32_t S, uint32_t *tile_base) {
    __shared__ uint32_t s_part[kPartBlock];
    const uint32_t per = (S + kPartBlock - 1) / kPartBlock;
    const uint32_t s0 = threadIdx.x * per, s1 = s0 + per < S ? s0 + per : S;
    uint32_t mine = 0;
    for (uint32_t s = s0; s < s1; ++s) mine += (seg_off[s + 1] - seg_off[s] + kPartTile - 1) / kPartTile;
    s_part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < kPartBlock; ++i) { const uint32_t v = s_part[i]; s_part[i] = run; run += v; }
        tile_base[S] = run;
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t s = s0; s < s1; ++s) { tile_base[s] = run; run += (seg_off[s + 1] - seg_off[s] + kPartTile - 1) / kPartTile; }
}

__device__ __forceinline__ uint32_t part_segment_of(const uint32_t *tile_base, uint32_t S, uint32_t tile) {
    uint32_t lo = 0, hi = S;   // the last s with tile_base[s] <= tile (empty segments share a base with their successor: skip them)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_base[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

struct PartTile {
    uint32_t seg, first, count, table_at, table_stride;   // keys [first, first + count); its bins at table[table_at + d * table_stride]
};
__device__ __forceinline__ PartTile part_tile(const uint32_t *seg_off, const uint32_t *tile_base, uint32_t S, uint32_t tile, uint32_t bins) {
    PartTile t;
    t.seg = part_segment_of(tile_base, S, tile);
    const uint32_t b0 = tile_base[t.seg], tiles_s = tile_base[t.seg + 1] - b0, tin = tile - b0;
    const uint32_t s_lo = seg_off[t.seg], s_hi = seg_off[t.seg + 1];
    t.first = s_lo + tin * kPartTile;
    t.count = s_hi - t.first < kPartTile ? s_hi - t.first : kPartTile;
    t.table_at = b0 * bins + tin;
    t.table_stride = tiles_s;
    return t;
}

// Every tile's description, once per level: found tile by tile inside the histogram and scatter kernels it was a chain of dependent loads
// at the start of every tile — the binary search over tile_base (eight steps at the second level's 256 segments), then the segment's
// bounds — which an in-order wave cannot overlap with anything.
__global__ void k_part_tile_desc(const uint32_t *seg_off, const uint32_t *tile_base, uint32_t S, uint32_t bins, PartTile *desc) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile < tile_base[S]) desc[tile] = part_tile(seg_off, tile_base, S, tile, bins);
}

// Which tiles a workgroup of a scatter kernel takes.  The runs a tile writes continue, byte for byte, the runs of the tile before it
// (same digit), so neighbouring tiles fill the same 128-byte lines.  Workgroups are dealt to the eight XCDs in turn (blockIdx & 7) and
// every XCD has an L2 of its own: with tile = blockIdx (+ k * gridDim) those two halves of a line are always written through two
// different L2s.  Here XCD x walks the x-th eighth of the tiles, its workgroups side by side on neighbouring tiles, so a line's halves
// meet in one L2 and leave as one full-line write.  (Grids that are not a multiple of 8 walk tile = blockIdx + k * gridDim.)
struct XcdWalk {
    uint32_t chunk;   // tiles per XCD (n_tiles when the grid is not a multiple of 8: then every workgroup strides over all tiles)
    bool split;
    __device__ explicit XcdWalk(uint32_t n_tiles) : chunk((n_tiles + 7u) / 8u), split(gridDim.x >= 8u && (gridDim.x & 7u) == 0u) {
        if (!split) chunk = n_tiles;   // (a grid of fewer than 8 workgroups would never advance; one that is not a multiple of 8 would visit tiles twice)
    }
    __device__ uint32_t first() const { return split ? blockIdx.x >> 3 : blockIdx.x; }
    __device__ uint32_t stride() const { return split ? gridDim.x >> 3 : gridDim.x; }
    __device__ uint32_t tile(uint32_t it) const { return split ? (blockIdx.x & 7u) * chunk + it : it; }
};

// `top` < 64: keys with a bit at or above `top` (the k-mer set's "no k-mer here" sentinel) are left out of the partition — counted
// into *n_dropped by the first level, skipped by the scatter — so that every digit below `top` orders real keys only.
__global__ __launch_bounds__(kPartBlock) void k_part_hist(const PartTile *desc, const uint64_t *keys, const uint32_t *seg_off, const uint32_t *tile_base, uint32_t S,
                                                           uint32_t shift, uint32_t bits, uint32_t top, uint32_t *table, uint32_t *n_dropped) {
    __shared__ uint32_t s_cnt[kPartBins], s_drop;
    const uint32_t n_tiles = tile_base[S], bins = 1u << bits;
    const XcdWalk walk(n_tiles);   // (the table's entries of neighbouring tiles share lines, like the scatter's runs)
    PartTile t_next = desc[walk.tile(walk.first()) < n_tiles ? walk.tile(walk.first()) : 0u];   // (the next tile's description is asked for a tile ahead)
    for (uint32_t it = walk.first(); it < walk.chunk; it += walk.stride()) {
        const uint32_t tile = walk.tile(it);
        if (tile >= n_tiles) break;
        const PartTile t = t_next;
        {
            const uint32_t nt = walk.tile(it + walk.stride());
            t_next = desc[it + walk.stride() < walk.chunk && nt < n_tiles ? nt : tile];
        }
        constexpr uint32_t PER = kPartTile / kPartBlock;
        uint64_t k[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {   // the tile's loads in flight together
            const uint32_t i = j * kPartBlock + threadIdx.x;
            k[j] = i < t.count ? keys[t.first + i] : 0ull;
        }
        s_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_drop = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (j * kPartBlock + threadIdx.x < t.count) {
                if (top < 64 && (k[j] >> top)) atomicAdd(&s_drop, 1u);
                else atomicAdd(&s_cnt[(uint32_t)(k[j] >> shift) & (bins - 1)], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < bins) table[t.table_at + threadIdx.x * t.table_stride] = s_cnt[threadIdx.x];
        if (threadIdx.x == 0 && s_drop) atomicAdd(n_dropped, s_drop);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kPartBlock) void k_part_scatter(const PartTile *desc, const uint64_t *keys, uint64_t *out, const uint32_t *seg_off, const uint32_t *tile_base,
                                                              uint32_t S, uint32_t shift, uint32_t bits, uint32_t top, const uint32_t *table) {
    __shared__ uint64_t s_stage[kPartTile];
    __shared__ uint32_t s_cnt[kPartBins], s_pre[kPartBins], s_cur[kPartBins], s_goff[kPartBins], s_wave[kPartBlock / 64];
    constexpr uint32_t PER = kPartTile / kPartBlock;
    const uint32_t n_tiles = tile_base[S], bins = 1u << bits;
    const XcdWalk walk(n_tiles);
    PartTile t_next = desc[walk.tile(walk.first()) < n_tiles ? walk.tile(walk.first()) : 0u];   // (the next tile's description is asked for a tile ahead)
    for (uint32_t it = walk.first(); it < walk.chunk; it += walk.stride()) {
        const uint32_t tile = walk.tile(it);
        if (tile >= n_tiles) break;
        const PartTile t = t_next;
        {
            const uint32_t nt = walk.tile(it + walk.stride());
            t_next = desc[it + walk.stride() < walk.chunk && nt < n_tiles ? nt : tile];
        }
        s_cnt[threadIdx.x] = 0;
        s_goff[threadIdx.x] = threadIdx.x < bins ? table[t.table_at + threadIdx.x * t.table_stride] : 0u;
        __syncthreads();
        uint64_t k[PER];
        bool keep[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {   // all of the tile's loads in flight before the first LDS atomic
            const uint32_t i = j * kPartBlock + threadIdx.x;
            k[j] = i < t.count ? keys[t.first + i] : ~0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            keep[j] = j * kPartBlock + threadIdx.x < t.count && !(top < 64 && (k[j] >> top));
            if (keep[j]) atomicAdd(&s_cnt[(uint32_t)(k[j] >> shift) & (bins - 1)], 1u);
        }
        __syncthreads();
        {   // exclusive prefix of the 256 bin counts: one bin per thread, a wave scan + the waves' totals
            const uint32_t v = s_cnt[threadIdx.x];
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)(threadIdx.x & 63) >= d) incl += u;
            }
            if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s_wave[w];
            s_pre[threadIdx.x] = base + incl - v;
            s_cur[threadIdx.x] = base + incl - v;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (keep[j]) s_stage[atomicAdd(&s_cur[(uint32_t)(k[j] >> shift) & (bins - 1)], 1u)] = k[j];
        }
        __syncthreads();
        const uint32_t kept = s_pre[kPartBins - 1] + s_cnt[kPartBins - 1];
        for (uint32_t i = threadIdx.x; i < kept; i += kPartBlock) {
            const uint64_t key = s_stage[i];
            const uint32_t d = (uint32_t)(key >> shift) & (bins - 1);
            out[s_goff[d] + (i - s_pre[d])] = key;
        }
        __syncthreads();
    }
}

// the next level's segments: (s, d) starts where its first tile's bin starts; new_off[S << bits] = n
__global__ void k_part_offsets(const uint32_t *table, const uint32_t *tile_base, uint32_t S, uint32_t bits, uint32_t n_in, const uint32_t *n_dropped,
                               uint32_t *new_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bins = 1u << bits;
    const uint32_t n = n_in - *n_dropped;   // the keys that went through the partition
    if (i > S * bins) return;
    if (i == S * bins) { new_off[i] = n; return; }
    const uint32_t s = i >> bits, d = i & (bins - 1);
    const uint32_t b0 = tile_base[s], tiles_s = tile_base[s + 1] - b0;
    // an empty segment has no tiles: all its children start where the next non-empty entry starts — the scan makes the table's
    // entry at the successor's position exactly that (or n past the end)
    const uint32_t at = b0 * bins + d * tiles_s;
    new_off[i] = at < tile_base[S] * bins ? table[at] : n;
}

// ---------------------------------------------------------------------------------------------------------------- runs finished in LDS
// One workgroup sorts one run (the keys share every bit above `bits`) with a STABLE least-significant-digit radix sort that never
// leaves the CU (the robust path: its cost does not depend on how the keys are spread): 8-bit digits, keys in registers between
// passes, one LDS key array.  Wave w owns the positions [w * chunk, (w+1) * chunk)
// of the run and walks them 64 at a time; a key's rank among equal digits = those in earlier waves + those in this wave's earlier
// rounds + those in lower lanes of its round (the lanes with the same digit are found with eight ballots).  No atomics decide an
// order, so equal digits keep their order and the passes compose.  MAXR = rounds per wave (capacity 256 * MAXR keys).
template <int MAXR>
__global__ __launch_bounds__(kPartBlock) void k_run_sort(const uint64_t *in, uint64_t *out, const uint32_t *run_off, uint32_t n_runs, uint32_t bits,
                                                          uint32_t min_size, uint32_t max_size, const uint32_t *list, const uint32_t *list_n) {
    __shared__ uint64_t s_key[kPartBlock * MAXR];
    __shared__ uint32_t s_cnt[4][kPartBins];
    __shared__ uint32_t s_wave[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t passes = (bits + 7) / 8;
    const uint32_t n_todo = list ? *list_n : n_runs;   // list: only the runs named there (the bucket kernel's hard ones)
    for (uint32_t at = blockIdx.x; at < n_todo; at += gridDim.x) {
        const uint32_t run = list ? list[at] : at;
        const uint32_t start = run_off[run], N = run_off[run + 1] - start;
        if (N < min_size || N > max_size) continue;   // (uniform over the block: another launch, or the big-run path, takes it)
        const uint32_t rounds = (N + kPartBlock - 1) / kPartBlock;   // per wave; positions [0, N) of the run are the valid ones in every pass
        const uint32_t chunk = rounds * 64;
        uint64_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
            key[r] = ((uint32_t)r < rounds && p < N) ? in[start + p] : 0ull;
        }
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t shift = 8 * pass;
            const uint32_t dmask = bits - shift >= 8 ? 0xFFu : ((1u << (bits - shift)) - 1u);
            s_cnt[0][threadIdx.x] = 0; s_cnt[1][threadIdx.x] = 0; s_cnt[2][threadIdx.x] = 0; s_cnt[3][threadIdx.x] = 0;
            __syncthreads();
            uint32_t woff[MAXR];   // rank among the wave's keys of the same digit
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                woff[r] = 0;
                if ((uint32_t)r < rounds) {   // wave-uniform
                    const bool valid = w * chunk + (uint32_t)r * 64 + lane < N;
                    const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
                    uint64_t m = __ballot(valid);   // ... narrowed to the valid lanes holding the same digit
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const uint64_t bal = __ballot((d >> b) & 1u);
                        m &= ((d >> b) & 1u) ? bal : ~bal;
                    }
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    uint32_t old = 0;
                    // the group's lowest lane books the group's slots (LDS atomics of one wave run in program order: the next round sees them)
                    if (valid && below == 0) old = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll((unsigned long long)m));
                    old = __shfl(old, valid ? __ffsll((unsigned long long)m) - 1 : (int)lane, 64);
                    woff[r] = old + below;
                }
            }
            __syncthreads();
            {   // bin d (one per thread): first slot of (digit d, wave w') = keys of smaller digits + digit d's keys in earlier waves
                const uint32_t c0 = s_cnt[0][threadIdx.x], c1 = s_cnt[1][threadIdx.x], c2 = s_cnt[2][threadIdx.x], c3 = s_cnt[3][threadIdx.x];
                const uint32_t v = c0 + c1 + c2 + c3;
                uint32_t incl = v;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t u = __shfl_up(incl, dd, 64);
                    if ((int)lane >= dd) incl += u;
                }
                if (lane == 63) s_wave[w] = incl;
                __syncthreads();
                uint32_t base = incl - v;
                for (uint32_t ww = 0; ww < w; ++ww) base += s_wave[ww];
                s_cnt[0][threadIdx.x] = base; s_cnt[1][threadIdx.x] = base + c0; s_cnt[2][threadIdx.x] = base + c0 + c1;
                s_cnt[3][threadIdx.x] = base + c0 + c1 + c2;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MAXR; ++r)
                if ((uint32_t)r < rounds && w * chunk + (uint32_t)r * 64 + lane < N)
                    s_key[s_cnt[w][(uint32_t)(key[r] >> shift) & dmask] + woff[r]] = key[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
                if ((uint32_t)r < rounds && p < N) key[r] = s_key[p];
            }
            // (the next pass's first barrier separates these reads from its scatter)
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
            if ((uint32_t)r < rounds && p < N) out[start + p] = key[r];
        }
    }
}

// The common case, several times cheaper: a run of evenly spread keys (k-mer codes are) is bucket-sorted in LDS.  Its next 11 bits
// pick one of 2048 buckets — about one DISTINCT key per bucket — by counting, a scan and a scatter through LDS atomics.  Inside a
// bucket every key finds its own place by comparing itself with the bucket's other members (rank = members that are smaller, or
// equal and placed before it): the bucket's keys work side by side, so the many copies of one k-mer that reads bring (coverage) cost
// m comparisons per key, not a serial insertion.  A run whose buckets would take more than crowded_at comparisons per key on average
// (deep coverage: many copies of every k-mer; repetitive sequence: many keys agreeing in 27+ bits) is left alone: its number is appended
// to hard_list for k_run_dedupe_sort (cid_rundedupe.hpp), which hands what is not copies on to the radix kernel above, whose cost does
// not depend on the keys.  Capacity 256 * MAXR keys: MAXR = 8 takes the runs of up to 2048
// keys at five workgroups per CU, MAXR = 16 those of 2049 .. 4096 (min_size) and names the larger ones.
constexpr uint32_t kBucketBits = 11, kBuckets = 1u << kBucketBits, kBucketWork = 32;
template <int MAXR>
__global__ __launch_bounds__(kPartBlock) void k_run_bucket_sort(const uint64_t *in, uint64_t *out, const uint32_t *run_off, uint32_t n_runs, uint32_t bits,
                                                                 uint32_t min_size, uint32_t *n_hard, uint32_t *hard_list, uint32_t crowded_at = kBucketWork) {
    __shared__ uint64_t s_key[kPartBlock * MAXR];
    __shared__ uint16_t s_pre[kBuckets + 1];   // (a run holds at most 4096 keys: 16 bits, and a workgroup more per CU)
    __shared__ uint32_t s_cur[kBuckets];
    __shared__ uint32_t s_wave[4], s_work[4];
    constexpr uint32_t BPT = kBuckets / kPartBlock;   // buckets per thread
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t bshift = bits > kBucketBits ? bits - kBucketBits : 0u;
    const uint32_t bmask = bits >= kBucketBits ? kBuckets - 1u : ((1u << bits) - 1u);
    // (the same look-ahead as k_run_bucket_sort_pair's made THIS kernel slower — 3.07-3.11 -> 3.25-3.29 ms per set in code order: with
    // sixteen keys per thread in its larger instantiation the next run's keys cost it its occupancy)
    for (uint32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const uint32_t start = run_off[run], N = run_off[run + 1] - start;
        if (N < min_size) continue;    // (uniform over the block) nothing there, or a smaller instantiation's run
        if (N > kPartBlock * MAXR) {   // too large for this kernel's LDS (with MAXR = 16, the largest instantiation: the radix kernel's)
            if (threadIdx.x == 0 && MAXR == 16) hard_list[atomicAdd(n_hard, 1u)] = run;
            continue;
        }
#pragma unroll
        for (uint32_t j = 0; j < BPT; ++j) s_cur[threadIdx.x * BPT + j] = 0;
        __syncthreads();
        uint64_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {   // the run's loads in flight together, before the first LDS atomic
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            key[r] = p < N ? in[start + p] : 0ull;
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) atomicAdd(&s_cur[(uint32_t)(key[r] >> bshift) & bmask], 1u);
        __syncthreads();
        {   // exclusive prefix over the 2048 bucket counts (8 per thread), and the comparisons the buckets will take: sum of count^2
            uint32_t c[BPT], sum = 0, sq = 0;
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { c[j] = s_cur[threadIdx.x * BPT + j]; sum += c[j]; sq += c[j] * c[j]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
            if (lane == 63) s_wave[w] = incl;
            if (lane == 0) s_work[w] = sq;
            __syncthreads();
            uint32_t base = incl - sum;
            for (uint32_t ww = 0; ww < w; ++ww) base += s_wave[ww];
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { s_pre[threadIdx.x * BPT + j] = (uint16_t)base; s_cur[threadIdx.x * BPT + j] = base; base += c[j]; }
            if (threadIdx.x == kPartBlock - 1) s_pre[kBuckets] = (uint16_t)base;
        }
        __syncthreads();
        if (s_work[0] + s_work[1] + s_work[2] + s_work[3] > crowded_at * N) {   // (uniform) crowded buckets: named for k_run_dedupe_sort / the radix kernel
            if (threadIdx.x == 0) hard_list[atomicAdd(n_hard, 1u)] = run;
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) s_key[atomicAdd(&s_cur[(uint32_t)(key[r] >> bshift) & bmask], 1u)] = key[r];
        __syncthreads();
        uint32_t dest[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {   // the key now at position p: its place among its bucket's members
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            dest[r] = p;
            if (p < N) {
                const uint64_t k = s_key[p];
                key[r] = k;
                const uint32_t bkt = (uint32_t)(k >> bshift) & bmask;
                const uint32_t lo = s_pre[bkt], hi = s_pre[bkt + 1];
                uint32_t rank = 0;
                for (uint32_t q = lo; q < hi; ++q) {
                    const uint64_t o = s_key[q];
                    rank += (o < k || (o == k && q < p)) ? 1u : 0u;
                }
                dest[r] = lo + rank;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) s_key[dest[r]] = key[r];
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < N; p += kPartBlock) out[start + p] = s_key[p];
        __syncthreads();
    }
}

// per run: 1 if it is larger than `cap` (it takes the big-run path); also the largest run's size
__global__ void k_run_sizes(const uint32_t *run_off, uint32_t n_runs, uint32_t cap, uint32_t *n_big, uint32_t *big_list, uint32_t big_cap, uint32_t *max_size) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sz = i < n_runs ? run_off[i + 1] - run_off[i] : 0u;
    uint32_t mx = sz;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_size, mx);
    if (sz > cap) {
        const uint32_t at = atomicAdd(n_big, 1u);
        if (at < big_cap) big_list[at] = i;
    }
}

// ---------------------------------------------------------------------------------------------- (u32 key, u64 value) pairs
// The same partition and run sorts for a k-mer set that is built FOR an index (cid_kmerset_set_target_index): every window carries,
// next to its 2-bit code, a 32-bit key derived from the row its first hash selects (cid_kmerset.hip: row0_key — monotone in the row
// number and spread evenly over 32 bits).  The partition passes consume the key's leading bits, the runs are finished in LDS on
// (rest of the key, code): the set comes out ordered by (first row, code).  Equal k-mers share the key, so they are still adjacent
// and the run-length count is unchanged, while the search's first-row fetches of neighbouring k-mers fall into the same 128-byte
// lines.  A key of all ones marks "no k-mer here" (row0_key never produces it) and is left out by the first level.
// (kNoKey: cid_kernels.hpp)

__global__ __launch_bounds__(kPartBlock) void k_part_hist_key(const PartTile *desc, const uint32_t *keys, const uint32_t *seg_off, const uint32_t *tile_base, uint32_t S,
                                                               uint32_t shift, uint32_t bits, uint32_t first_level, uint32_t *table, uint32_t *n_dropped) {
    __shared__ uint32_t s_cnt[kPartBins], s_drop;
    constexpr uint32_t PER = kPartTile / kPartBlock;
    const uint32_t n_tiles = tile_base[S], bins = 1u << bits;
    const XcdWalk walk(n_tiles);
    PartTile t_next = desc[walk.tile(walk.first()) < n_tiles ? walk.tile(walk.first()) : 0u];   // (the next tile's description is asked for a tile ahead)
    for (uint32_t it = walk.first(); it < walk.chunk; it += walk.stride()) {
        const uint32_t tile = walk.tile(it);
        if (tile >= n_tiles) break;
        const PartTile t = t_next;
        {
            const uint32_t nt = walk.tile(it + walk.stride());
            t_next = desc[it + walk.stride() < walk.chunk && nt < n_tiles ? nt : tile];
        }
        uint32_t k[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {   // the tile's loads in flight together
            const uint32_t i = j * kPartBlock + threadIdx.x;
            k[j] = i < t.count ? keys[t.first + i] : kNoKey;
        }
        s_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_drop = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = j * kPartBlock + threadIdx.x;
            if (i < t.count) {
                if (first_level && k[j] == kNoKey) atomicAdd(&s_drop, 1u);
                else atomicAdd(&s_cnt[(k[j] >> shift) & (bins - 1)], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < bins) table[t.table_at + threadIdx.x * t.table_stride] = s_cnt[threadIdx.x];
        if (threadIdx.x == 0 && s_drop) atomicAdd(n_dropped, s_drop);
        __syncthreads();
    }
}

// Keys and values are staged side by side in LDS (48 KiB: three workgroups per CU) and leave as one run per digit each.
__global__ __launch_bounds__(kPartBlock) void k_part_scatter_pair(const PartTile *desc, const uint32_t *keys, const uint64_t *vals, uint32_t *keys_out, uint64_t *vals_out,
                                                                   const uint32_t *seg_off, const uint32_t *tile_base, uint32_t S, uint32_t shift,
                                                                   uint32_t bits, uint32_t first_level, const uint32_t *table) {
    __shared__ uint64_t s_val[kPartTile];
    __shared__ uint32_t s_key[kPartTile];
    __shared__ uint32_t s_cnt[kPartBins], s_pre[kPartBins], s_cur[kPartBins], s_goff[kPartBins], s_wave[kPartBlock / 64];
    constexpr uint32_t PER = kPartTile / kPartBlock;
    const uint32_t n_tiles = tile_base[S], bins = 1u << bits;
    const XcdWalk walk(n_tiles);
    PartTile t_next = desc[walk.tile(walk.first()) < n_tiles ? walk.tile(walk.first()) : 0u];   // (the next tile's description is asked for a tile ahead)
    for (uint32_t it = walk.first(); it < walk.chunk; it += walk.stride()) {
        const uint32_t tile = walk.tile(it);
        if (tile >= n_tiles) break;
        const PartTile t = t_next;
        {
            const uint32_t nt = walk.tile(it + walk.stride());
            t_next = desc[it + walk.stride() < walk.chunk && nt < n_tiles ? nt : tile];
        }
        s_cnt[threadIdx.x] = 0;
        s_goff[threadIdx.x] = threadIdx.x < bins ? table[t.table_at + threadIdx.x * t.table_stride] : 0u;
        __syncthreads();
        uint32_t k[PER];
        uint64_t v[PER];
        bool keep[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {   // all of the tile's loads in flight before the first is used
            const uint32_t i = j * kPartBlock + threadIdx.x;
            k[j] = i < t.count ? keys[t.first + i] : kNoKey;
            v[j] = i < t.count ? vals[t.first + i] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = j * kPartBlock + threadIdx.x;
            keep[j] = i < t.count && !(first_level && k[j] == kNoKey);
            if (keep[j]) atomicAdd(&s_cnt[(k[j] >> shift) & (bins - 1)], 1u);
        }
        __syncthreads();
        {   // exclusive prefix of the 256 bin counts: one bin per thread, a wave scan + the waves' totals
            const uint32_t c = s_cnt[threadIdx.x];
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)(threadIdx.x & 63) >= d) incl += u;
            }
            if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += s_wave[w];
            s_pre[threadIdx.x] = base + incl - c;
            s_cur[threadIdx.x] = base + incl - c;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (keep[j]) {
                const uint32_t at = atomicAdd(&s_cur[(k[j] >> shift) & (bins - 1)], 1u);
                s_val[at] = v[j];
                s_key[at] = k[j];
            }
        }
        __syncthreads();
        const uint32_t kept = s_pre[kPartBins - 1] + s_cnt[kPartBins - 1];
        for (uint32_t i = threadIdx.x; i < kept; i += kPartBlock) {
            const uint32_t key = s_key[i];
            const uint32_t d = (key >> shift) & (bins - 1);
            const uint32_t to = s_goff[d] + (i - s_pre[d]);
            vals_out[to] = s_val[i];
            keys_out[to] = key;
        }
        __syncthreads();
    }
}

// A pair's place in its run: by the `kbits` low bits of the key, then by the value (`vbits` significant bits).
struct PairOrder {
    uint32_t kbits, vbits;
    __device__ __forceinline__ uint32_t kmask() const { return kbits >= 32 ? 0xFFFFFFFFu : ((1u << kbits) - 1u); }
    // the leading kBucketBits of (key's low kbits ++ value's vbits)
    __device__ __forceinline__ uint32_t bucket(uint32_t key, uint64_t val, uint32_t bucket_bits) const {
        const uint32_t kk = key & kmask();
        if (kbits >= bucket_bits) return kk >> (kbits - bucket_bits);
        const uint32_t take = bucket_bits - kbits;    // the rest comes from the top of the value
        const uint32_t top = vbits >= take ? (uint32_t)(val >> (vbits - take)) : (uint32_t)(val << (take - vbits));
        return (kk << take) | (top & ((1u << take) - 1u));
    }
};

// k_run_bucket_sort for pairs: the run's values come out in (key, value) order; the keys are not written (nothing after the sort
// needs them).  The bucket of a pair = the leading 11 bits of what the run still differs in.
template <int MAXR>
__global__ __launch_bounds__(kPartBlock) void k_run_bucket_sort_pair(const uint32_t *keys, const uint64_t *vals, uint64_t *out, const uint32_t *run_off,
                                                                      uint32_t n_runs, PairOrder ord, uint32_t min_size, uint32_t *n_hard, uint32_t *hard_list,
                                                                      uint32_t crowded_at = kBucketWork) {
    __shared__ uint64_t s_val[kPartBlock * MAXR];
    __shared__ uint32_t s_key[kPartBlock * MAXR];
    __shared__ uint16_t s_pre[kBuckets + 1];   // (a run holds at most 4096 keys: 16 bits, and a workgroup more per CU)
    __shared__ uint32_t s_cur[kBuckets];
    __shared__ uint32_t s_wave[4], s_work[4];
    constexpr uint32_t BPT = kBuckets / kPartBlock;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t total_bits = ord.kbits + ord.vbits;
    const uint32_t bbits = total_bits < kBucketBits ? total_bits : kBucketBits;
    const uint32_t kmask = ord.kmask();
    // The NEXT run's pairs are asked for before this run's are worked on, and the offsets of the run after that: a run is two trips to
    // memory (its offsets, then its pairs) and seven barriers of LDS work, and a wave issues in order — it cannot be ahead of a load it has
    // not asked for yet.  Every load unconditional, from an address that exists (pair 0 stands in for what a run does not have).
    // 1.04 -> 0.81 ms per 120 M pairs.  (The same in k_part_scatter_pair — 48 registers of the next tile's pairs — made it slower:
    // 0.91 -> 0.94 ms; its tiles are not waiting for their loads.)
    uint32_t start_n = 0, N_n = 0, start_nn = 0, N_nn = 0;
    uint64_t val_n[MAXR];
    uint32_t key_n[MAXR];
    auto offsets = [&](uint32_t run, uint32_t &st, uint32_t &n) {
        const uint32_t r = run < n_runs ? run : 0u;
        st = run_off[r];
        n = run < n_runs ? run_off[r + 1] - st : 0u;
    };
    auto pairs = [&](uint32_t st, uint32_t n) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            const uint32_t at = p < n ? st + p : 0u;
            key_n[r] = keys[at];
            val_n[r] = vals[at];
        }
    };
    offsets(blockIdx.x, start_n, N_n);
    offsets(blockIdx.x + gridDim.x, start_nn, N_nn);
    pairs(start_n, N_n);
    for (uint32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const uint32_t start = start_n, N = N_n;
        uint64_t val[MAXR];
        uint32_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) { key[r] = key_n[r] & kmask; val[r] = val_n[r]; }
        start_n = start_nn; N_n = N_nn;
        pairs(start_n, N_n);
        offsets(run + 2u * gridDim.x, start_nn, N_nn);
        if (N < min_size) continue;
        if (N > kPartBlock * MAXR) {
            if (threadIdx.x == 0 && MAXR == 16) hard_list[atomicAdd(n_hard, 1u)] = run;
            continue;
        }
#pragma unroll
        for (uint32_t j = 0; j < BPT; ++j) s_cur[threadIdx.x * BPT + j] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) atomicAdd(&s_cur[ord.bucket(key[r], val[r], bbits)], 1u);
        __syncthreads();
        {
            uint32_t c[BPT], sum = 0, sq = 0;
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { c[j] = s_cur[threadIdx.x * BPT + j]; sum += c[j]; sq += c[j] * c[j]; }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
            if (lane == 63) s_wave[w] = incl;
            if (lane == 0) s_work[w] = sq;
            __syncthreads();
            uint32_t base = incl - sum;
            for (uint32_t ww = 0; ww < w; ++ww) base += s_wave[ww];
#pragma unroll
            for (uint32_t j = 0; j < BPT; ++j) { s_pre[threadIdx.x * BPT + j] = (uint16_t)base; s_cur[threadIdx.x * BPT + j] = base; base += c[j]; }
            if (threadIdx.x == kPartBlock - 1) s_pre[kBuckets] = (uint16_t)base;
        }
        __syncthreads();
        if (s_work[0] + s_work[1] + s_work[2] + s_work[3] > crowded_at * N) {   // (uniform) crowded buckets: named for k_run_dedupe_sort / the radix kernel
            if (threadIdx.x == 0) hard_list[atomicAdd(n_hard, 1u)] = run;
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) {
                const uint32_t at = atomicAdd(&s_cur[ord.bucket(key[r], val[r], bbits)], 1u);
                s_val[at] = val[r];
                s_key[at] = key[r];
            }
        __syncthreads();
        uint32_t dest[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = (uint32_t)r * kPartBlock + threadIdx.x;
            dest[r] = p;
            if (p < N) {
                const uint64_t v = s_val[p];
                const uint32_t k = s_key[p];
                val[r] = v;
                const uint32_t bkt = ord.bucket(k, v, bbits);
                const uint32_t lo = s_pre[bkt], hi = s_pre[bkt + 1];
                uint32_t rank = 0;
                for (uint32_t q = lo; q < hi; ++q) {
                    const uint64_t ov = s_val[q];
                    const uint32_t ok = s_key[q];
                    rank += (ok < k || (ok == k && (ov < v || (ov == v && q < p)))) ? 1u : 0u;
                }
                dest[r] = lo + rank;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((uint32_t)r * kPartBlock + threadIdx.x < N) s_val[dest[r]] = val[r];
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < N; p += kPartBlock) out[start + p] = s_val[p];
        __syncthreads();
    }
}

// k_run_sort for pairs (the robust path of the runs the bucket kernel names): a stable LSD radix sort in LDS over the value's digits,
// then over the key's remaining bits.
template <int MAXR>
__global__ __launch_bounds__(kPartBlock) void k_run_sort_pair(const uint32_t *keys, const uint64_t *vals, uint64_t *out, const uint32_t *run_off, uint32_t n_runs,
                                                               PairOrder ord, uint32_t min_size, uint32_t max_size, const uint32_t *list, const uint32_t *list_n) {
    __shared__ uint64_t s_val[kPartBlock * MAXR];
    __shared__ uint32_t s_key[kPartBlock * MAXR];
    __shared__ uint32_t s_cnt[4][kPartBins];
    __shared__ uint32_t s_wave[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t vpasses = (ord.vbits + 7) / 8, kpasses = (ord.kbits + 7) / 8;
    const uint32_t kmask = ord.kmask();
    const uint32_t n_todo = list ? *list_n : n_runs;
    for (uint32_t at = blockIdx.x; at < n_todo; at += gridDim.x) {
        const uint32_t run = list ? list[at] : at;
        const uint32_t start = run_off[run], N = run_off[run + 1] - start;
        if (N < min_size || N > max_size) continue;
        const uint32_t rounds = (N + kPartBlock - 1) / kPartBlock;
        const uint32_t chunk = rounds * 64;
        uint64_t val[MAXR];
        uint32_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
            const bool in = (uint32_t)r < rounds && p < N;
            val[r] = in ? vals[start + p] : 0ull;
            key[r] = in ? (keys[start + p] & kmask) : 0u;
        }
        for (uint32_t pass = 0; pass < vpasses + kpasses; ++pass) {
            const bool on_val = pass < vpasses;
            const uint32_t shift = 8 * (on_val ? pass : pass - vpasses);
            const uint32_t left = (on_val ? ord.vbits : ord.kbits) - shift;
            const uint32_t dmask = left >= 8 ? 0xFFu : ((1u << left) - 1u);
            s_cnt[0][threadIdx.x] = 0; s_cnt[1][threadIdx.x] = 0; s_cnt[2][threadIdx.x] = 0; s_cnt[3][threadIdx.x] = 0;
            __syncthreads();
            uint32_t woff[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                woff[r] = 0;
                if ((uint32_t)r < rounds) {
                    const bool valid = w * chunk + (uint32_t)r * 64 + lane < N;
                    const uint32_t d = (on_val ? (uint32_t)(val[r] >> shift) : (key[r] >> shift)) & dmask;
                    uint64_t m = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const uint64_t bal = __ballot((d >> b) & 1u);
                        m &= ((d >> b) & 1u) ? bal : ~bal;
                    }
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    uint32_t old = 0;
                    if (valid && below == 0) old = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll((unsigned long long)m));
                    old = __shfl(old, valid ? __ffsll((unsigned long long)m) - 1 : (int)lane, 64);
                    woff[r] = old + below;
                }
            }
            __syncthreads();
            {
                const uint32_t c0 = s_cnt[0][threadIdx.x], c1 = s_cnt[1][threadIdx.x], c2 = s_cnt[2][threadIdx.x], c3 = s_cnt[3][threadIdx.x];
                const uint32_t v = c0 + c1 + c2 + c3;
                uint32_t incl = v;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t u = __shfl_up(incl, dd, 64);
                    if ((int)lane >= dd) incl += u;
                }
                if (lane == 63) s_wave[w] = incl;
                __syncthreads();
                uint32_t base = incl - v;
                for (uint32_t ww = 0; ww < w; ++ww) base += s_wave[ww];
                s_cnt[0][threadIdx.x] = base; s_cnt[1][threadIdx.x] = base + c0; s_cnt[2][threadIdx.x] = base + c0 + c1;
                s_cnt[3][threadIdx.x] = base + c0 + c1 + c2;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MAXR; ++r)
                if ((uint32_t)r < rounds && w * chunk + (uint32_t)r * 64 + lane < N) {
                    const uint32_t d = (on_val ? (uint32_t)(val[r] >> shift) : (key[r] >> shift)) & dmask;
                    const uint32_t to = s_cnt[w][d] + woff[r];
                    s_val[to] = val[r];
                    s_key[to] = key[r];
                }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
                if ((uint32_t)r < rounds && p < N) { val[r] = s_val[p]; key[r] = s_key[p]; }
            }
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const uint32_t p = w * chunk + (uint32_t)r * 64 + lane;
            if ((uint32_t)r < rounds && p < N) out[start + p] = val[r];
        }
    }
}

inline uint32_t part_max_tiles(uint32_t n, uint32_t S) { return n / kPartTile + S + 1; }

// The histogram kernels write every bin of every tile there is, and the scan runs over the table's planned size: only what lies behind the
// last tile's bins has to be zero.  (The whole table was cleared before every level: 30 MB per level for 120 M keys, 0.1 ms per set.)
__attribute__((unused)) static __global__ void k_part_zero_tail(uint32_t *table, const uint32_t *tile_base, uint32_t S, uint32_t bins, uint64_t table_n) {
    for (uint64_t i = (uint64_t)tile_base[S] * bins + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < table_n; i += (uint64_t)gridDim.x * blockDim.x) table[i] = 0;
}

}  // namespace cid
