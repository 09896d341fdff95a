// Single-pass exclusive prefix sum (decoupled look-back) for the long-read path and the k-mer set: ONE kernel, every element read
// once and its prefix handed to an output functor once.  Tiles of kScanTile elements are drawn from an atomic ticket (a tile's
// predecessors have all started), a tile publishes {status, value} as ONE 8-byte word with a relaxed agent-scope store and looks
// back through its predecessors' words 64 at a time with relaxed agent-scope loads — self-contained granules, so no fence is needed
// between workgroups (MI355X_MICROARCH.md §inter-workgroup visibility: per-XCD L2s are not coherent, agent-scope accesses are).
//   In:  uint64_t operator()(uint64_t i)  — element i (any width below 2^62 in total)
//   Out: void operator()(uint64_t i, uint64_t exclusive_prefix, uint64_t value)
// state: ceil(n / kScanTile) + 2 u64 words, zeroed by the launcher (scan_launch); state[tiles] = ticket, state[tiles + 1] = total.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cid {

constexpr uint32_t kScanBlock = 256, kScanPer = 8, kScanTile = kScanBlock * kScanPer;
constexpr uint64_t kScanValueMask = (1ull << 62) - 1ull;

__host__ __device__ inline uint64_t scan_tiles(uint64_t n) { return (n + kScanTile - 1) / kScanTile; }
__host__ __device__ inline size_t scan_state_words(uint64_t n) { return (size_t)scan_tiles(n) + 2; }

__device__ __forceinline__ uint64_t scan_wave_sum(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The ticket of this workgroup's tile (tiles are handed out in launch order: a tile's predecessors have all started).
__device__ __forceinline__ uint64_t scan_ticket(uint64_t *state, uint64_t tiles) {
    __shared__ uint64_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<unsigned long long *>(&state[tiles]), 1ull);
    __syncthreads();
    return s_tile;
}

// The decoupled look-back of one tile, called by every thread of the workgroup with the tile's total: publishes the aggregate, sums the
// predecessors' words 64 at a time until one carries an inclusive prefix, publishes the tile's own inclusive prefix; returns the tile's
// exclusive prefix to every thread.  The last tile leaves the grand total in state[tiles + 1].
__device__ __forceinline__ uint64_t scan_lookback_block(uint64_t *state, uint64_t tile, uint64_t tiles, uint64_t tile_sum) {
    __shared__ uint64_t s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        uint64_t excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&state[0], (2ull << 62) | (tile_sum & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[tile], (1ull << 62) | (tile_sum & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t base = (int64_t)tile - 1;
            while (true) {
                const int64_t idx = base - lane;
                uint64_t st = (2ull << 62);   // before tile 0: an inclusive prefix of zero
                if (idx >= 0) st = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t)(st >> 62);
                if (__any(flag == 0)) { __builtin_amdgcn_s_sleep(1); continue; }   // a predecessor has not published yet
                const uint64_t done = __ballot(flag == 2);
                const uint64_t val = st & kScanValueMask;
                if (done) {
                    const int first = __builtin_ctzll(done);   // the nearest tile with an inclusive prefix
                    excl += scan_wave_sum(lane <= first ? val : 0ull);
                    break;
                }
                excl += scan_wave_sum(val);
                base -= 64;
            }
            if (lane == 0) __hip_atomic_store(&state[tile], (2ull << 62) | ((excl + tile_sum) & kScanValueMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = excl;
            if (tile == tiles - 1) state[tiles + 1] = excl + tile_sum;
        }
    }
    __syncthreads();
    return s_excl;
}

// Inclusive scan of one value per thread over the workgroup: returns this thread's exclusive prefix inside the tile, *tile_sum = the tile's total.
__device__ __forceinline__ uint64_t scan_block_exclusive(uint64_t mine, uint64_t *tile_sum) {
    __shared__ uint64_t s_wave[kScanBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint64_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < (int)(kScanBlock / 64); ++w) {
        if (w < wave) wave_base += s_wave[w];
        total += s_wave[w];
    }
    *tile_sum = total;
    return wave_base + (inc - mine);
}

template <typename In, typename Out>
__global__ __launch_bounds__(kScanBlock) void k_scan_lookback(In in, Out out, uint64_t n, uint64_t *state) {
    const uint64_t tiles = scan_tiles(n);
    const uint64_t tile = scan_ticket(state, tiles);
    if (tile >= tiles) return;
    // blocked arrangement: thread t owns kScanPer consecutive elements
    const uint64_t i0 = tile * kScanTile + (uint64_t)threadIdx.x * kScanPer;
    uint64_t v[kScanPer], mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; ++j) {
        v[j] = i0 + j < n ? in(i0 + j) : 0ull;
        mine += v[j];
    }
    uint64_t tile_sum;
    const uint64_t local = scan_block_exclusive(mine, &tile_sum);
    uint64_t run = scan_lookback_block(state, tile, tiles, tile_sum) + local;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; ++j) {
        if (i0 + j < n) out(i0 + j, run, v[j]);
        run += v[j];
    }
}

// state: scan_state_words(n) words of device scratch; after the kernel state[scan_tiles(n) + 1] holds the grand total
template <typename In, typename Out>
hipError_t scan_launch(In in, Out out, uint64_t n, uint64_t *state, hipStream_t st) {
    hipError_t e = hipMemsetAsync(state, 0, scan_state_words(n) * 8, st);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL((k_scan_lookback<In, Out>), dim3((unsigned)scan_tiles(n)), dim3(kScanBlock), 0, st, in, out, n, state);
    return hipGetLastError();
}

// plain arrays (in place allowed: a thread reads its elements before it writes them)
struct ScanInU32 {
    const uint32_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct ScanInU64 {
    const uint64_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct ScanOutU32 {
    uint32_t *p;
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t) const { p[i] = (uint32_t)excl; }
};
struct ScanOutU64 {
    uint64_t *p;
    uint64_t init;   // added to every prefix
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t) const { p[i] = init + excl; }
};

}  // namespace cid
