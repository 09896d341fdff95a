// Run-length count of a sorted array in ONE pass (+ a pass over the tiles): distinct values and their multiplicities — the k-mer set's
// last step (kmer.rs:87-125: a map's keys and values).  Tiles of kScanTile elements in ticket order; a thread owns kScanPer consecutive
// elements, flags the heads of runs among them, the tile's head count goes through the decoupled look-back of cid_scan.hpp and gives
// every head its rank; the heads' positions meet in LDS, where a head's multiplicity is the distance to the next one.  Only a tile's
// LAST head needs another tile (its run may go on): k_rle_tails closes those from the tiles' first-head positions.
//   state: scan_state_words(n) words, zeroed (scan_launch's layout: state[tiles + 1] = number of runs afterwards)
//   tile_info: 3 * scan_tiles(n) u32 — first head position | last head position | last head rank (kRleNone in [0]: a tile without heads)
#pragma once
#include "cid_scan.hpp"

namespace cid {

constexpr uint32_t kRleNone = 0xFFFFFFFFu;

__global__ __launch_bounds__(kScanBlock) void k_rle(const uint64_t *in, uint32_t n, uint64_t *uniq, uint32_t *counts, uint64_t *state, uint32_t *tile_info) {
    __shared__ uint32_t s_pos[kScanTile + 1];
    const uint64_t tiles = scan_tiles(n);
    const uint64_t tile = scan_ticket(state, tiles);
    if (tile >= tiles) return;
    const uint32_t i0 = (uint32_t)(tile * kScanTile) + threadIdx.x * kScanPer;
    uint64_t v[kScanPer];
    uint64_t prev = 0;
    if (i0 > 0 && i0 < n) prev = in[i0 - 1];
    uint32_t heads = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; ++j) {
        v[j] = i0 + j < n ? in[i0 + j] : 0ull;
        const bool head = i0 + j < n && (i0 + j == 0 || v[j] != (j ? v[j - 1] : prev));
        heads |= head ? 1u << j : 0u;
    }
    uint64_t tile_heads;
    const uint32_t local = (uint32_t)scan_block_exclusive((uint64_t)__popc(heads), &tile_heads);
    const uint64_t tile_excl = scan_lookback_block(state, tile, tiles, tile_heads);
    uint32_t q = local;
#pragma unroll
    for (uint32_t j = 0; j < kScanPer; ++j) {
        if ((heads >> j) & 1u) {
            uniq[tile_excl + q] = v[j];
            s_pos[q] = i0 + j;
            ++q;
        }
    }
    __syncthreads();
    const uint32_t th = (uint32_t)tile_heads;
    for (uint32_t r = threadIdx.x; r + 1 < th; r += kScanBlock) counts[tile_excl + r] = s_pos[r + 1] - s_pos[r];
    if (threadIdx.x == 0) {
        tile_info[3 * tile] = th ? s_pos[0] : kRleNone;
        tile_info[3 * tile + 1] = th ? s_pos[th - 1] : 0u;
        tile_info[3 * tile + 2] = th ? (uint32_t)(tile_excl + th - 1) : 0u;
    }
}
// the multiplicity of every tile's last head: up to the first head of a later tile, or to the end
__global__ void k_rle_tails(const uint32_t *tile_info, uint32_t tiles, uint32_t n, uint32_t *counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles || tile_info[3 * t] == kRleNone) return;
    uint32_t end = n;
    for (uint32_t u = t + 1; u < tiles; ++u)
        if (tile_info[3 * u] != kRleNone) { end = tile_info[3 * u]; break; }
    counts[tile_info[3 * t + 2]] = end - tile_info[3 * t + 1];
}

// uniq / counts: room for n entries; *d_n_runs (device u64) receives the number of runs.  Asynchronous on `st`.
inline hipError_t rle_launch(const uint64_t *in, uint32_t n, uint64_t *uniq, uint32_t *counts, uint64_t *state, uint32_t *tile_info, uint64_t *d_n_runs,
                             hipStream_t st) {
    hipError_t e = hipMemsetAsync(state, 0, scan_state_words(n) * 8, st);
    if (e != hipSuccess) return e;
    if (n) {
        const uint32_t tiles = (uint32_t)scan_tiles(n);
        hipLaunchKernelGGL(k_rle, dim3(tiles), dim3(kScanBlock), 0, st, in, n, uniq, counts, state, tile_info);
        hipLaunchKernelGGL(k_rle_tails, dim3((tiles + 255) / 256), dim3(256), 0, st, tile_info, tiles, n, counts);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipMemcpyAsync(d_n_runs, state + scan_tiles(n) + 1, 8, hipMemcpyDeviceToDevice, st);
}

}  // namespace cid
