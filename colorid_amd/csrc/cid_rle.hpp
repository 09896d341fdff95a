// Run-length count of a sorted array in ONE pass (+ a pass over the tiles): distinct values and their multiplicities — the k-mer set's
// last step (kmer.rs:87-125: a map's keys and values).  Tiles of kRleTile elements in ticket order; the heads of runs are flagged (wave
// ballots give their ranks inside the wave), the tile's head count goes through the decoupled look-back of cid_scan.hpp and gives every
// head its rank; the heads' positions meet in LDS, where a head's multiplicity is the distance to the next one.  Only a tile's
// LAST head needs another tile (its run may go on): k_rle_tails closes those from the tiles' first-head positions.
//   state: rle_tiles(n) + 2 words, zeroed by rle_launch (scan_launch's layout: state[tiles + 1] = number of runs afterwards)
//   tile_info: 3 * rle_tiles(n) u32 — first head position | last head position | last head rank (kRleNone in [0]: a tile without heads)
#pragma once
#include "cid_scan.hpp"

namespace cid {

constexpr uint32_t kRleNone = 0xFFFFFFFFu;
constexpr uint32_t kRlePer = 16, kRleTile = kScanBlock * kRlePer;   // elements per lane and per tile
__host__ __device__ inline uint64_t rle_tiles(uint64_t n) { return (n + kRleTile - 1) / kRleTile; }

__global__ __launch_bounds__(kScanBlock) void k_rle(const uint64_t *in, uint32_t n, uint64_t *uniq, uint32_t *counts, uint64_t *state, uint32_t *tile_info) {
    __shared__ uint32_t s_pos[kRleTile + 1];
    const uint64_t tiles = rle_tiles(n);
    const uint64_t tile = scan_ticket(state, tiles);
    if (tile >= tiles) return;
    // A wave owns 64 * kRlePer consecutive elements, element j * 64 + lane in lane `lane`'s v[j]: every load and (where runs are short) every
    // store of a wave instruction covers 512 consecutive bytes.  (Eight consecutive elements per THREAD, as the generic scan has them, made
    // each instruction touch 64 lines: 1.47 ms per 120 M keys against 0.6 for the same bytes.)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w0 = (uint32_t)(tile * kRleTile) + wave * (64u * kRlePer);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint64_t v[kRlePer], masks[kRlePer];
    const uint64_t before = w0 > 0 && w0 < n ? in[w0 - 1] : 0ull;
#pragma unroll
    for (uint32_t j = 0; j < kRlePer; ++j) {
        const uint32_t i = w0 + j * 64u + lane;
        v[j] = i < n ? in[i] : 0ull;
    }
    uint32_t wave_heads = 0;
#pragma unroll
    for (uint32_t j = 0; j < kRlePer; ++j) {
        const uint32_t i = w0 + j * 64u + lane;
        const uint64_t up = __shfl_up(v[j], 1, 64);
        const uint64_t carried = j ? __shfl(v[j ? j - 1 : 0], 63, 64) : before;   // the element before this row of 64
        const uint64_t prev = lane ? up : carried;
        const bool head = i < n && (i == 0 || v[j] != prev);
        masks[j] = __ballot(head);
        wave_heads += (uint32_t)__popcll(masks[j]);
    }
    uint64_t tile_heads;
    const uint32_t wave_base = (uint32_t)__shfl((uint32_t)scan_block_exclusive(lane == 0 ? (uint64_t)wave_heads : 0ull, &tile_heads), 0, 64);
    const uint64_t tile_excl = scan_lookback_block(state, tile, tiles, tile_heads);
    uint32_t pre = wave_base;
#pragma unroll
    for (uint32_t j = 0; j < kRlePer; ++j) {
        if ((masks[j] >> lane) & 1ull) {
            const uint32_t q = pre + (uint32_t)__popcll(masks[j] & lt_mask);
            uniq[tile_excl + q] = v[j];
            s_pos[q] = w0 + j * 64u + lane;
        }
        pre += (uint32_t)__popcll(masks[j]);
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
    hipError_t e = hipMemsetAsync(state, 0, (rle_tiles(n) + 2) * 8, st);
    if (e != hipSuccess) return e;
    if (n) {
        const uint32_t tiles = (uint32_t)rle_tiles(n);
        hipLaunchKernelGGL(k_rle, dim3(tiles), dim3(kScanBlock), 0, st, in, n, uniq, counts, state, tile_info);
        hipLaunchKernelGGL(k_rle_tails, dim3((tiles + 255) / 256), dim3(256), 0, st, tile_info, tiles, n, counts);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipMemcpyAsync(d_n_runs, state + rle_tiles(n) + 1, 8, hipMemcpyDeviceToDevice, st);
}

}  // namespace cid
