// Device-side building blocks for the BIGSI query kernels (gfx950 / CDNA4, wave64).
//
//   * (cid_hash.hpp) XXH3-64-with-seed for inputs of 1..128 bytes read out of LDS — the hash colorid
//     calls at src/simple_bloom.rs:22-23, src/batch_search_pe.rs:48-49, src/read_id_mt_pe.rs:78-79 ... —
//     and the exact `% bloom_size` by a precomputed multiply-shift,
//   * wave-private staging of 64 k-mers' bytes in LDS with aligned 16-byte global loads.
//
// No MFMA anywhere: the path is 64-bit integer multiplies, byte shuffles and bitwise AND.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cid_hash.hpp"

namespace cid {

constexpr int kWave = 64;
constexpr uint32_t kMaxK = 128;

// ---------------------------------------------------------------- k-mer byte staging

// Bytes per wave for the k-mer image: 64 k-mers + 16 bytes of slack for the over-reads of lds_rd64.
__host__ __device__ inline uint32_t kmer_img_bytes(uint32_t k) { return ((kWave * k + 16u) + 15u) & ~15u; }

// Copy k-mers [first, first+64) (clipped to n_kmers) of a packed n_kmers*k byte array into this wave's
// LDS image.  `kmers` must be 16-byte aligned; 64*k is a multiple of 16, so every tile starts aligned.
__device__ __forceinline__ void stage_kmers(uint32_t *img, const uint8_t *kmers, uint64_t n_kmers, uint64_t first,
                                            uint32_t k, int lane) {
    const uint64_t total = n_kmers * k;
    const uint64_t g0 = first * k;
    const uint32_t nchunk = 4u * k;  // 64*k/16
    for (uint32_t c = lane; c < nchunk; c += kWave) {
        const uint64_t g = g0 + 16ull * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= total) {
            v = *reinterpret_cast<const uint4 *>(kmers + g);
        } else if (g < total) {  // the one ragged chunk at the very end of the array
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) {  // fully unrolled: w[] stays in registers
                const uint32_t byte = (g + b < total) ? (uint32_t)kmers[g + b] : 0u;
                w[b >> 2] |= byte << (8u * (b & 3u));
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(img + 4u * c) = v;
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace cid
