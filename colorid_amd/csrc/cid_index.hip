// Index maintenance kernels for MI355X (gfx950): .bxi rows <-> dense matrix, Bloom insert (simple_bloom.rs:19-26).
#include "cid_gather.hpp"

namespace cid {

// ------------------------------------------------------------------------------------------------
// index maintenance

// .bxi rows -> dense matrix (src/bigsi.rs:59-63 feeds this): one thread per (row, u32 word)
__global__ void k_put_rows(uint32_t *mat32, uint32_t rs, const uint64_t *row_ids, const uint32_t *words, uint32_t w32,
                           uint64_t n_rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * w32) return;
    const uint64_t r = i / w32, w = i % w32;
    mat32[row_ids[r] * (2ull * rs) + w] = words[i];
}

// The same from the file's own bytes: record i = { u64 row ; u64 n_words ; n_words x u32 ; u64 n_bits } (bincode of
// (usize, BitVec)), rec_bytes = 24 + 4*w32_rec, every field 4-byte aligned.  The index takes the record's
// words [w_off, w_off + w32_take) — all of them, or its colour stripe of a wider file (cid_group_stripes_put_records).  One
// thread per (record, taken word); word 0's thread also checks the record against the FILE's shape (w32_rec words, n_colors
// bits).  err[0] |= 1 bad word count, 2 bad bit count, 4 row >= bloom_size, 8 bits past n_colors.
__global__ void k_put_records(uint32_t *mat32, uint32_t rs, const uint32_t *rec32, uint32_t w32_rec, uint32_t w_off, uint32_t w32_take,
                              uint64_t n_records, uint64_t bloom_size, uint32_t n_colors, uint32_t tail_mask, uint32_t *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_records * w32_take) return;
    const uint64_t r = i / w32_take;
    const uint32_t w = (uint32_t)(i % w32_take);
    const uint32_t *rec = rec32 + r * (6ull + w32_rec);
    const uint64_t row = (uint64_t)rec[0] | ((uint64_t)rec[1] << 32);
    if (w == 0) {
        const uint64_t nw = (uint64_t)rec[2] | ((uint64_t)rec[3] << 32);
        const uint64_t nbits = (uint64_t)rec[4 + w32_rec] | ((uint64_t)rec[5 + w32_rec] << 32);
        uint32_t e = (nw != w32_rec ? 1u : 0u) | (nbits != n_colors ? 2u : 0u) | (row >= bloom_size ? 4u : 0u) |
                     ((rec[4 + w32_rec - 1] & ~tail_mask) ? 8u : 0u);
        if (e) atomicOr(err, e);
    }
    if (row < bloom_size) mat32[row * (2ull * rs) + w] = rec[4 + w_off + w];
}

__global__ void k_get_rows(const uint32_t *mat32, uint32_t rs, const uint64_t *row_ids, uint32_t *words, uint32_t w32,
                           uint64_t n_rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * w32) return;
    const uint64_t r = i / w32, w = i % w32;
    words[i] = mat32[row_ids[r] * (2ull * rs) + w];
}

// Bloom insert (src/simple_bloom.rs:19-26) straight into the transposed matrix (src/build.rs:116-128)
__global__ __launch_bounds__(kBlock) void k_insert_kmers(InsertParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    uint32_t *img = reinterpret_cast<uint32_t *>(smem + (size_t)wave * kmer_img_bytes(p.k));
    const uint64_t n_tiles = (p.n_kmers + kWave - 1) / kWave;
    const uint64_t tile0 = (uint64_t)blockIdx.x * p.tiles_per_block;
    const uint64_t tile1 = tile0 + p.tiles_per_block < n_tiles ? tile0 + p.tiles_per_block : n_tiles;
    unsigned int *mat32 = reinterpret_cast<unsigned int *>(p.mat);
    for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) {
        const uint64_t first = tile * kWave;
        auto set_bit = [&](uint32_t c, uint64_t h) {
            const uint64_t row = mod_m(h, p.mod);
            atomicOr(&mat32[row * (2ull * p.rs) + (c >> 5)], 1u << (c & 31u));
        };
        if (p.codes) {
            if (first + lane < p.n_kmers) {
                const uint32_t c = p.colour_of_kmer ? p.colour_of_kmer[first + lane] : p.colour;
                uint64_t code = p.codes[first + lane];
                uint32_t klen = p.k;
                if (p.m_size) { code = minimizer_code(code, p.k, p.m_size); klen = p.m_size; }
                const uint64_t lsb = rev_fields(code, klen);
                if (c < p.n_colors) xxh3_seeds_from(CodeReader{lsb}, klen, p.n_hash, HashSel::of(p.mod), [&](uint32_t, uint64_t h) { set_bit(c, h); });
            }
            continue;
        }
        wave_lds_fence();
        stage_kmers(img, p.kmers, p.n_kmers, first, p.k, lane);
        wave_lds_fence();
        if (p.m_size) {  // ASCII k-mers into a minimizer index: byte-wise find_minimizer, then hash its m_size bytes
            uint32_t *mimg = reinterpret_cast<uint32_t *>(smem + (size_t)(kBlock / kWave) * kmer_img_bytes(p.k) + (size_t)wave * kmer_img_bytes(p.m_size));
            uint8_t *mimg8 = reinterpret_cast<uint8_t *>(mimg);
            const bool have = first + lane < p.n_kmers;
            if (have) {
                const uint8_t *seq = reinterpret_cast<const uint8_t *>(img) + (uint32_t)lane * p.k;
                const uint32_t cand = find_minimizer_bytes(seq, p.k, p.m_size);
                for (uint32_t t = 0; t < p.m_size; ++t) mimg8[(uint32_t)lane * p.m_size + t] = mini_byte(seq, cand, p.m_size, t);
            }
            wave_lds_fence();
            if (have) {
                const uint32_t c = p.colour_of_kmer ? p.colour_of_kmer[first + lane] : p.colour;
                if (c < p.n_colors)
                    xxh3_seeds(mimg, (uint32_t)lane * p.m_size, p.m_size, p.n_hash, HashSel::of(p.mod), [&](uint32_t, uint64_t h) { set_bit(c, h); });
            }
            continue;
        }
        if (first + lane < p.n_kmers) {
            const uint32_t c = p.colour_of_kmer ? p.colour_of_kmer[first + lane] : p.colour;
            if (c < p.n_colors) xxh3_seeds(img, (uint32_t)lane * p.k, p.k, p.n_hash, HashSel::of(p.mod), [&](uint32_t, uint64_t h) { set_bit(c, h); });
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers

hipError_t launch_put_rows(uint64_t *mat, uint32_t rs, const uint64_t *d_row_ids, const uint32_t *d_words, uint32_t w32,
                           uint64_t n_rows, hipStream_t stream) {
    const uint64_t n = n_rows * w32;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_put_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(mat), rs, d_row_ids, d_words, w32, n_rows);
    return hipGetLastError();
}

hipError_t launch_put_records(uint64_t *mat, uint32_t rs, const uint32_t *d_records, uint32_t w32_rec, uint32_t w_off, uint32_t w32_take,
                              uint64_t n_records, uint64_t bloom_size, uint32_t n_colors, uint32_t *d_err, hipStream_t stream) {
    const uint64_t n = n_records * w32_take;
    if (n == 0) return hipSuccess;
    const uint32_t tail_bits = n_colors % 32;
    hipLaunchKernelGGL(k_put_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(mat), rs, d_records,
                       w32_rec, w_off, w32_take, n_records, bloom_size, n_colors, tail_bits ? ((1u << tail_bits) - 1u) : 0xFFFFFFFFu, d_err);
    return hipGetLastError();
}

hipError_t launch_get_rows(const uint64_t *mat, uint32_t rs, const uint64_t *d_row_ids, uint32_t *d_words, uint32_t w32,
                           uint64_t n_rows, hipStream_t stream) {
    const uint64_t n = n_rows * w32;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_get_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(mat), rs, d_row_ids, d_words, w32, n_rows);
    return hipGetLastError();
}

hipError_t launch_insert_kmers(const InsertParams &p, hipStream_t stream) {
    const size_t shmem = (size_t)(kBlock / kWave) * (kmer_img_bytes(p.k) + (p.m_size ? kmer_img_bytes(p.m_size) : 0));
    const int grid = grid_for(p.n_kmers, p.tiles_per_block);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_kmers, dim3(grid), dim3(kBlock), shmem, stream, p);
    return hipGetLastError();
}

}  // namespace cid
