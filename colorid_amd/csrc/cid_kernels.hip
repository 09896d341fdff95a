// BIGSI query kernels for MI355X (gfx950).  Hand-written HIP; wave64; HBM-bound bitwise work.
//
// Data layout in HBM: the index is a dense row-major bit matrix, row r = the colour bit-vector of Bloom
// position r, `rs` u64 words per row: rs = 1, or a power of two 2..128 (16 B .. 1 KiB per row, so that a row never
// straddles a 128-byte line it does not fill), or — beyond 8192 colours, "wide" rows — a multiple of 128 words.
// Absent rows of the reference's sparse map are all-zero rows here.
//
// Work decomposition (search kernels): one wave owns a tile of 64 k-mers at a time; a block owns a contiguous
// range of tiles (dynamic balance over the CUs, no cross-workgroup communication except the final atomics).
//   1. the tile's 64*k bytes are copied HBM -> LDS with aligned 16-byte loads (wave-private image) — or, when the
//      k-mers arrive as 2-bit codes, one u64 per lane is read and re-expanded to ASCII in registers;
//   2. lane l hashes k-mer l with seeds 0..n-1 (XXH3-64), reduces mod bloom_size and parks the n row numbers in
//      LDS ("hash rows");
//   3. the wave re-maps itself so that LPR = rs/2 adjacent lanes cover one row with 16 bytes each
//      (LPR = 1 and 8 bytes for rs = 1; wide rows: the whole wave, in rs/128 steps): every row costs exactly one
//      coalesced request per 128-byte line, all n loads of a k-mer are issued back-to-back, then ANDed in registers;
//   4. kernel-specific epilogue on the AND words.
// read_id kernels: one wave per read(-pair); see k_readid / k_readid_list.
#include "cid_kernels.hpp"

namespace cid {

// ------------------------------------------------------------------------------------------------
// gather + AND of one k-mer's n rows, this lane's 16-byte (or 8-byte) column slice

struct V16 { uint64_t x, y; };

template <bool NARROW>
__device__ __forceinline__ V16 load_slice(const uint64_t *p) {
    if constexpr (NARROW) {
        return V16{*p, ~0ull};
    } else {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
        return V16{v.x, v.y};
    }
}

// ridx: this wave's row numbers, ridx[s*64 + kmer_in_tile].  ZERO_DETECT also reports whether any of the
// n slices was all-zero in this lane (the caller ANDs those masks across the row's lanes).
template <int NH, bool NARROW, bool ZERO_DETECT>
__device__ __forceinline__ V16 gather_and_fixed(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, int kk,
                                                uint32_t col_word, uint32_t s0, uint32_t &zero_mask) {
    V16 v[NH];
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        const uint64_t row = ridx[(s0 + s) * kWave + kk];
        v[s] = load_slice<NARROW>(mat + row * rs + col_word);
    }
    V16 a{~0ull, ~0ull};
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        if constexpr (ZERO_DETECT) {
            const uint64_t o = NARROW ? v[s].x : (v[s].x | v[s].y);
            zero_mask |= (o == 0) ? (1u << (s0 + s)) : 0u;
        }
        a.x &= v[s].x;
        a.y &= v[s].y;
    }
    return a;
}

template <bool NARROW, bool ZERO_DETECT>
__device__ __forceinline__ V16 gather_and(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, int kk,
                                          uint32_t col_word, uint32_t n, uint32_t &zero_mask) {
    zero_mask = 0;
    switch (n) {  // n is wave-uniform; the common sizes are fully unrolled so all loads are in flight together
    case 1: return gather_and_fixed<1, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, 0, zero_mask);
    case 2: return gather_and_fixed<2, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, 0, zero_mask);
    case 3: return gather_and_fixed<3, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, 0, zero_mask);
    case 4: return gather_and_fixed<4, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, 0, zero_mask);
    default: break;
    }
    V16 a{~0ull, ~0ull};
    uint32_t s = 0;
    for (; s + 4 <= n; s += 4) {
        const V16 b = gather_and_fixed<4, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, s, zero_mask);
        a.x &= b.x; a.y &= b.y;
    }
    for (; s < n; ++s) {
        const V16 b = gather_and_fixed<1, NARROW, ZERO_DETECT>(mat, rs, ridx, kk, col_word, s, zero_mask);
        a.x &= b.x; a.y &= b.y;
    }
    return a;
}

// Sum over the LPR adjacent lanes that share a row (LPR is a power of two <= 64).
template <int LOG_LPR>
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
    for (int o = 1; o < (1 << LOG_LPR); o <<= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Per-colour counting without one atomic per hit: every lane keeps, for its own 128 (or 64) colour bits,
// PLANES bit-sliced binary counters (plane j = bit j of each colour's count).  Adding an AND word is a ripple
// carry over the planes (pure VALU, independent of how many colours are set); after 2^PLANES-1 additions
// the counters are drained into the block's LDS histogram with one atomic per colour seen since the last drain.
template <int PLANES, bool NARROW>
struct VCount {
    V16 pl[PLANES];
    uint32_t adds;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < PLANES; ++j) pl[j] = V16{0, 0};
        adds = 0;
    }
    __device__ __forceinline__ void add(V16 a) {
#pragma unroll
        for (int j = 0; j < PLANES; ++j) {
            const V16 t{pl[j].x & a.x, NARROW ? 0ull : (pl[j].y & a.y)};
            pl[j].x ^= a.x;
            if constexpr (!NARROW) pl[j].y ^= a.y;
            a = t;
        }
        ++adds;  // wave-uniform
    }
    __device__ __forceinline__ bool full() const { return adds == (1u << PLANES) - 1u; }
    __device__ __forceinline__ void drain_word(uint32_t *hist, uint32_t base, bool hi) {
        uint64_t any = 0;
#pragma unroll
        for (int j = 0; j < PLANES; ++j) any |= hi ? pl[j].y : pl[j].x;
        while (any) {
            const uint32_t b = (uint32_t)__builtin_ctzll(any);
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < PLANES; ++j) cnt |= (uint32_t)(((hi ? pl[j].y : pl[j].x) >> b) & 1ull) << j;
            atomicAdd(&hist[base + b], cnt);
            any &= any - 1;
        }
    }
    __device__ __forceinline__ void drain(uint32_t *hist, uint32_t col_word) {
        drain_word(hist, col_word * 64u, false);
        if constexpr (!NARROW) drain_word(hist, col_word * 64u + 64u, true);
        clear();
    }
};

// Steps 1+2 of the header comment for one tile.  Returns nothing; fills ridx[s*64 + lane].
__device__ __forceinline__ void stage_and_hash(uint32_t *img, uint32_t *ridx, const uint8_t *kmers, const uint64_t *codes,
                                               uint64_t n_kmers, uint64_t first, uint32_t k, uint32_t n, const ModMagic &mm,
                                               int lane) {
    wave_lds_fence();  // previous tile's readers are done with img/ridx
    if (codes) {  // packed input: 8 bytes per k-mer, ASCII re-expanded in registers (no LDS image)
        if (first + lane < n_kmers) {
            const uint64_t lsb = rev_fields(codes[first + lane], k);
            xxh3_seeds_from(CodeReader{lsb}, k, n, [&](uint32_t s, uint64_t h) { ridx[s * kWave + lane] = (uint32_t)mod_m(h, mm); });
        } else {
            for (uint32_t s = 0; s < n; ++s) ridx[s * kWave + lane] = 0;
        }
        wave_lds_fence();
        return;
    }
    stage_kmers(img, kmers, n_kmers, first, k, lane);
    wave_lds_fence();
    if (first + lane < n_kmers) {
        xxh3_seeds(img, (uint32_t)lane * k, k, n, [&](uint32_t s, uint64_t h) {
            ridx[s * kWave + lane] = (uint32_t)mod_m(h, mm);
        });
    } else {
        for (uint32_t s = 0; s < n; ++s) ridx[s * kWave + lane] = 0;
    }
    wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------
// a5: proportional search  (src/batch_search_pe.rs:45-84, :125-164)

template <int LOG_LPR, bool NARROW>
__global__ __launch_bounds__(kBlock) void k_search_count(SearchParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr int KPW = kWave / LPR;  // k-mers per sub-pass
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const uint32_t C = p.n_colors;

    uint64_t *s_sum = reinterpret_cast<uint64_t *>(smem);                    // [C] sum of freq of unique hits
    uint32_t *s_hits = reinterpret_cast<uint32_t *>(smem + 8ull * p.c_pad);  // [C]
    uint32_t *s_nu = s_hits + p.c_pad;                                       // [C]
    uint8_t *wbase = smem + 16ull * p.c_pad + (size_t)wave * p.wave_bytes;
    uint32_t *img = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wbase + kmer_img_bytes(p.k));

    for (uint32_t c = threadIdx.x; c < p.c_pad; c += blockDim.x) { s_sum[c] = 0; s_hits[c] = 0; s_nu[c] = 0; }
    __syncthreads();

    const uint64_t n_tiles = (p.n_kmers + kWave - 1) / kWave;
    const uint64_t tile0 = (uint64_t)blockIdx.x * p.tiles_per_block;
    const uint64_t tile1 = tile0 + p.tiles_per_block < n_tiles ? tile0 + p.tiles_per_block : n_tiles;
    const uint32_t col = lane & (LPR - 1);
    const uint32_t col_word = NARROW ? 0u : 2u * col;
    const bool col_live = col_word < p.w64;  // lanes past the row's real width neither load nor count

    VCount<kPlanes, NARROW> vc;
    vc.clear();
    for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) {
        const uint64_t first = tile * kWave;
        stage_and_hash(img, ridx, p.kmers, p.codes, p.n_kmers, first, p.k, p.n_hash, p.mod, lane);
#pragma unroll 1
        for (int sub = 0; sub < LPR; ++sub) {
            const int kk = sub * KPW + (lane >> LOG_LPR);
            const uint64_t kmer = first + kk;
            const bool live = kmer < p.n_kmers;
            V16 a{0, 0};
            uint32_t zm;
            if (live && col_live) a = gather_and<NARROW, false>(p.mat, p.rs, ridx, kk, col_word, p.n_hash, zm);
            if constexpr (NARROW) a.y = 0;
            const uint32_t pc = (uint32_t)(__popcll(a.x) + __popcll(a.y));
            const uint32_t total = group_sum<LOG_LPR>(pc);
            vc.add(a);  // hits[c] += bit c, for this lane's colours
            if (vc.full()) vc.drain(s_hits, col_word);
            if (p.pop_total) {  // striped: uniqueness is decided after all stripes (k_unique_finalize)
                if (live) {
                    if (col == 0) p.pop_total[kmer] += total;
                    if (total == 1u && pc == 1u)
                        p.cand[kmer] = p.colour_base + (a.x ? col_word * 64u + (uint32_t)__builtin_ctzll(a.x)
                                                            : col_word * 64u + 64u + (uint32_t)__builtin_ctzll(a.y));
                }
            } else if (p.want_unique && live) {
                if (total == 1u) {
                    if (pc == 1u) {
                        const uint32_t c = a.x ? col_word * 64u + (uint32_t)__builtin_ctzll(a.x)
                                               : col_word * 64u + 64u + (uint32_t)__builtin_ctzll(a.y);
                        atomicAdd(&s_nu[c], 1u);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&s_sum[c]),
                                  (unsigned long long)(p.freq ? p.freq[kmer] : 1u));
                        if (p.unique_colour) p.unique_colour[kmer] = c;
                    }
                } else if (col == 0 && p.unique_colour) {
                    p.unique_colour[kmer] = 0xFFFFFFFFu;
                }
            }
        }
    }
    vc.drain(s_hits, col_word);
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < C; c += blockDim.x) {
        const uint32_t h = s_hits[c];
        if (h) atomicAdd(reinterpret_cast<unsigned long long *>(&p.hits[c]), (unsigned long long)h);
        if (p.want_unique) {
            const uint32_t u = s_nu[c];
            if (u) {
                if (p.n_unique) atomicAdd(reinterpret_cast<unsigned long long *>(&p.n_unique[c]), (unsigned long long)u);
                if (p.sum_unique_freq)
                    atomicAdd(reinterpret_cast<unsigned long long *>(&p.sum_unique_freq[c]), (unsigned long long)s_sum[c]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a4: perfect search  (src/perfect_search.rs:25-52): AND over every row of every k-mer

template <int LOG_LPR, bool NARROW>
__global__ __launch_bounds__(kBlock) void k_search_perfect(SearchParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr int KPW = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;

    uint64_t *s_and = reinterpret_cast<uint64_t *>(smem);  // [rs] block-level AND
    uint8_t *wbase = smem + 16ull * p.c_pad + (size_t)wave * p.wave_bytes;
    uint32_t *img = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wbase + kmer_img_bytes(p.k));

    for (uint32_t c = threadIdx.x; c < p.rs; c += blockDim.x) s_and[c] = ~0ull;
    __syncthreads();

    const uint64_t n_tiles = (p.n_kmers + kWave - 1) / kWave;
    const uint64_t tile0 = (uint64_t)blockIdx.x * p.tiles_per_block;
    const uint64_t tile1 = tile0 + p.tiles_per_block < n_tiles ? tile0 + p.tiles_per_block : n_tiles;
    const uint32_t col = lane & (LPR - 1);
    const uint32_t col_word = NARROW ? 0u : 2u * col;
    const bool col_live = col_word < p.w64;

    V16 acc{~0ull, ~0ull};
    uint32_t missing = 0;
    for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) {
        const uint64_t first = tile * kWave;
        stage_and_hash(img, ridx, p.kmers, p.codes, p.n_kmers, first, p.k, p.n_hash, p.mod, lane);
#pragma unroll 1
        for (int sub = 0; sub < LPR; ++sub) {
            const int kk = sub * KPW + (lane >> LOG_LPR);
            const bool live = first + kk < p.n_kmers;
            uint32_t zm = 0;
            if (live && col_live) {
                const V16 a = gather_and<NARROW, true>(p.mat, p.rs, ridx, kk, col_word, p.n_hash, zm);
                acc.x &= a.x; acc.y &= a.y;
            } else {
                zm = ~0u;  // a dead lane holds no bits of any row
            }
            // a row is absent (== all-zero) iff every live lane of its group saw a zero slice for that seed
            uint32_t all_zero = zm;
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) all_zero &= __shfl_xor(all_zero, o, kWave);
            const uint32_t seeds = p.n_hash >= 32 ? ~0u : ((1u << p.n_hash) - 1u);
            if (p.zero_acc) {  // striped: a row is absent only if it is zero in every stripe
                if (live && col == 0) p.zero_acc[first + kk] &= (all_zero & seeds);
            } else if (live && (all_zero & seeds)) missing = 1;
        }
    }
    // lanes with the same column slice -> one value per slice per wave
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {
        acc.x &= __shfl_xor(acc.x, o, kWave);
        acc.y &= __shfl_xor(acc.y, o, kWave);
    }
    if (lane < LPR && col_live) {
        atomicAnd(reinterpret_cast<unsigned long long *>(&s_and[col_word]), (unsigned long long)acc.x);
        if (!NARROW) atomicAnd(reinterpret_cast<unsigned long long *>(&s_and[col_word + 1]), (unsigned long long)acc.y);
    }
    if (__any(missing) && lane == 0) atomicOr(p.missing, 1);
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < p.w64; c += blockDim.x)
        atomicAnd(reinterpret_cast<unsigned long long *>(&p.and_words[c]), (unsigned long long)s_and[c]);
}



// ------------------------------------------------------------------------------------------------
// Wide rows (more than 8192 colours; rs = a multiple of 128 words): a whole wave covers one row, 1 KiB per step, one
// k-mer at a time.  These kernels stream KiBs per k-mer, so per-colour results go straight to global atomics.

__device__ __forceinline__ uint32_t wave_and_u32(uint32_t v) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) v &= __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__global__ __launch_bounds__(kBlock) void k_search_count_wide(SearchParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    uint8_t *wbase = smem + (size_t)wave * p.wave_bytes;
    uint32_t *img = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wbase + kmer_img_bytes(p.k));
    const uint64_t n_tiles = (p.n_kmers + kWave - 1) / kWave;
    const uint64_t tile0 = (uint64_t)blockIdx.x * p.tiles_per_block;
    const uint64_t tile1 = tile0 + p.tiles_per_block < n_tiles ? tile0 + p.tiles_per_block : n_tiles;
    const uint32_t steps = p.rs / 128u;
    for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) {
        const uint64_t first = tile * kWave;
        stage_and_hash(img, ridx, p.kmers, p.codes, p.n_kmers, first, p.k, p.n_hash, p.mod, lane);
        const uint32_t cnt = p.n_kmers - first < (uint64_t)kWave ? (uint32_t)(p.n_kmers - first) : (uint32_t)kWave;
        for (uint32_t kk = 0; kk < cnt; ++kk) {
            const uint64_t kmer = first + kk;
            uint32_t mine = 0, ucol = 0;
            for (uint32_t j = 0; j < steps; ++j) {
                const uint32_t col_word = 128u * j + 2u * lane;
                if (col_word >= p.w64) continue;
                uint32_t zm;
                const V16 a = gather_and<false, false>(p.mat, p.rs, ridx, (int)kk, col_word, p.n_hash, zm);
                const uint32_t pc = (uint32_t)(__popcll(a.x) + __popcll(a.y));
                if (!pc) continue;
                mine += pc;
                ucol = a.x ? col_word * 64u + (uint32_t)__builtin_ctzll(a.x) : col_word * 64u + 64u + (uint32_t)__builtin_ctzll(a.y);
                uint64_t w = a.x;
                while (w) { atomicAdd(reinterpret_cast<unsigned long long *>(&p.hits[col_word * 64u + (uint32_t)__builtin_ctzll(w)]), 1ull); w &= w - 1; }
                w = a.y;
                while (w) { atomicAdd(reinterpret_cast<unsigned long long *>(&p.hits[col_word * 64u + 64u + (uint32_t)__builtin_ctzll(w)]), 1ull); w &= w - 1; }
            }
            if (p.want_unique) {
                const uint32_t total = wave_sum_u32(mine);
                if (total == 1u) {
                    if (mine == 1u) {
                        if (p.n_unique) atomicAdd(reinterpret_cast<unsigned long long *>(&p.n_unique[ucol]), 1ull);
                        if (p.sum_unique_freq)
                            atomicAdd(reinterpret_cast<unsigned long long *>(&p.sum_unique_freq[ucol]), (unsigned long long)(p.freq ? p.freq[kmer] : 1u));
                        if (p.unique_colour) p.unique_colour[kmer] = ucol;
                    }
                } else if (lane == 0 && p.unique_colour) {
                    p.unique_colour[kmer] = 0xFFFFFFFFu;
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_search_perfect_wide(SearchParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    uint8_t *wbase = smem + (size_t)wave * p.wave_bytes;
    uint32_t *img = reinterpret_cast<uint32_t *>(wbase);
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wbase + kmer_img_bytes(p.k));
    uint64_t *s_and = reinterpret_cast<uint64_t *>(wbase + ((kmer_img_bytes(p.k) + 4u * kWave * p.n_hash + 15u) & ~15u));  // [rs] per wave
    for (uint32_t w = lane; w < p.rs; w += kWave) s_and[w] = ~0ull;
    const uint64_t n_tiles = (p.n_kmers + kWave - 1) / kWave;
    const uint64_t tile0 = (uint64_t)blockIdx.x * p.tiles_per_block;
    const uint64_t tile1 = tile0 + p.tiles_per_block < n_tiles ? tile0 + p.tiles_per_block : n_tiles;
    const uint32_t steps = p.rs / 128u;
    const uint32_t seeds = p.n_hash >= 32 ? ~0u : ((1u << p.n_hash) - 1u);
    uint32_t missing = 0;
    for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) {
        const uint64_t first = tile * kWave;
        stage_and_hash(img, ridx, p.kmers, p.codes, p.n_kmers, first, p.k, p.n_hash, p.mod, lane);
        const uint32_t cnt = p.n_kmers - first < (uint64_t)kWave ? (uint32_t)(p.n_kmers - first) : (uint32_t)kWave;
        for (uint32_t kk = 0; kk < cnt; ++kk) {
            uint32_t zml = ~0u;
            for (uint32_t j = 0; j < steps; ++j) {
                const uint32_t col_word = 128u * j + 2u * lane;
                if (col_word >= p.w64) continue;
                uint32_t zm;
                const V16 a = gather_and<false, true>(p.mat, p.rs, ridx, (int)kk, col_word, p.n_hash, zm);
                s_and[col_word] &= a.x;       // lane-owned words: plain read-modify-write
                s_and[col_word + 1] &= a.y;
                zml &= zm;
            }
            if (wave_and_u32(zml) & seeds) missing = 1;   // a row is absent iff it is zero in every step of every lane
        }
    }
    wave_lds_fence();
    for (uint32_t w = lane; w < p.w64; w += kWave) atomicAnd(reinterpret_cast<unsigned long long *>(&p.and_words[w]), (unsigned long long)s_and[w]);
    if (missing && lane == 0) atomicOr(p.missing, 1);
}

// read_id over wide rows: the chunk's distinct k-mers one at a time, in order; counts go straight to the (pre-zeroed)
// report row.  s_words / s_R: rs u64 words each per wave (the AND word of the current k-mer, the colours of the first S).
__device__ __forceinline__ void readid_search_chunk_wide(const uint64_t *mat, uint32_t rs, uint32_t w64, uint32_t n, uint32_t C, uint32_t S,
                                                         const uint32_t *ridx, uint64_t *s_words, uint64_t *s_R, uint32_t *row_out,
                                                         uint64_t dmask, uint32_t nd, bool &stopped, int lane) {
    if (stopped || !dmask) return;
    const uint32_t steps = rs / 128u;
    const uint32_t seeds = n >= 32 ? ~0u : ((1u << n) - 1u);
    uint32_t q = nd;
    for (uint64_t dm = dmask; dm; dm &= dm - 1, ++q) {
        const int kk = __builtin_ctzll(dm);
        uint32_t zml = ~0u;
        for (uint32_t j = 0; j < steps; ++j) {
            const uint32_t col_word = 128u * j + 2u * lane;
            if (col_word >= w64) continue;
            uint32_t zm;
            const V16 a = gather_and<false, true>(mat, rs, ridx, kk, col_word, n, zm);
            s_words[col_word] = a.x;
            s_words[col_word + 1] = a.y;
            zml &= zm;
        }
        if (wave_and_u32(zml) & seeds) {  // absent row: *report.entry(no_hits_num) += 1; break
            if (lane == 0) atomicAdd(&row_out[C], 1u);
            stopped = true;
            return;
        }
        for (uint32_t j = 0; j < steps; ++j) {
            const uint32_t col_word = 128u * j + 2u * lane;
            if (col_word >= w64) continue;
            uint64_t x = s_words[col_word], y = s_words[col_word + 1];
            if (S > 0) {
                if (q < S) { s_R[col_word] |= x; s_R[col_word + 1] |= y; }
                else { x &= s_R[col_word]; y &= s_R[col_word + 1]; }
            }
            while (x) { atomicAdd(&row_out[col_word * 64u + (uint32_t)__builtin_ctzll(x)], 1u); x &= x - 1; }
            while (y) { atomicAdd(&row_out[col_word * 64u + 64u + (uint32_t)__builtin_ctzll(y)], 1u); y &= y - 1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a6/a7/a9/a10: per-read classification counts (src/read_id_mt_pe.rs:300-331).  One wave per read(-pair):
//   windows with stride d -> seq::has_no_n filter -> canonical choice on raw bytes (src/kmer.rs:221-243)
//   -> per-read set in first-occurrence order -> search_index_classic (:66-102) or search_index (:104-165)
//   with the reference's "absent row => count it once under no_hits_num and stop" rule applied in k-mer order.
// Two key paths, chosen per read (wave-uniform):
//   packed : the read holds no lower-case base and k <= 32 — bases are packed 2 bits each in LDS, a window's code
//            is three dword reads, the canonical choice is one integer compare, the set is an LDS hash table on the
//            64-bit code (exact), and the hash inputs are re-expanded to ASCII in registers;
//   bytes  : anything else (lower-case bases are hashed as they are; k > 32) — byte strings in
//            LDS, 32-bit tag match confirmed on the bytes.

__device__ __forceinline__ bool good_base(uint32_t b) {  // src/seq.rs:59-64
    const uint32_t u = b & 0xDFu;
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}
__device__ __forceinline__ uint32_t comp_base(uint32_t b) {  // src/kmer.rs:847-863 restricted to ACGTacgt
    const uint32_t low = b & 0x1Fu;
    return b ^ ((low == 1u || low == 0x14u) ? 0x15u : 0x04u);
}
// byte t of the canonical string of the window described by info = pos | rc << 31
__device__ __forceinline__ uint32_t canon_byte(const uint8_t *bases, uint32_t info, uint32_t k, uint32_t t) {
    const uint32_t pos = info & 0x7FFFFFFFu;
    return (info >> 31) ? comp_base(bases[pos + k - 1 - t]) : (uint32_t)bases[pos + t];
}
// find_minimizer (src/kmer.rs:971-986) on a k-byte canonical string `seq` in LDS, byte-wise and case-sensitive as the
// reference compares; candidate = (start i, reverse-complement flag): byte t is seq[i+t] or comp(seq[i+m-1-t]).
__device__ __forceinline__ uint32_t mini_byte(const uint8_t *seq, uint32_t cand, uint32_t m, uint32_t t) {
    const uint32_t i = cand & 0xFFFFu;
    return (cand >> 16) ? comp_base(seq[i + m - 1 - t]) : (uint32_t)seq[i + t];
}
__device__ __forceinline__ bool mini_less(const uint8_t *seq, uint32_t a, uint32_t b, uint32_t m) {
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t x = mini_byte(seq, a, m, t), y = mini_byte(seq, b, m, t);
        if (x != y) return x < y;
    }
    return false;
}
__device__ __forceinline__ uint32_t find_minimizer_bytes(const uint8_t *seq, uint32_t k, uint32_t m) {
    uint32_t best = 0;  // &seq[..m]: position 0, forward only
    for (uint32_t i = 1; i + m <= k; ++i) {
        if (mini_less(seq, i, best, m)) best = i;
        if (mini_less(seq, i | (1u << 16), best, m)) best = i | (1u << 16);
    }
    return best;
}
__device__ __forceinline__ uint8_t upper_base(uint32_t b) { return (uint8_t)((b >= 'a' && b <= 'z') ? b - 32u : b); }

// `nbits` (<= 64) bits starting at bit `bit` of a little-endian dword array (readable 2 dwords past the end)
__device__ __forceinline__ uint64_t bits_at(const uint32_t *w, uint32_t bit, uint32_t nbits) {
    const uint32_t i = bit >> 5, sh = bit & 31u;
    const uint64_t lo = ((uint64_t)w[i + 1] << 32) | w[i];
    uint64_t v = lo >> sh;
    if (sh) v |= (uint64_t)w[i + 2] << (64u - sh);
    return nbits >= 64 ? v : (v & ((1ull << nbits) - 1ull));
}

constexpr int kReadRunUnroll = 2;   // sub-passes of a read's search whose row loads are in flight together
constexpr int kReadPlanes = 3;  // a lane adds one word per sub-pass: drained every 7 additions


// The in-order search (read_id_mt_pe.rs:66-102 classic / :104-165 sampled) over a dense run of distinct k-mers: k-mer j (0 <= j < count, order index q_base + j) has its row
// numbers at ridx[s*stride + j].  U sub-passes (U * 64/LPR k-mers) have all their row loads issued before the first is
// consumed: a read's search is a chain of dependent gather rounds, and what bounds the kernel is how many of them there are.
template <int NH, int U, bool NARROW>
__device__ __forceinline__ void gather_run_fixed(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, uint32_t stride, const uint32_t (&j)[U],
                                                 const bool (&live)[U], uint32_t col_word, uint32_t s0, V16 (&a)[U], uint32_t (&zm)[U]) {
    V16 v[U][NH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const uint64_t row = live[u] ? ridx[(s0 + s) * stride + j[u]] : 0u;   // idle lanes read row 0: no branch, an L2 hit
            v[u][s] = load_slice<NARROW>(mat + row * rs + col_word);
        }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const uint64_t o = NARROW ? v[u][s].x : (v[u][s].x | v[u][s].y);
            zm[u] |= (o == 0) ? (1u << (s0 + s)) : 0u;
            a[u].x &= v[u][s].x;
            a[u].y &= v[u][s].y;
        }
}

template <int LOG_LPR, bool NARROW, int U>
__device__ __forceinline__ void readid_search_run(const uint64_t *mat, uint32_t rs, uint32_t n, uint32_t C, uint32_t S, const uint32_t *ridx,
                                                  uint32_t stride, uint32_t count, uint32_t q_base, uint32_t *hist, bool &stopped,
                                                  VCount<kReadPlanes, NARROW> &vc, V16 &R, int lane) {
    constexpr int LPR = 1 << LOG_LPR;
    constexpr int KPW = kWave / LPR;
    if (stopped || !count) return;
    const uint32_t col = lane & (LPR - 1);
    const uint32_t col_word = NARROW ? 0u : 2u * col;   // slices past the last colour word are zero padding of the row
    const uint32_t seeds_mask = n >= 32 ? ~0u : ((1u << n) - 1u);
#pragma unroll 1
    for (uint32_t j0 = 0; j0 < count; j0 += U * KPW) {
        uint32_t j[U];
        bool live[U];
        V16 a[U];
        uint32_t zm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            j[u] = j0 + u * KPW + (lane >> LOG_LPR);
            live[u] = j[u] < count;
            a[u] = V16{~0ull, ~0ull};
            zm[u] = 0;
        }
        switch (n) {   // n is wave-uniform; the common sizes are fully unrolled
        case 1: gather_run_fixed<1, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, 0, a, zm); break;
        case 2: gather_run_fixed<2, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, 0, a, zm); break;
        case 3: gather_run_fixed<3, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, 0, a, zm); break;
        case 4: gather_run_fixed<4, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, 0, a, zm); break;
        default: {
            uint32_t sd = 0;
            for (; sd + 4 <= n; sd += 4) gather_run_fixed<4, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, sd, a, zm);
            for (; sd < n; ++sd) gather_run_fixed<1, U, NARROW>(mat, rs, ridx, stride, j, live, col_word, sd, a, zm);
        }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u * KPW >= count) break;   // wave-uniform
            V16 w = a[u];
            if constexpr (NARROW) w.y = 0;
            uint32_t all_zero = zm[u];
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) all_zero &= __shfl_xor(all_zero, o, kWave);
            bool lv = live[u];
            const bool miss = lv && (all_zero & seeds_mask);
            const uint64_t bm = __ballot(miss);
            // keep only the k-mers before the first absent row (lane order == k-mer order in a sub-pass)
            if (bm) lv = lv && (lane >> LOG_LPR) < (__builtin_ctzll(bm) >> LOG_LPR);
            if (!lv) { w.x = 0; w.y = 0; }
            if (S > 0) {
                const uint32_t q = q_base + j[u];
                if (q_base + j0 + u * KPW < S) {   // wave-uniform: this sub-pass holds some of the first S k-mers
                    V16 ra = q < S ? w : V16{0, 0};
#pragma unroll
                    for (int o = LPR; o < kWave; o <<= 1) {
                        ra.x |= __shfl_xor(ra.x, o, kWave);
                        ra.y |= __shfl_xor(ra.y, o, kWave);
                    }
                    R.x |= ra.x; R.y |= ra.y;
                }
                if (q >= S) { w.x &= R.x; w.y &= R.y; }
            }
            vc.add(w);
            if (vc.full()) vc.drain(hist, col_word);
            if (bm) {
                stopped = true;
                if (lane == 0) hist[C] += 1;  // *report.entry(no_hits_num) += 1; break
                return;
            }
        }
    }
}

// Output of one read: drain the counters, copy the histogram to the report row, clear it for the next read.
template <bool NARROW, bool WIDE>
__device__ __forceinline__ void readid_finish_read(VCount<kReadPlanes, NARROW> &vc, uint32_t *hist, uint32_t col_word, uint32_t *row_out,
                                                   uint32_t C, int lane) {
    if constexpr (!WIDE) {
        vc.drain(hist, col_word);
        wave_lds_fence();
        for (uint32_t c = lane; c <= C; c += kWave) { row_out[c] = hist[c]; hist[c] = 0; }
    }
}

// ---- k_readid: reads without lower-case bases, k <= 32.  Per-wave LDS: bases | ridx (WIDE only: 64*n) | hist | rall (not
// WIDE: win_cap*n) | table keys + indices | 2-bit bases | bad-base bits.  A read with a lower-case base (its case must be
// kept) is appended to p.redo_list for k_readid_bytes.
template <int LOG_LPR, bool NARROW, bool WIDE, bool MINI>
__global__ __launch_bounds__(kBlock, 4) void k_readid(ReadIdParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t RS = NARROW ? 1u : 2u * LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;
    const uint32_t klen = MINI ? p.m_size : k;   // length of the hashed key

    uint8_t *wb = smem + (size_t)wave * p.wave_bytes;
    uint8_t *s_bases = wb;                                                     // bases_cap
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wb + p.bases_cap);           // WIDE: 64*n, this chunk's rows
    uint32_t *hist = ridx + (WIDE ? kWave * n : 0u);                           // hist_pad
    uint32_t *rall = hist + p.hist_pad;                                        // not WIDE: win_cap*n, rows of the read's distinct k-mers in order
    const uint32_t rcap = p.win_cap;
    unsigned long long *t_key = reinterpret_cast<unsigned long long *>(rall + (WIDE ? 0u : rcap * n));   // table_slots
    uint32_t *t_idx = reinterpret_cast<uint32_t *>(t_key + p.table_slots);     // table_slots
    uint32_t *s_pack = t_idx + p.table_slots;                                  // bases_cap/16 + 4 dwords, 16 bases each
    uint32_t *s_bad = s_pack + (p.bases_cap / 16 + 4);                         // bases_cap/32 + 4 dwords, 1 bit per base

    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;

    const uint32_t col_word = NARROW ? 0u : 2u * (lane & (LPR - 1));
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t tmask = p.table_slots - 1;

    const uint64_t r_begin = (uint64_t)blockIdx.x * p.reads_per_block;
    const uint64_t r_end = r_begin + p.reads_per_block < p.n_reads ? r_begin + p.reads_per_block : p.n_reads;
    for (uint64_t read = r_begin + wave; read < r_end; read += waves) {
        if (p.skip && p.skip[read]) continue;
        wave_lds_fence();
        const uint64_t s0 = p.read_seq0[read], s1 = p.read_seq0[read + 1];
        const uint64_t g0 = p.seq_off[s0];
        const uint32_t first_len = s1 > s0 ? (uint32_t)(p.seq_off[s0 + 1] - g0) : 0u;
        uint32_t *row_out = p.report + read * (uint64_t)(C + 1);
        if (s1 == s0 || first_len < k) {  // too_short: only the first mate is tested (read_id_mt_pe.rs:305)
            if constexpr (!WIDE)  // (wide rows: the host zeroes the whole report before the launch)
                for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) { p.n_kmers[read] = 0; p.status[read] = 1; }
            continue;
        }
        const uint32_t tb = (uint32_t)(p.seq_off[s1] - g0);
        bool lower = false;
        for (uint32_t i = lane; i < tb; i += kWave) {
            const uint8_t b = p.bases[g0 + i];
            s_bases[i] = b;
            lower = lower || (good_base(b) && (b & 0x20u));
        }
        if (__any(lower)) {   // the byte-string kernel takes this read
            if (lane == 0) p.redo_list[atomicAdd(p.redo_count, 1u)] = (uint32_t)read;
            continue;
        }
        wave_lds_fence();
        // 16 bases per lane-step: 2-bit codes (A,C,G,T = 0..3, anything else 0 + its bad bit)
        for (uint32_t j0 = 0; j0 * 16 < tb + 64; j0 += kWave) {
            const uint32_t j = j0 + lane;
            uint32_t code = 0, bad = 0;
            for (uint32_t t = 0; t < 16; ++t) {
                const uint32_t i = j * 16 + t;
                const uint32_t b = i < tb ? s_bases[i] : 'N';
                const uint32_t c2 = (b >> 1) & 3u;            // A 00, C 01, T 10, G 11  ->  swap G/T below
                code |= (c2 ^ (c2 >> 1)) << (2 * t);          // A 0, C 1, G 2, T 3
                bad |= (good_base(b) ? 0u : 1u) << t;
            }
            const uint32_t bad_hi = __shfl_down(bad, 1, kWave);
            if (j * 16 < tb + 64) {
                s_pack[j] = code;
                if (!(lane & 1)) s_bad[j >> 1] = bad | (bad_hi << 16);
            }
        }
        for (uint32_t t = lane; t < p.table_slots; t += kWave) { t_key[t] = ~0ull; t_idx[t] = ~0u; }
        wave_lds_fence();

        uint32_t nd = 0;       // distinct k-mers so far == the reference's `counter`
        uint32_t wbase = 0;    // windows enumerated so far (first-occurrence order index)
        bool stopped = false;  // an absent row was met: nothing after it is searched
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};           // colours seen in the first S k-mers (this lane's slice)
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        for (uint64_t s = s0; s < s1; ++s) {
            const uint32_t off = (uint32_t)(p.seq_off[s] - g0);
            const uint32_t len = (uint32_t)(p.seq_off[s + 1] - p.seq_off[s]);
            if (len < k) continue;  // a mate shorter than k contributes nothing
            const uint32_t nw = (len - k) / p.stride_d + 1;
            for (uint32_t c0 = 0; c0 < nw; c0 += kWave) {
                const uint32_t wi = c0 + lane;
                const uint32_t pos = off + wi * p.stride_d;
                if constexpr (WIDE) wave_lds_fence();  // the previous chunk's gathers are done with ridx
                bool valid = wi < nw;
                uint64_t lsb = 0;
                if (valid) {
                    valid = bits_at(s_bad, pos, k) == 0;                   // seq::has_no_n over the window
                    lsb = bits_at(s_pack, 2 * pos, 2 * k);
                }
                uint64_t msb = 0;
                uint64_t canon = canonical_code(lsb, k, &msb);
                if constexpr (MINI) {  // .mxi: the set holds the k-mers' minimizers (kmer.rs:363-394)
                    msb = minimizer_code(msb, k, klen);
                    canon = rev_fields(msb, klen);
                }
                // exact set with first-occurrence order: slot key = canonical code, slot value = smallest window index
                uint32_t slot = (uint32_t)((msb * 0x9E3779B97F4A7C15ull) >> 40) & tmask;
                if (valid) {
                    while (true) {
                        const unsigned long long old = atomicCAS(&t_key[slot], ~0ull, (unsigned long long)msb);
                        if (old == ~0ull || old == msb) break;
                        slot = (slot + 1) & tmask;
                    }
                    atomicMin(&t_idx[slot], wbase + wi);
                }
                wave_lds_fence();
                const bool distinct = valid && t_idx[slot] == wbase + wi;
                const uint64_t dmask = __ballot(distinct);
                if (distinct) {   // WIDE: this chunk's slot; else the read's list at the k-mer's order index
                    uint32_t *dst = WIDE ? ridx + lane : rall + nd + (uint32_t)__popcll(dmask & lt_mask);
                    const uint32_t st = WIDE ? (uint32_t)kWave : rcap;
                    xxh3_seeds_from(CodeReader{canon}, klen, n, [&](uint32_t sd, uint64_t h) { dst[sd * st] = (uint32_t)mod_m(h, p.mod); });
                }
                if constexpr (WIDE) {
                    wave_lds_fence();
                    uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                    readid_search_chunk_wide(p.mat, p.rs, p.w64, n, C, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane);
                }
                nd += (uint32_t)__popcll(dmask);
            }
            wbase += nw;
        }
        if constexpr (!WIDE) {
            // the set is complete: search its nd k-mers in order, several sub-passes of row loads in flight at a time
            wave_lds_fence();
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll>(p.mat, RS, n, C, S, rall, rcap, nd, 0u, hist, stopped, vc, R, lane);
        }
        readid_finish_read<NARROW, WIDE>(vc, hist, col_word, row_out, C, lane);
        if (lane == 0) { p.n_kmers[read] = nd; p.status[read] = 0; }
    }
}

// ---- k_readid_bytes: the same per-read work on byte strings — reads with lower-case bases (p.redo_list, filled by
// k_readid) or every read when k > 32 (p.redo_list == NULL).  Per-wave LDS: bases | ridx (64*n) | hist | rall | tags |
// window infos | k-mer image | minimizer image + distinct minimizer strings (.mxi).  A 32-bit tag match is confirmed on the bytes.
template <int LOG_LPR, bool NARROW, bool WIDE>
__global__ __launch_bounds__(kBlock, 2) void k_readid_bytes(ReadIdParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t RS = NARROW ? 1u : 2u * LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;

    uint8_t *wb = smem + (size_t)wave * p.wave_bytes;
    uint8_t *s_bases = wb;                                                     // bases_cap
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wb + p.bases_cap);           // 64*n: this chunk's rows
    uint32_t *hist = ridx + kWave * n;                                         // hist_pad
    uint32_t *rall = hist + p.hist_pad;                                        // not WIDE: win_cap*n
    const uint32_t rcap = p.win_cap;
    uint32_t *s_tag = rall + (WIDE ? 0u : rcap * n);                           // win_cap
    uint32_t *s_info = s_tag + p.win_cap;                                      // win_cap
    uint32_t *img = s_info + p.win_cap;                                        // kmer_img_bytes(k)
    uint8_t *img8 = reinterpret_cast<uint8_t *>(img);
    uint32_t *mimg = img + kmer_img_bytes(k) / 4;                              // .mxi only: kmer_img_bytes(m_size) minimizer image
    uint8_t *mimg8 = reinterpret_cast<uint8_t *>(mimg);
    uint8_t *s_mstr = mimg8 + kmer_img_bytes(p.m_size ? p.m_size : 1);         // .mxi only: win_cap x m_size distinct minimizers

    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
    const uint32_t col_word = NARROW ? 0u : 2u * (lane & (LPR - 1));
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    const uint64_t n_items = p.redo_list ? (uint64_t)*p.redo_count : p.n_reads;
    for (uint64_t item = (uint64_t)blockIdx.x * waves + wave; item < n_items; item += (uint64_t)gridDim.x * waves) {
        const uint64_t read = p.redo_list ? (uint64_t)p.redo_list[item] : item;
        if (p.skip && p.skip[read]) continue;
        wave_lds_fence();
        const uint64_t s0 = p.read_seq0[read], s1 = p.read_seq0[read + 1];
        const uint64_t g0 = p.seq_off[s0];
        const uint32_t first_len = s1 > s0 ? (uint32_t)(p.seq_off[s0 + 1] - g0) : 0u;
        uint32_t *row_out = p.report + read * (uint64_t)(C + 1);
        if (s1 == s0 || first_len < k) {  // too_short: only the first mate is tested (read_id_mt_pe.rs:305)
            if constexpr (!WIDE)
                for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) { p.n_kmers[read] = 0; p.status[read] = 1; }
            continue;
        }
        const uint32_t tb = (uint32_t)(p.seq_off[s1] - g0);
        for (uint32_t i = lane; i < tb; i += kWave) s_bases[i] = p.bases[g0 + i];
        wave_lds_fence();

        uint32_t nd = 0;
        bool stopped = false;
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        for (uint64_t s = s0; s < s1; ++s) {
            const uint32_t off = (uint32_t)(p.seq_off[s] - g0);
            const uint32_t len = (uint32_t)(p.seq_off[s + 1] - p.seq_off[s]);
            if (len < k) continue;
            const uint32_t nw = (len - k) / p.stride_d + 1;
            for (uint32_t c0 = 0; c0 < nw; c0 += kWave) {
                const uint32_t wi = c0 + lane;
                const uint32_t pos = off + wi * p.stride_d;
                wave_lds_fence();  // the previous chunk is done with ridx / img
                bool valid = wi < nw;
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) valid = valid && good_base(s_bases[pos + t]);
                uint32_t rc = 1;  // palindromes take the reverse-complement branch (same string)
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) {
                        const uint32_t f = s_bases[pos + t], r = comp_base(s_bases[pos + k - 1 - t]);
                        if (f != r) { rc = f < r ? 0u : 1u; break; }
                    }
                const uint32_t info = pos | (rc << 31);
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) img8[(uint32_t)lane * k + t] = (uint8_t)canon_byte(s_bases, info, k, t);
                wave_lds_fence();
                uint32_t tag = 0;
                bool dup = false;
                const uint64_t vmask = __ballot(valid);
                uint64_t dmask;
                if (p.m_size) {  // .mxi: the key is the (upper-cased) minimizer of the canonical k-mer; strings kept in s_mstr
                    const uint32_t m = p.m_size;
                    if (valid) {
                        const uint8_t *seq = img8 + (uint32_t)lane * k;
                        const uint32_t cand = find_minimizer_bytes(seq, k, m);
                        for (uint32_t t = 0; t < m; ++t) mimg8[(uint32_t)lane * m + t] = upper_base(mini_byte(seq, cand, m, t));
                    }
                    wave_lds_fence();
                    if (valid)
                        xxh3_seeds(mimg, (uint32_t)lane * m, m, n, [&](uint32_t sd, uint64_t h) {
                            if (sd == 0) tag = (uint32_t)h ^ (uint32_t)(h >> 32);
                            ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod);
                        });
                    for (uint32_t q = 0; q < nd; ++q) {
                        if (valid && !dup && s_tag[q] == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < m && same; ++t) same = s_mstr[q * m + t] == mimg8[(uint32_t)lane * m + t];
                            dup = same;
                        }
                    }
                    for (int j = 0; j < kWave - 1; ++j) {
                        if (!((vmask >> j) & 1ull)) continue;
                        const uint32_t tj = __builtin_amdgcn_readlane(tag, j);
                        if (valid && !dup && j < lane && tj == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < m && same; ++t) same = mimg8[(uint32_t)j * m + t] == mimg8[(uint32_t)lane * m + t];
                            dup = same;
                        }
                    }
                    dmask = __ballot(valid && !dup);
                    if (valid && !dup) {
                        const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                        s_tag[q] = tag;
                        for (uint32_t t = 0; t < m; ++t) s_mstr[q * m + t] = mimg8[(uint32_t)lane * m + t];
                    }
                } else {
                    if (valid)
                        xxh3_seeds(img, (uint32_t)lane * k, k, n, [&](uint32_t sd, uint64_t h) {
                            if (sd == 0) tag = (uint32_t)h ^ (uint32_t)(h >> 32);
                            ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod);
                        });
                    for (uint32_t q = 0; q < nd; ++q) {  // against the distinct k-mers of earlier chunks
                        if (valid && !dup && s_tag[q] == tag) {
                            const uint32_t oi = s_info[q];
                            bool same = true;
                            for (uint32_t t = 0; t < k && same; ++t) same = canon_byte(s_bases, oi, k, t) == img8[(uint32_t)lane * k + t];
                            dup = same;
                        }
                    }
                    for (int j = 0; j < kWave - 1; ++j) {  // against lower lanes of this chunk
                        if (!((vmask >> j) & 1ull)) continue;
                        const uint32_t tj = __builtin_amdgcn_readlane(tag, j);
                        const uint32_t ij = __builtin_amdgcn_readlane(info, j);
                        if (valid && !dup && j < lane && tj == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < k && same; ++t) same = canon_byte(s_bases, ij, k, t) == img8[(uint32_t)lane * k + t];
                            dup = same;
                        }
                    }
                    dmask = __ballot(valid && !dup);
                    if (valid && !dup) {
                        const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                        s_tag[q] = tag;
                        s_info[q] = info;
                    }
                }
                if constexpr (WIDE) {
                    wave_lds_fence();
                    uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                    readid_search_chunk_wide(p.mat, p.rs, p.w64, n, C, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane);
                } else if (valid && !dup) {
                    const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                    for (uint32_t sd = 0; sd < n; ++sd) rall[sd * rcap + q] = ridx[sd * kWave + lane];
                }
                nd += (uint32_t)__popcll(dmask);
            }
        }
        if constexpr (!WIDE) {
            wave_lds_fence();
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll>(p.mat, RS, n, C, S, rall, rcap, nd, 0u, hist, stopped, vc, R, lane);
        }
        readid_finish_read<NARROW, WIDE>(vc, hist, col_word, row_out, C, lane);
        if (lane == 0) { p.n_kmers[read] = nd; p.status[read] = 0; }
    }
}

struct BaseReader {  // a key that lives in HBM as a stretch of the read (forward or reverse complement), see k_general_keys
    const uint8_t *b;
    uint32_t len, rc, upper;
    __device__ __forceinline__ uint32_t rd8(uint32_t o) const {
        uint32_t c = rc ? comp_base(b[len - 1 - o]) : (uint32_t)b[o];
        if (upper) c = upper_base(c);
        return c;
    }
    __device__ __forceinline__ uint32_t rd32(uint32_t o) const { return rd8(o) | (rd8(o + 1) << 8) | (rd8(o + 2) << 16) | (rd8(o + 3) << 24); }
    __device__ __forceinline__ uint64_t rd64(uint32_t o) const { return (uint64_t)rd32(o) | ((uint64_t)rd32(o + 4) << 32); }
};

template <int LOG_LPR, bool NARROW, bool WIDE = false>
__global__ __launch_bounds__(kBlock, 4) void k_readid_list(ReadIdListParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;
    uint32_t *ridx = reinterpret_cast<uint32_t *>(smem + (size_t)wave * p.wave_bytes);
    uint32_t *hist = ridx + kWave * n;
    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
    const uint32_t col_word = NARROW ? 0u : 2u * (lane & ((1 << LOG_LPR) - 1));
    for (uint64_t read = (uint64_t)blockIdx.x * waves + wave; read < p.n_reads; read += (uint64_t)gridDim.x * waves) {
        wave_lds_fence();
        uint32_t *row_out = p.report + read * (uint64_t)(C + 1);
        if (p.status[read] == 2) continue;
        if (p.status[read] == 1) {  // too_short, decided on the host side of the call
            if constexpr (!WIDE)
                for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) p.n_kmers[read] = 0;
            continue;
        }
        const uint64_t d0 = p.list_start[read], d1 = p.list_start[read + 1];
        uint32_t nd = 0;
        bool stopped = false;
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        for (uint64_t c0 = d0; c0 < d1 && !stopped; c0 += kWave) {
            const bool have = c0 + lane < d1;
            wave_lds_fence();
            if (have) {
                const uint64_t e = p.list_codes[c0 + lane];
                if (p.bases) {
                    xxh3_seeds_from(BaseReader{p.bases + (e >> 1), k, (uint32_t)(e & 1ull), p.upper}, k, n,
                                    [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod); });
                } else {
                    xxh3_seeds_from(CodeReader{rev_fields(e, k)}, k, n, [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod); });
                }
            }
            const uint64_t dmask = __ballot(have);
            wave_lds_fence();
            if constexpr (WIDE) {
                uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                readid_search_chunk_wide(p.mat, p.rs, p.w64, n, C, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane);
            } else {   // the chunk's entries are dense from lane 0
                readid_search_run<LOG_LPR, NARROW, kReadRunUnroll>(p.mat, NARROW ? 1u : 2u << LOG_LPR, n, C, S, ridx, (uint32_t)kWave,
                                                                   (uint32_t)__popcll(dmask), nd, hist, stopped, vc, R, lane);
            }
            nd += (uint32_t)__popcll(dmask);
        }
        if constexpr (!WIDE) {
            vc.drain(hist, col_word);
            wave_lds_fence();
            for (uint32_t c = lane; c <= C; c += kWave) { row_out[c] = hist[c]; hist[c] = 0; }
        }
        if (lane == 0) p.n_kmers[read] = (uint32_t)(d1 - d0);
    }
}

// Striped a5 epilogue: a k-mer hits exactly one colour of the WHOLE index iff the stripes' popcounts sum to 1.
// Per-block LDS histograms (when the whole colour range fits) keep the global atomics to one per colour per block.
__global__ __launch_bounds__(256) void k_unique_finalize(const uint32_t *pop_total, const uint32_t *cand, const uint32_t *freq,
                                                        uint64_t n_kmers, uint32_t n_colors_total, uint32_t use_lds, uint64_t per_block,
                                                        uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    extern __shared__ __align__(16) uint8_t smem[];
    unsigned long long *s_sum = reinterpret_cast<unsigned long long *>(smem);
    uint32_t *s_nu = reinterpret_cast<uint32_t *>(smem + 8ull * n_colors_total);
    if (use_lds) {
        for (uint32_t c = threadIdx.x; c < n_colors_total; c += blockDim.x) { s_sum[c] = 0; s_nu[c] = 0; }
        __syncthreads();
    }
    const uint64_t i0 = (uint64_t)blockIdx.x * per_block;
    const uint64_t i1 = i0 + per_block < n_kmers ? i0 + per_block : n_kmers;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        if (pop_total[i] == 1u) {
            const uint32_t c = cand[i];
            const unsigned long long f = freq ? freq[i] : 1u;
            if (unique_colour) unique_colour[i] = c;
            if (use_lds) {
                atomicAdd(&s_nu[c], 1u);
                atomicAdd(&s_sum[c], f);
            } else {
                if (n_unique) atomicAdd(reinterpret_cast<unsigned long long *>(&n_unique[c]), 1ull);
                if (sum_unique_freq) atomicAdd(reinterpret_cast<unsigned long long *>(&sum_unique_freq[c]), f);
            }
        } else if (unique_colour) {
            unique_colour[i] = 0xFFFFFFFFu;
        }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < n_colors_total; c += blockDim.x) {
            const uint32_t u = s_nu[c];
            if (!u) continue;
            if (n_unique) atomicAdd(reinterpret_cast<unsigned long long *>(&n_unique[c]), (unsigned long long)u);
            if (sum_unique_freq) atomicAdd(reinterpret_cast<unsigned long long *>(&sum_unique_freq[c]), s_sum[c]);
        }
    }
}

hipError_t launch_unique_finalize(const uint32_t *pop_total, const uint32_t *cand, const uint32_t *freq, uint64_t n_kmers,
                                  uint32_t n_colors_total, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour,
                                  hipStream_t stream) {
    if (n_kmers == 0) return hipSuccess;
    const size_t lds = 12ull * n_colors_total;
    const uint32_t use_lds = lds <= 96u * 1024u ? 1u : 0u;
    const size_t shmem = use_lds ? lds : 0;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_unique_finalize), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    uint64_t per_block = (n_kmers + 4095) / 4096;
    if (per_block < 4096) per_block = 4096;
    const unsigned grid = (unsigned)((n_kmers + per_block - 1) / per_block);
    hipLaunchKernelGGL(k_unique_finalize, dim3(grid), dim3(256), shmem, stream, pop_total, cand, freq, n_kmers, n_colors_total, use_lds, per_block,
                       n_unique, sum_unique_freq, unique_colour);
    return hipGetLastError();
}

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
                if (c < p.n_colors) xxh3_seeds_from(CodeReader{lsb}, klen, p.n_hash, [&](uint32_t, uint64_t h) { set_bit(c, h); });
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
                if (c < p.n_colors) xxh3_seeds(mimg, (uint32_t)lane * p.m_size, p.m_size, p.n_hash, [&](uint32_t, uint64_t h) { set_bit(c, h); });
            }
            continue;
        }
        if (first + lane < p.n_kmers) {
            const uint32_t c = p.colour_of_kmer ? p.colour_of_kmer[first + lane] : p.colour;
            if (c < p.n_colors) xxh3_seeds(img, (uint32_t)lane * p.k, p.k, p.n_hash, [&](uint32_t, uint64_t h) { set_bit(c, h); });
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers

static int log2u(uint32_t v) { int l = 0; while ((1u << l) < v) ++l; return l; }

template <typename KernelT, typename ParamsT>
static hipError_t launch_one(KernelT kernel, int grid, size_t shmem, hipStream_t stream, const ParamsT &p) {
    if (shmem > 64 * 1024) {  // up to the CU's 160 KiB of LDS on request
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), shmem, stream, p);
    return hipGetLastError();
}

#define CID_LAUNCH_BY_LAYOUT(KERNEL, log_lpr, narrow, grid, shmem, stream, params)           \
    do {                                                                                     \
        if (narrow) return launch_one(KERNEL<0, true>, grid, shmem, stream, params);         \
        switch (log_lpr) {                                                                   \
        case 0: return launch_one(KERNEL<0, false>, grid, shmem, stream, params);            \
        case 1: return launch_one(KERNEL<1, false>, grid, shmem, stream, params);            \
        case 2: return launch_one(KERNEL<2, false>, grid, shmem, stream, params);            \
        case 3: return launch_one(KERNEL<3, false>, grid, shmem, stream, params);            \
        case 4: return launch_one(KERNEL<4, false>, grid, shmem, stream, params);            \
        case 5: return launch_one(KERNEL<5, false>, grid, shmem, stream, params);            \
        case 6: return launch_one(KERNEL<6, false>, grid, shmem, stream, params);            \
        default: return hipErrorInvalidValue;                                                \
        }                                                                                    \
    } while (0)

size_t search_smem_bytes(const SearchParams &p) { return 16ull * p.c_pad + (size_t)(kBlock / kWave) * p.wave_bytes; }

int grid_for(uint64_t n_kmers, uint32_t tiles_per_block) {
    const uint64_t n_tiles = (n_kmers + kWave - 1) / kWave;
    return (int)((n_tiles + tiles_per_block - 1) / tiles_per_block);
}

hipError_t launch_search_count(const SearchParams &p, hipStream_t stream) {
    if (p.rs > 128) {
        const int g = grid_for(p.n_kmers, p.tiles_per_block);
        return g ? launch_one(k_search_count_wide, g, (size_t)(kBlock / kWave) * p.wave_bytes, stream, p) : hipSuccess;
    }
    const bool narrow = p.rs == 1;
    const int log_lpr = narrow ? 0 : log2u(p.rs / 2);
    const size_t shmem = search_smem_bytes(p);
    const int grid = grid_for(p.n_kmers, p.tiles_per_block);
    if (grid == 0) return hipSuccess;
    CID_LAUNCH_BY_LAYOUT(k_search_count, log_lpr, narrow, grid, shmem, stream, p);
}

hipError_t launch_search_perfect(const SearchParams &p, hipStream_t stream) {
    if (p.rs > 128) {
        const int g = grid_for(p.n_kmers, p.tiles_per_block);
        return g ? launch_one(k_search_perfect_wide, g, (size_t)(kBlock / kWave) * p.wave_bytes, stream, p) : hipSuccess;
    }
    const bool narrow = p.rs == 1;
    const int log_lpr = narrow ? 0 : log2u(p.rs / 2);
    const size_t shmem = search_smem_bytes(p);
    const int grid = grid_for(p.n_kmers, p.tiles_per_block);
    if (grid == 0) return hipSuccess;
    CID_LAUNCH_BY_LAYOUT(k_search_perfect, log_lpr, narrow, grid, shmem, stream, p);
}

template <typename KernelT>
static hipError_t launch_readid_one(KernelT kernel, const ReadIdParams &p, int waves_per_block, int grid, hipStream_t stream) {
    const size_t shmem = (size_t)waves_per_block * p.wave_bytes;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves_per_block * kWave), shmem, stream, p);
    return hipGetLastError();
}

template <bool MINI>
static hipError_t launch_readid_packed(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return launch_readid_one(k_readid<0, false, true, MINI>, p, wpb, grid, stream);
    if (p.rs == 1) return launch_readid_one(k_readid<0, true, false, MINI>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid<0, false, false, MINI>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid<1, false, false, MINI>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid<2, false, false, MINI>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid<3, false, false, MINI>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid<4, false, false, MINI>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid<5, false, false, MINI>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid<6, false, false, MINI>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

// k <= 32, no lower-case base: one block per p.reads_per_block reads
hipError_t launch_readid(const ReadIdParams &p, int waves_per_block, hipStream_t stream) {
    const int grid = (int)((p.n_reads + p.reads_per_block - 1) / p.reads_per_block);
    return p.m_size ? launch_readid_packed<true>(p, waves_per_block, grid, stream) : launch_readid_packed<false>(p, waves_per_block, grid, stream);
}

// byte-string keys: the reads listed in p.redo_list (count on the device), or all of them
hipError_t launch_readid_bytes(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return launch_readid_one(k_readid_bytes<0, false, true>, p, wpb, grid, stream);
    if (p.rs == 1) return launch_readid_one(k_readid_bytes<0, true, false>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid_bytes<0, false, false>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid_bytes<1, false, false>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid_bytes<2, false, false>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid_bytes<3, false, false>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid_bytes<4, false, false>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid_bytes<5, false, false>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid_bytes<6, false, false>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

template <typename KernelT>
static hipError_t launch_readid_list_one(KernelT kernel, const ReadIdListParams &p, int grid, hipStream_t stream) {
    const size_t shmem = (size_t)(kBlock / kWave) * p.wave_bytes;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), shmem, stream, p);
    return hipGetLastError();
}

hipError_t launch_readid_list(const ReadIdListParams &p, int grid, hipStream_t stream) {
    if (p.n_reads == 0) return hipSuccess;
    if (p.rs > 128) return launch_readid_list_one(k_readid_list<0, false, true>, p, grid, stream);
    if (p.rs == 1) return launch_readid_list_one(k_readid_list<0, true>, p, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_list_one(k_readid_list<0, false>, p, grid, stream);
    case 1: return launch_readid_list_one(k_readid_list<1, false>, p, grid, stream);
    case 2: return launch_readid_list_one(k_readid_list<2, false>, p, grid, stream);
    case 3: return launch_readid_list_one(k_readid_list<3, false>, p, grid, stream);
    case 4: return launch_readid_list_one(k_readid_list<4, false>, p, grid, stream);
    case 5: return launch_readid_list_one(k_readid_list<5, false>, p, grid, stream);
    case 6: return launch_readid_list_one(k_readid_list<6, false>, p, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_put_rows(uint64_t *mat, uint32_t rs, const uint64_t *d_row_ids, const uint32_t *d_words, uint32_t w32,
                           uint64_t n_rows, hipStream_t stream) {
    const uint64_t n = n_rows * w32;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_put_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(mat), rs, d_row_ids, d_words, w32, n_rows);
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
