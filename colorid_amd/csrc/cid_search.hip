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
// This file: a5 (k_search_count), a4 (k_search_perfect), their wide-row variants and the colour-stripe finalize.
#include "cid_gather.hpp"

namespace cid {

// the XCD (0..7) this wave runs on: HW_REG_XCC_ID (hardware register 20), bits 3:0
__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }
constexpr uint32_t kQueueStride = 32;   // one work-queue head per 128-byte line

// ------------------------------------------------------------------------------------------------
// Colour stripes: the one per-k-mer fact that needs every stripe is "the AND word over ALL colours has exactly one set bit"
// (src/batch_search_pe.rs:75-82).  It travels as one u32 per k-mer,  n << 26 | (colour + 1),  where n = min(set bits, 2) over
// the stripes seen so far (saturating inside a GPU) and the colour field is only kept while n == 1.  Across GPUs the words are
// simply SUMMED (one RCCL all-reduce): the n fields add up to <= 2 * ranks (6 bits: up to 31 ranks) and, whenever the total is
// 1, exactly one rank contributed a colour field, so the low 26 bits are that colour + 1 (colours < 2^20, so up to 64 ranks'
// fields cannot carry into n).
constexpr uint32_t kFactShift = 26;
constexpr uint32_t kFactColourMask = (1u << kFactShift) - 1u;
__device__ __forceinline__ uint32_t stripe_fact_merge(uint32_t old, uint32_t pop, uint32_t global_colour) {
    const uint32_t n = min((old >> kFactShift) + min(pop, 2u), 2u);
    const uint32_t col = n == 1u ? (pop == 1u ? global_colour + 1u : (old & kFactColourMask)) : 0u;
    return (n << kFactShift) | col;
}

// ------------------------------------------------------------------------------------------------
// a5: proportional search  (src/batch_search_pe.rs:45-84, :125-164)

// PERSIST: the grid is sized to what is resident at once and every wave pulls 64-k-mer tiles from eight work queues, one per
// XCD (its own first, the others once that is empty).  Queue x walks the x-th eighth of the k-mer array front to back, so
// the waves of one XCD — which share that XCD's 4 MiB L2 — work on a window of a few ten thousand consecutive k-mers.  When the
// producer has grouped the k-mers by the index slice of their first row (cid_kmerset_order_for_index), that window's first-row
// lines are L2 hits.  The XCD a wave runs on is read from the hardware (HW_REG_XCC_ID), not inferred from blockIdx.
template <int LOG_LPR, bool NARROW, bool PERSIST, int UNROLL = 1>
__global__ __launch_bounds__(kBlock, UNROLL == 1 ? 4 : 3) void k_search_count(SearchParams p) {
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
    // Per-k-mer results are parked here and leave once per tile as one coalesced 256-byte store: written straight from the
    // sub-passes they were 32..128-byte partial-line stores and cost 0.6 ms (C = 256) to 1.9 ms (C = 1024) per 120 M k-mers.
    uint32_t *s_res = ridx + kWave * p.n_hash;   // [64] unique colour / stripe candidate colour
    uint32_t *s_pop = s_res + kWave;              // [64] striped: popcount of the stripe's AND word; else the k-mers' multiplicities

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
    auto do_tile = [&](uint64_t tile) {
        const uint64_t first = tile * kWave;
        // the tile's multiplicities: one coalesced load that is back long before the first sub-pass needs it
        const uint32_t my_freq = (p.freq && p.want_unique && !p.fact && first + lane < p.n_kmers) ? p.freq[first + lane] : 1u;
        stage_and_hash(img, ridx, p.kmers, p.codes, p.n_kmers, first, p.k, p.n_hash, p.mod, lane);
        if (!p.fact) { s_pop[lane] = my_freq; wave_lds_fence(); }
        // what follows a k-mer's gather: count its AND word's bits, decide uniqueness
        auto after_gather = [&](const int kk, const bool live, V16 a) {
            if constexpr (NARROW) a.y = 0;
            const uint32_t pc = (uint32_t)(__popcll(a.x) + __popcll(a.y));
            const uint32_t total = group_sum<LOG_LPR>(pc);
            vc.add(a);  // hits[c] += bit c, for this lane's colours
            if (vc.full()) vc.drain(s_hits, col_word);
            if (p.fact) {  // striped: uniqueness is decided after all stripes (k_unique_finalize)
                if (live) {
                    if (col == 0) { s_pop[kk] = total; s_res[kk] = 0xFFFFFFFFu; }
                    if (total == 1u && pc == 1u)   // (after col 0's store in program order: LDS ops of a wave execute in order)
                        s_res[kk] = p.colour_base + (a.x ? col_word * 64u + (uint32_t)__builtin_ctzll(a.x)
                                                         : col_word * 64u + 64u + (uint32_t)__builtin_ctzll(a.y));
                }
            } else if (p.want_unique && live) {
                if (total == 1u) {
                    if (pc == 1u) {
                        const uint32_t c = a.x ? col_word * 64u + (uint32_t)__builtin_ctzll(a.x)
                                               : col_word * 64u + 64u + (uint32_t)__builtin_ctzll(a.y);
                        atomicAdd(&s_nu[c], 1u);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&s_sum[c]), (unsigned long long)s_pop[kk]);
                        s_res[kk] = c;
                    }
                } else if (col == 0) {
                    s_res[kk] = 0xFFFFFFFFu;
                }
            }
        };
        constexpr int U = UNROLL < LPR ? UNROLL : LPR;   // sub-passes whose row loads are issued together
#pragma unroll 1
        for (int sub = 0; sub < LPR; sub += U) {
            if constexpr (U > 1) {   // rows of 64 bytes and more: a sub-pass covers only 64/LPR k-mers, so several are in flight at once
                uint32_t kk[U];
                bool live[U], ok[U];
                V16 a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    kk[u] = (uint32_t)((sub + u) * KPW + (lane >> LOG_LPR));
                    live[u] = first + kk[u] < p.n_kmers;
                    ok[u] = live[u] && col_live;
                }
                gather_and_multi<U, NARROW>(p.mat, p.rs, ridx, kk, ok, col_word, p.n_hash, a);
#pragma unroll
                for (int u = 0; u < U; ++u) after_gather((int)kk[u], live[u], a[u]);
                continue;
            }
            const int kk = sub * KPW + (lane >> LOG_LPR);
            const uint64_t kmer = first + kk;
            const bool live = kmer < p.n_kmers;
            V16 a{0, 0};
            uint32_t zm;
            bool mixed = false;
#ifdef CID_TUNE_BUILD
            if constexpr (LOG_LPR == 1 && !NARROW) {   // 32-byte rows: the last row of every k-mer through the scalar cache
                if (p.mixed && p.n_hash >= 2 && p.n_hash <= 4) {
                    mixed = true;
                    const uint32_t rlast = ridx[(p.n_hash - 1) * kWave + lane];
                    V16 m;
                    switch (p.n_hash) {
                    case 2: m = gather_and_mixed32<2>(p.mat, ridx, kk, col_word, rlast, sub * KPW); break;
                    case 3: m = gather_and_mixed32<3>(p.mat, ridx, kk, col_word, rlast, sub * KPW); break;
                    default: m = gather_and_mixed32<4>(p.mat, ridx, kk, col_word, rlast, sub * KPW); break;
                    }
                    if (live) a = m;
                }
            }
#endif
            if (!mixed && live && col_live) a = gather_and<NARROW, false>(p.mat, p.rs, ridx, kk, col_word, p.n_hash, zm);
            after_gather(kk, live, a);
        }
        if (p.fact || (p.want_unique && p.unique_colour)) {   // the tile's per-k-mer results, one coalesced store
            wave_lds_fence();
            const uint64_t kmer = first + lane;
            if (kmer < p.n_kmers) {
                if (p.fact) {
                    p.fact[kmer] = stripe_fact_merge(p.fact[kmer], s_pop[lane], s_res[lane]);
                } else {
                    __builtin_nontemporal_store(s_res[lane], &p.unique_colour[kmer]);   // written once, never read here
                }
            }
        }
    };
    if constexpr (PERSIST) {
        const uint64_t seg = (n_tiles + 7) / 8;                  // tiles per queue
        uint32_t q = xcc_id();                                     // wave-uniform: this wave's XCD
        uint32_t tried = 0;
        // the next ticket is drawn before the current tile is worked on, so the atomic's round trip hides behind the tile
        uint32_t ticket = 0;
        auto draw = [&](uint32_t queue) { if (lane == 0) ticket = atomicAdd(&p.queues[queue * kQueueStride], 1u); };
        draw(q);
        while (tried < 8) {
            const uint64_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            const uint64_t q_first = (uint64_t)q * seg;
            const uint64_t q_len = q_first >= n_tiles ? 0 : (n_tiles - q_first < seg ? n_tiles - q_first : seg);
            if (t >= q_len) {   // this queue is empty: on to the next one
                ++tried;
                q = (q + 1) & 7u;
                if (tried < 8) draw(q);
                continue;
            }
            draw(q);
            do_tile(q_first + t);
        }
    } else {
        for (uint64_t tile = tile0 + wave; tile < tile1; tile += kBlock / kWave) do_tile(tile);
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
            if (p.fact) {  // striped: uniqueness is decided after all stripes (k_unique_finalize)
                const uint32_t total = wave_sum_u32(mine);
                const uint32_t gcol = (uint32_t)__shfl((int)(p.colour_base + ucol), __builtin_ctzll(__ballot(mine != 0) | (1ull << 63)), kWave);
                if (lane == 0) p.fact[kmer] = stripe_fact_merge(p.fact[kmer], total, gcol);
            } else if (p.want_unique) {
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
            const uint32_t all_zero = wave_and_u32(zml) & seeds;   // a row is absent iff it is zero in every step of every lane
            if (p.zero_acc) {  // striped: a row is absent only if it is zero in every stripe
                if (lane == 0) p.zero_acc[first + kk] &= all_zero;
            } else if (all_zero) missing = 1;
        }
    }
    wave_lds_fence();
    for (uint32_t w = lane; w < p.w64; w += kWave) atomicAnd(reinterpret_cast<unsigned long long *>(&p.and_words[w]), (unsigned long long)s_and[w]);
    if (missing && lane == 0) atomicOr(p.missing, 1);
}

// Striped a5 epilogue: a k-mer hits exactly one colour of the WHOLE index iff the stripes' popcounts sum to 1.
// Per-block LDS histograms (when the whole colour range fits) keep the global atomics to one per colour per block.
__global__ __launch_bounds__(256) void k_unique_finalize(const uint32_t *fact, const uint32_t *freq,
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
        const uint32_t f = fact[i];
        if ((f >> kFactShift) == 1u) {
            const uint32_t c = (f & kFactColourMask) - 1u;
            const unsigned long long fq = freq ? freq[i] : 1u;
            if (unique_colour) unique_colour[i] = c;
            if (use_lds) {
                atomicAdd(&s_nu[c], 1u);
                atomicAdd(&s_sum[c], fq);
            } else {
                if (n_unique) atomicAdd(reinterpret_cast<unsigned long long *>(&n_unique[c]), 1ull);
                if (sum_unique_freq) atomicAdd(reinterpret_cast<unsigned long long *>(&sum_unique_freq[c]), fq);
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

hipError_t launch_unique_finalize(const uint32_t *fact, const uint32_t *freq, uint64_t n_kmers,
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
    hipLaunchKernelGGL(k_unique_finalize, dim3(grid), dim3(256), shmem, stream, fact, freq, n_kmers, n_colors_total, use_lds, per_block,
                       n_unique, sum_unique_freq, unique_colour);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launchers

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
    int grid = grid_for(p.n_kmers, p.tiles_per_block);
    if (grid == 0) return hipSuccess;
#ifdef CID_TUNE_BUILD   // the rejected schedulings exist only in libcolorid_hip_tune.so
    if (p.queues) {
        if (p.persist_grid < grid) grid = p.persist_grid;
        CID_LAUNCH_BY_LAYOUT2(k_search_count, log_lpr, narrow, true, grid, shmem, stream, p);
    }
#endif
    // rows of 64 and 128 bytes (a sub-pass covers only 16 or 8 k-mers): two sub-passes' row loads in flight per lane, -1.5 %
    // (tools/exp_unroll.py); wider rows gain nothing from it
    if (p.unroll == 2 && !narrow && log_lpr == 2) return launch_one(k_search_count<2, false, false, 2>, grid, shmem, stream, p);
    if (p.unroll == 2 && !narrow && log_lpr == 3) return launch_one(k_search_count<3, false, false, 2>, grid, shmem, stream, p);
    CID_LAUNCH_BY_LAYOUT2(k_search_count, log_lpr, narrow, false, grid, shmem, stream, p);
}

#ifdef CID_TUNE_BUILD
// how many blocks of the persistent kernel are resident at once (per CU, by LDS and registers)
int search_count_blocks_per_cu(const SearchParams &p) {
    const size_t shmem = search_smem_bytes(p);
    const bool narrow = p.rs == 1;
    const int log_lpr = narrow ? 0 : log2u(p.rs / 2);
    int nb = 0;
    auto ask = [&](auto kernel) {
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, shmem) != hipSuccess) nb = 0;
    };
    if (narrow) ask(k_search_count<0, true, true>);
    else switch (log_lpr) {
        case 0: ask(k_search_count<0, false, true>); break;
        case 1: ask(k_search_count<1, false, true>); break;
        case 2: ask(k_search_count<2, false, true>); break;
        case 3: ask(k_search_count<3, false, true>); break;
        case 4: ask(k_search_count<4, false, true>); break;
        case 5: ask(k_search_count<5, false, true>); break;
        case 6: ask(k_search_count<6, false, true>); break;
        default: break;
    }
    return nb;
}
#endif

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

hipError_t warm_search() {   // see warm_readid
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_unique_finalize));
}

}  // namespace cid
