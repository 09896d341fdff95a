// Device helpers shared by the search, read_id and index kernels (gfx950, wave64): the 16-byte column slice of a row,
// gather + AND of a k-mer's rows, bit-sliced per-colour counters, tile staging + hashing, launch plumbing.
#pragma once
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

// Mixed gather for 32-byte rows (rs = 4, two lanes per row): the k-mer's LAST row comes through the scalar data cache, the others
// through the vector path.  A vector miss always costs a whole 128-byte L2 line from HBM, a scalar miss a 64-byte one, and the
// two paths queue separately: tools/gather_probe variant 13 (this shape, bare) gathers 6-7 % faster than the all-vector form.
// All 64 lanes must call it (the scalar part is wave-uniform work): lane 2j / 2j+1 own the two 16-byte halves of the rows of
// k-mer kk0 + j; rlast = lane l holds the last row number of the tile's k-mer l.  Dead k-mers carry row 0.
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // M0 is "reserved" to the compiler's mind; nothing here depends on its value
// v_writelane with an SGPR value takes its lane select from M0 (one SGPR on the constant bus); LDS instructions no longer use M0
#define CID_WRITELANE(acc, val, ln) asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(acc) : "s"(val), "s"(ln) : "m0")
template <int NH>
__device__ __forceinline__ V16 gather_and_mixed32(const uint64_t *mat, const uint32_t *ridx, int kk, uint32_t col_word, uint32_t rlast, int kk0) {
    V16 v[NH - 1 > 0 ? NH - 1 : 1];
#pragma unroll
    for (int s = 0; s < NH - 1; ++s) {
        const uint64_t row = ridx[s * kWave + kk];
        v[s] = load_slice<false>(mat + row * 4u + col_word);
    }
    typedef const __attribute__((address_space(4))) u32x8_t *cptr;
    uint32_t t0 = ~0u, t1 = ~0u, t2 = ~0u, t3 = ~0u;
    constexpr int kBatch = 4;   // scalar loads in flight per round: 8 SGPRs each (the probe: 4 and 8 gather equally fast)
#pragma unroll 1
    for (int b = 0; b < 32; b += kBatch) {
        u32x8_t q[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)rlast, kk0 + b + u);
            q[u] = *(cptr)(mat + (uint64_t)row * 4u);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int l0 = 2 * (b + u);
            CID_WRITELANE(t0, q[u][0], l0); CID_WRITELANE(t0, q[u][4], l0 + 1);
            CID_WRITELANE(t1, q[u][1], l0); CID_WRITELANE(t1, q[u][5], l0 + 1);
            CID_WRITELANE(t2, q[u][2], l0); CID_WRITELANE(t2, q[u][6], l0 + 1);
            CID_WRITELANE(t3, q[u][3], l0); CID_WRITELANE(t3, q[u][7], l0 + 1);
        }
    }
    V16 a{((uint64_t)t1 << 32) | t0, ((uint64_t)t3 << 32) | t2};
#pragma unroll
    for (int s = 0; s < NH - 1; ++s) { a.x &= v[s].x; a.y &= v[s].y; }
    return a;
}
#pragma clang diagnostic pop

// Branch-free variant for several sub-passes at once: k-mer j[u] has its row numbers at ridx[s*stride + j[u]].
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

// U sub-passes' worth of row loads (U x n per lane) issued before any is consumed; a[u] must start as all-ones, zm[u] as 0.
template <int U, bool NARROW>
__device__ __forceinline__ void gather_run(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, uint32_t stride, const uint32_t (&j)[U],
                                           const bool (&live)[U], uint32_t col_word, uint32_t n, V16 (&a)[U], uint32_t (&zm)[U]) {
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
}

// U k-mers' rows (U x n loads per lane) issued before any is consumed, branch-free: a lane that is not `ok` reads row 0's first
// slice (an L2 hit) and gets zeros.
template <int NH, int U, bool NARROW>
__device__ __forceinline__ void gather_and_multi_fixed(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, const uint32_t (&kk)[U],
                                                       const bool (&ok)[U], uint32_t col_word, uint32_t s0, V16 (&a)[U]) {
    V16 v[U][NH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const uint64_t row = ok[u] ? ridx[(s0 + s) * kWave + kk[u]] : 0u;
            v[u][s] = load_slice<NARROW>(mat + row * rs + (ok[u] ? col_word : 0u));
        }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < NH; ++s) { a[u].x &= v[u][s].x; a[u].y &= v[u][s].y; }
}
template <int U, bool NARROW>
__device__ __forceinline__ void gather_and_multi(const uint64_t *mat, uint32_t rs, const uint32_t *ridx, const uint32_t (&kk)[U],
                                                 const bool (&ok)[U], uint32_t col_word, uint32_t n, V16 (&a)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = V16{~0ull, ~0ull};
    switch (n) {   // n is wave-uniform
    case 1: gather_and_multi_fixed<1, U, NARROW>(mat, rs, ridx, kk, ok, col_word, 0, a); break;
    case 2: gather_and_multi_fixed<2, U, NARROW>(mat, rs, ridx, kk, ok, col_word, 0, a); break;
    case 3: gather_and_multi_fixed<3, U, NARROW>(mat, rs, ridx, kk, ok, col_word, 0, a); break;
    case 4: gather_and_multi_fixed<4, U, NARROW>(mat, rs, ridx, kk, ok, col_word, 0, a); break;
    default: {
        uint32_t sd = 0;
        for (; sd + 4 <= n; sd += 4) gather_and_multi_fixed<4, U, NARROW>(mat, rs, ridx, kk, ok, col_word, sd, a);
        for (; sd < n; ++sd) gather_and_multi_fixed<1, U, NARROW>(mat, rs, ridx, kk, ok, col_word, sd, a);
    }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (!ok[u]) a[u] = V16{0, 0};
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
            xxh3_seeds_from(CodeReader{lsb}, k, n, HashSel::of(mm), [&](uint32_t s, uint64_t h) { ridx[s * kWave + lane] = (uint32_t)mod_m(h, mm); });
        } else {
            for (uint32_t s = 0; s < n; ++s) ridx[s * kWave + lane] = 0;
        }
        wave_lds_fence();
        return;
    }
    stage_kmers(img, kmers, n_kmers, first, k, lane);
    wave_lds_fence();
    if (first + lane < n_kmers) {
        xxh3_seeds(img, (uint32_t)lane * k, k, n, HashSel::of(mm), [&](uint32_t s, uint64_t h) {
            ridx[s * kWave + lane] = (uint32_t)mod_m(h, mm);
        });
    } else {
        for (uint32_t s = 0; s < n; ++s) ridx[s * kWave + lane] = 0;
    }
    wave_lds_fence();
}

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


// A value every lane of the wave holds alike, moved to scalar registers.  (A load through a wave-uniform address comes back in a
// vector register, and what is computed from it stays there, when stores of the kernel may alias it: per-read offsets and sizes
// handled this way cost k_readid a dozen VGPRs and the spills that go with them.)
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
    return ((uint64_t)wave_uniform((uint32_t)(v >> 32)) << 32) | wave_uniform((uint32_t)v);
}

// ------------------------------------------------------------------------------------------------ bases as bytes
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

// Four bases held in one dword: their 2-bit codes (A 0, C 1, G 2, T 3 in either case; any other byte: what its bits 1-2 say) as 8 bits,
// and one bit per base for "none of ACGTacgt" (seq.rs:59-64) and for "bit 5 set" (lower case, if it is a base at all).
__device__ __forceinline__ uint32_t byte_tops_to_nibble(uint32_t y) {   // bits 7, 15, 23, 31 -> bits 0..3
    y >>= 7;
    return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xFu;
}
__device__ __forceinline__ void pack_four_bases(uint32_t w, uint32_t &code8, uint32_t &bad4, uint32_t &low4) {
    uint32_t x = (w >> 1) & 0x03030303u;            // A 00, C 01, T 10, G 11
    x ^= (x >> 1) & 0x01010101u;                    // A 0, C 1, G 2, T 3
    code8 = (x | (x >> 6) | (x >> 12) | (x >> 18)) & 0xFFu;
    const uint32_t u = w & 0xDFDFDFDFu;
    auto nz = [](uint32_t z) { return ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z; };   // bit 7 of each byte: the byte is not zero
    bad4 = byte_tops_to_nibble(nz(u ^ 0x41414141u) & nz(u ^ 0x43434343u) & nz(u ^ 0x47474747u) & nz(u ^ 0x54545454u) & 0x80808080u);
    low4 = byte_tops_to_nibble((w << 2) & 0x80808080u);
}

// `nbits` (<= 64) bits starting at bit `bit` of a little-endian dword array (readable 2 dwords past the end)
__device__ __forceinline__ uint64_t bits_at(const uint32_t *w, uint32_t bit, uint32_t nbits) {
    const uint32_t i = bit >> 5, sh = bit & 31u;
    const uint64_t lo = ((uint64_t)w[i + 1] << 32) | w[i];
    uint64_t v = lo >> sh;
    if (sh) v |= (uint64_t)w[i + 2] << (64u - sh);
    return nbits >= 64 ? v : (v & ((1ull << nbits) - 1ull));
}

// ------------------------------------------------------------------------------------------------ launch plumbing
static inline int log2u(uint32_t v) { int l = 0; while ((1u << l) < v) ++l; return l; }

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

#define CID_LAUNCH_BY_LAYOUT2(KERNEL, log_lpr, narrow, FLAG, grid, shmem, stream, params)           \
    do {                                                                                           \
        if (narrow) return launch_one(KERNEL<0, true, FLAG>, grid, shmem, stream, params);         \
        switch (log_lpr) {                                                                         \
        case 0: return launch_one(KERNEL<0, false, FLAG>, grid, shmem, stream, params);            \
        case 1: return launch_one(KERNEL<1, false, FLAG>, grid, shmem, stream, params);            \
        case 2: return launch_one(KERNEL<2, false, FLAG>, grid, shmem, stream, params);            \
        case 3: return launch_one(KERNEL<3, false, FLAG>, grid, shmem, stream, params);            \
        case 4: return launch_one(KERNEL<4, false, FLAG>, grid, shmem, stream, params);            \
        case 5: return launch_one(KERNEL<5, false, FLAG>, grid, shmem, stream, params);            \
        case 6: return launch_one(KERNEL<6, false, FLAG>, grid, shmem, stream, params);            \
        default: return hipErrorInvalidValue;                                                      \
        }                                                                                          \
    } while (0)

}  // namespace cid
