// Integer arithmetic of the BIGSI hash step, shared verbatim by the gfx950 kernels and by the CPU unit
// test of that arithmetic (tests/cpu_shim/, compiled with g++): XXH3-64-with-seed for 1..128-byte inputs
// read out of a byte image (LDS on the device) and the exact `% bloom_size`.
//
// Under hipcc every function here is a __device__ function; nothing in the product's host code calls them.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define CID_FN __device__ __forceinline__
#define CID_DEVCONST __device__
#else
#define CID_FN inline
#define CID_DEVCONST
#endif

namespace cid {

CID_FN uint64_t umul64hi(uint64_t a, uint64_t b) {
#ifdef __HIPCC__
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// ({hi,lo} >> 8*sh) & 0xffffffff, sh in 0..3  (v_alignbyte_b32)
CID_FN uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
#ifdef __HIPCC__
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * sh));
#endif
}

// ---------------------------------------------------------------- exact modulo by an invariant

struct ModMagic {
    uint64_t m;      // divisor (bloom_size)
    uint64_t magic;  // multiplier
    uint32_t shift;
    uint32_t flags;  // 1 = power of two (mask), 2 = "add" fix-up step; bits 8..15 = the index's hash variant (kHash*)
    uint64_t xmul;   // the variant's XXH3 avalanche multiplier (a kernel argument like the rest: no select in the hash loops)
};
// Hash variants (include/colorid_hip.h CID_HASH_*): which XXH3 the index was built with.
constexpr uint32_t kHashV08 = 0;   // published XXH3_64bits_withSeed (xxHash >= 0.8.0), pinned to known answers
constexpr uint32_t kHashV07 = 1;   // the XXH3 draft of xxHash v0.7.1 / v0.7.2 (candidate for crate xxh3 0.1.x; unverified, see below)
CID_FN uint32_t hash_variant_of(const ModMagic &mm) { return (mm.flags >> 8) & 0xFFu; }

CID_FN uint64_t mod_m(uint64_t h, const ModMagic &mm) {
    if (mm.flags & 1u) return h & (mm.m - 1);
    uint64_t q = umul64hi(h, mm.magic);
    if (mm.flags & 2u) {
        uint64_t t = ((h - q) >> 1) + q;
        q = t >> mm.shift;
    } else {
        q >>= mm.shift;
    }
    // bloom_size <= 2^32 and 2^32 itself takes the mask path, so m and the remainder fit 32 bits: the subtraction is exact
    // modulo 2^32 and one 32-bit multiply replaces the 64-bit one
    return (uint64_t)((uint32_t)h - (uint32_t)q * (uint32_t)mm.m);
}

// ---------------------------------------------------------------- XXH3-64 (published v0.8 spec), inputs <= 128 B

// First 128 bytes of the default secret as little-endian u64 words (every read the <=128-byte
// branches make is 8-byte aligned except the two u32 reads of the 1..3-byte branch = halves of word 0).
CID_DEVCONST constexpr uint64_t kSecretW[16] = {
    0xbe4ba423396cfeb8ULL, 0x1cad21f72c81017cULL, 0xdb979083e96dd4deULL, 0x1f67b3b7a4a44072ULL,
    0x78e5c0cc4ee679cbULL, 0x2172ffcc7dd05a82ULL, 0x8e2443f7744608b8ULL, 0x4c263a81e69035e0ULL,
    0xcb00c391bb52283cULL, 0xa32e531b8b65d088ULL, 0x4ef90da297486471ULL, 0xd8acdea946ef1938ULL,
    0x3f349ce33f76faa8ULL, 0x1d4f0bc7c7bbdcf9ULL, 0x3159b4cd4be0518aULL, 0x647378d9c97e9fc8ULL,
};

constexpr uint64_t P32_1 = 0x9E3779B1ULL;
constexpr uint64_t P64_1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t P64_2 = 0xC2B2AE3D27D4EB4FULL;
constexpr uint64_t P64_3 = 0x165667B19E3779F9ULL;
constexpr uint64_t PMX1 = 0x165667919E3779F9ULL;
constexpr uint64_t PMX2 = 0x9FB21C651E98DF25ULL;

CID_FN uint64_t mul128_fold64(uint64_t a, uint64_t b) { return (a * b) ^ umul64hi(a, b); }
// v0.8: h ^= h >> 37; h *= 0x165667919E3779F9; h ^= h >> 32.  The v0.7.1/v0.7.2 draft multiplies by PRIME64_3 instead.
CID_FN uint64_t xxh3_avalanche(uint64_t h, uint64_t mult = PMX1) {
    h ^= h >> 37; h *= mult; h ^= h >> 32;
    return h;
}
CID_FN uint64_t avalanche_mult_of(uint32_t hv) { return hv == kHashV07 ? P64_3 : PMX1; }
CID_FN uint64_t xxh64_avalanche(uint64_t h) {
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}
CID_FN uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
CID_FN uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

// Unaligned little-endian reads out of an LDS byte image through aligned dwords + v_alignbyte.
// The image must be readable 8 bytes past the last byte asked for (callers pad).
CID_FN uint32_t lds_rd32(const uint32_t *img, uint32_t off) {
    uint32_t w = off >> 2, sh = off & 3u;
    return alignbyte(img[w + 1], img[w], sh);
}
CID_FN uint64_t lds_rd64(const uint32_t *img, uint32_t off) {
    uint32_t w = off >> 2, sh = off & 3u;
    uint32_t a = img[w], b = img[w + 1], c = img[w + 2];
    uint32_t lo = alignbyte(b, a, sh);
    uint32_t hi = alignbyte(c, b, sh);
    return ((uint64_t)hi << 32) | lo;
}
CID_FN uint32_t lds_rd8(const uint32_t *img, uint32_t off) {
    return (img[off >> 2] >> (8u * (off & 3u))) & 0xffu;
}

// Input readers: where the k-mer's bytes come from.
struct LdsReader {  // bytes [off, off+len) of an LDS (or host) dword image
    const uint32_t *img;
    uint32_t off;
    CID_FN uint64_t rd64(uint32_t o) const { return lds_rd64(img, off + o); }
    CID_FN uint32_t rd32(uint32_t o) const { return lds_rd32(img, off + o); }
    CID_FN uint32_t rd8(uint32_t o) const { return lds_rd8(img, off + o); }
};

// 4 bases (2 bits each, base j at bits 2j+1:2j; A,C,G,T = 0..3) -> their 4 upper-case ASCII bytes, little-endian
CID_FN uint32_t ascii4(uint32_t x8) {
    uint32_t y = (x8 | (x8 << 12)) & 0x000F000Fu;
    y = (y | (y << 6)) & 0x03030303u;           // one code (0..3) per byte
#if defined(__HIPCC__) && !defined(CID_ASCII4_ARITH)
    return __builtin_amdgcn_perm(0u, 0x54474341u, y);   // v_perm_b32 as a 4-entry byte table: selector 0..3 picks 'A','C','G','T'
#else
    const uint32_t b0 = y & 0x01010101u, b1 = (y >> 1) & 0x01010101u;
    return 0x41414141u + 2u * y + 2u * b1 + 11u * (b0 & b1);  // 'A'+{0,2,6,19}: no byte ever carries
#endif
}

struct CodeReader {  // an upper-case ACGT k-mer (k <= 32) held as a 2-bit code, base j at bits 2j+1:2j
    uint64_t code;
    CID_FN uint32_t rd32(uint32_t o) const { return ascii4((uint32_t)(code >> (2u * o)) & 0xFFu); }
    CID_FN uint64_t rd64(uint32_t o) const {
        const uint32_t x = (uint32_t)(code >> (2u * o)) & 0xFFFFu;
        return ((uint64_t)ascii4(x >> 8) << 32) | ascii4(x & 0xFFu);
    }
    CID_FN uint32_t rd8(uint32_t o) const { return ascii4((uint32_t)(code >> (2u * o)) & 3u) & 0xFFu; }
};

// kHashV07 — XXH3 as drafted in xxHash v0.7.1 / v0.7.2 (Aug-Oct 2019), the code base a 2019 Rust port with the entry point
// `xxh3::hash64_with_seed` (crate xxh3 0.1.x, later twox-hash's xxh3 module) was written against.  Restated from memory of that
// source: NO known answer exists in this image, so this variant is a CANDIDATE for `colorid hashcheck`, not a parity claim.
//   17..128 bytes: exactly the v0.8 construction (seed folded into the secret words, len * PRIME64_1, mix16B pairs) with the
//                  avalanche multiplier PRIME64_3;
//   9..16: lo = in[0..8] ^ (secret[0..8] + seed), hi = in[len-8..] ^ (secret[8..16] - seed); avalanche(len + lo + hi + fold(lo*hi));
//   4..8 : keyed = (in32[0] | in32[len-4] << 32) ^ (secret[0..8] + seed); mix = len + (keyed ^ keyed >> 51) * PRIME32_1;
//          avalanche((mix ^ mix >> 47) * PRIME64_2);
//   1..3 : combined = c1 | c2 << 8 | c3 << 16 | len << 24; avalanche((combined ^ (secret32[0] + seed)) * PRIME64_1).
template <typename Reader, typename Emit>
CID_FN void xxh3_v07_short(const Reader &in, uint32_t len, uint32_t n, Emit &&emit) {
    if (len > 8) {
        const uint64_t i_lo = in.rd64(0), i_hi = in.rd64(len - 8);
        for (uint32_t s = 0; s < n; ++s) {
            const uint64_t lo = i_lo ^ (kSecretW[0] + s), hi = i_hi ^ (kSecretW[1] - s);
            emit(s, xxh3_avalanche((uint64_t)len + lo + hi + mul128_fold64(lo, hi), P64_3));
        }
    } else if (len >= 4) {
        const uint64_t in64 = (uint64_t)in.rd32(0) + ((uint64_t)in.rd32(len - 4) << 32);
        for (uint32_t s = 0; s < n; ++s) {
            const uint64_t keyed = in64 ^ (kSecretW[0] + s);
            const uint64_t mix = (uint64_t)len + (keyed ^ (keyed >> 51)) * P32_1;
            emit(s, xxh3_avalanche((mix ^ (mix >> 47)) * P64_2, P64_3));
        }
    } else {
        const uint32_t c1 = in.rd8(0), c2 = in.rd8(len >> 1), c3 = in.rd8(len - 1);
        const uint64_t combined = (uint64_t)(c1 | (c2 << 8) | (c3 << 16) | (len << 24));
        for (uint32_t s = 0; s < n; ++s) emit(s, xxh3_avalanche((combined ^ ((uint64_t)(uint32_t)kSecretW[0] + s)) * P64_1, P64_3));
    }
}

// which XXH3: the variant id (only consulted for inputs of <= 16 bytes) and its avalanche multiplier.  A kernel that is compiled
// for the published variant only passes the constants (HashSel::published()) and keeps two SGPRs.
struct HashSel {
    uint32_t hv;
    uint64_t xmul;
    CID_FN static HashSel of(const ModMagic &mm) { return HashSel{hash_variant_of(mm), mm.xmul}; }
    CID_FN static HashSel published() { return HashSel{kHashV08, PMX1}; }
};

// All n seeds (0..n-1) of one k-mer; emit(seed, hash).  len and the hash variant are wave-uniform.
template <typename Reader, typename Emit>
CID_FN void xxh3_seeds_from(const Reader &in, uint32_t len, uint32_t n, const HashSel hs, Emit &&emit) {
    if (len <= 16 && hs.hv == kHashV07) { xxh3_v07_short(in, len, n, emit); return; }
    const uint64_t xmul = hs.xmul;
    if (len > 16 && len <= 32) {  // the k = 21/27/31 case: 2 x mix16B, inputs read once for all seeds
        const uint64_t a0 = in.rd64(0), a1 = in.rd64(8);
        const uint64_t b0 = in.rd64(len - 16), b1 = in.rd64(len - 8);
        for (uint32_t s = 0; s < n; ++s) {
            uint64_t acc = (uint64_t)len * P64_1;
            acc += mul128_fold64(a0 ^ (kSecretW[0] + s), a1 ^ (kSecretW[1] - s));
            acc += mul128_fold64(b0 ^ (kSecretW[2] + s), b1 ^ (kSecretW[3] - s));
            emit(s, xxh3_avalanche(acc, xmul));
        }
    } else if (len > 32) {  // 33..128: (len-1)/32 + 1 front/back pairs
        const uint32_t nb = ((len - 1) >> 5) + 1;
        for (uint32_t s = 0; s < n; ++s) {
            uint64_t acc = (uint64_t)len * P64_1;
            for (uint32_t i = 0; i < nb; ++i) {
                const uint32_t f = 16 * i, b = len - 16 * (i + 1);
                acc += mul128_fold64(in.rd64(f) ^ (kSecretW[4 * i] + s), in.rd64(f + 8) ^ (kSecretW[4 * i + 1] - s));
                acc += mul128_fold64(in.rd64(b) ^ (kSecretW[4 * i + 2] + s), in.rd64(b + 8) ^ (kSecretW[4 * i + 3] - s));
            }
            emit(s, xxh3_avalanche(acc, xmul));
        }
    } else if (len > 8) {  // 9..16
        const uint64_t i_lo = in.rd64(0), i_hi = in.rd64(len - 8);
        for (uint32_t s = 0; s < n; ++s) {
            const uint64_t lo = i_lo ^ ((kSecretW[3] ^ kSecretW[4]) + s);
            const uint64_t hi = i_hi ^ ((kSecretW[5] ^ kSecretW[6]) - s);
            emit(s, xxh3_avalanche((uint64_t)len + bswap64(lo) + hi + mul128_fold64(lo, hi)));
        }
    } else if (len >= 4) {  // 4..8
        const uint64_t i1 = in.rd32(0), i2 = in.rd32(len - 4);
        const uint64_t in64 = i2 + (i1 << 32);
        for (uint32_t s = 0; s < n; ++s) {
            uint64_t seed = (uint64_t)s ^ ((uint64_t)__builtin_bswap32(s) << 32);
            uint64_t h = in64 ^ ((kSecretW[1] ^ kSecretW[2]) - seed);
            h ^= rotl64(h, 49) ^ rotl64(h, 24);
            h *= PMX2;
            h ^= (h >> 35) + len;
            h *= PMX2;
            emit(s, h ^ (h >> 28));
        }
    } else {  // 1..3
        const uint32_t c1 = in.rd8(0), c2 = in.rd8(len >> 1), c3 = in.rd8(len - 1);
        const uint32_t combined = (c1 << 16) | (c2 << 24) | c3 | (len << 8);
        const uint64_t flip = (uint64_t)((uint32_t)kSecretW[0] ^ (uint32_t)(kSecretW[0] >> 32));
        for (uint32_t s = 0; s < n; ++s) emit(s, xxh64_avalanche((uint64_t)combined ^ (flip + s)));
    }
}

template <typename Emit>
CID_FN void xxh3_seeds(const uint32_t *img, uint32_t off, uint32_t len, uint32_t n, const HashSel hs, Emit &&emit) {
    xxh3_seeds_from(LdsReader{img, off}, len, n, hs, emit);
}

// ---------------------------------------------------------------- 2-bit k-mer codes (upper-case ACGT, k <= 32)

// Reverse the order of the 2k-bit code's 2-bit fields (base j <-> base k-1-j), result right-aligned.
CID_FN uint64_t rev_fields(uint64_t code, uint32_t k) {
    uint64_t x = code;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = bswap64(x);
    return x >> (64u - 2u * k);
}
CID_FN uint64_t code_mask(uint32_t k) { return k >= 32 ? ~0ULL : ((1ULL << (2u * k)) - 1ULL); }

// A window's code read LSB-first (base j at bits 2j+1:2j) is `lsb`.  Returns the canonical k-mer
// (the byte-wise smaller of the window and its reverse complement; ties -> reverse complement, same string)
// in LSB-first form for hashing and in MSB-first form (`*msb`, base 0 most significant = lexicographic order).
CID_FN uint64_t canonical_code(uint64_t lsb, uint32_t k, uint64_t *msb) {
    const uint64_t mask = code_mask(k);
    const uint64_t f_msb = rev_fields(lsb, k);
    const uint64_t rc_msb = ~lsb & mask;            // complement, then the LSB-first reading IS the reversed order
    const uint64_t rc_lsb = ~f_msb & mask;
    const bool fwd = f_msb < rc_msb;                 // `l[i..i+k] < l_r[..]` (src/kmer.rs:104,231)
    *msb = fwd ? f_msb : rc_msb;
    return fwd ? lsb : rc_lsb;
}

// Minimizer of a canonical k-mer held as an MSB-first code (src/kmer.rs:971-986, find_minimizer): the smallest m-mer
// among the k-mer's m-mers at positions 0..k-m and the reverse complements of those at positions 1..k-m (position 0's
// reverse complement is never tried — the reference starts from `&seq[..m]`).  Returns an MSB-first m-field code.
CID_FN uint64_t minimizer_code(uint64_t msb, uint32_t k, uint32_t m) {
    const uint64_t mask = code_mask(m);
    uint64_t best = (msb >> (2u * (k - m))) & mask;
    for (uint32_t i = 1; i + m <= k; ++i) {
        const uint64_t f = (msb >> (2u * (k - m - i))) & mask;
        const uint64_t r = rev_fields(~f & mask, m);
        best = f < best ? f : best;
        best = r < best ? r : best;
    }
    return best;
}

}  // namespace cid
