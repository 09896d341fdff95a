/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the hash the reference calls at
 *   src/simple_bloom.rs:22-23,30-31   src/perfect_search.rs:28-29,86-87
 *   src/batch_search_pe.rs:48-49,128-129   src/read_id_mt_pe.rs:78-79,119-120,143-144
 * i.e.  xxh3::hash64_with_seed(bytes, seed)  from the crates.io crate `xxh3 ^0.1.1`
 * (Cargo.toml:9; no Cargo.lock in the reference tree, so the resolved version is not pinned).
 * The crate is NOT vendored under /root/reference, so the algorithm restated here is the
 * PUBLISHED one: XXH3_64bits_withSeed of xxHash v0.8.x (the frozen XXH3 spec).
 *
 * PARITY STATUS: pinned to the published XXH3 known answers (python-xxhash 3.8.1 /
 * libxxhash 0.8.x: tests/golden/xxh3_kat.json).  PARITY UNPINNED against the 2019 crate
 * xxh3 0.1.x itself (it may implement a pre-0.8 draft of XXH3; the reference has no test
 * that asserts a hash value — src/simple_bloom.rs:45-67 checks insert/contains agreement only).
 */
#include <stdint.h>
#include <string.h>
#include <stddef.h>

static const uint8_t kSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

#define P32_1 0x9E3779B1U
#define P32_2 0x85EBCA77U
#define P32_3 0xC2B2AE3DU
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
#define PMX1 0x165667919E3779F9ULL
#define PMX2 0x9FB21C651E98DF25ULL

static uint64_t rd64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
static uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint32_t swap32(uint32_t x) {
    return (x >> 24) | ((x >> 8) & 0xff00U) | ((x << 8) & 0xff0000U) | (x << 24);
}
static uint64_t swap64(uint64_t x) {
    return ((uint64_t)swap32((uint32_t)x) << 32) | swap32((uint32_t)(x >> 32));
}
static uint64_t mul128_fold64(uint64_t a, uint64_t b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
static uint64_t xxh64_avalanche(uint64_t h) {
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}
static uint64_t xxh3_avalanche(uint64_t h) {
    h ^= h >> 37; h *= PMX1; h ^= h >> 32;
    return h;
}
static uint64_t rrmxmx(uint64_t h, uint64_t len) {
    h ^= rotl64(h, 49) ^ rotl64(h, 24);
    h *= PMX2;
    h ^= (h >> 35) + len;
    h *= PMX2;
    return h ^ (h >> 28);
}
static uint64_t mix16(const uint8_t *in, const uint8_t *sec, uint64_t seed) {
    return mul128_fold64(rd64(in) ^ (rd64(sec) + seed), rd64(in + 8) ^ (rd64(sec + 8) - seed));
}

static void accumulate_512(uint64_t acc[8], const uint8_t *in, const uint8_t *sec) {
    for (int i = 0; i < 8; ++i) {
        uint64_t dv = rd64(in + 8 * i);
        uint64_t dk = dv ^ rd64(sec + 8 * i);
        acc[i ^ 1] += dv;
        acc[i] += (uint64_t)(uint32_t)dk * (dk >> 32);
    }
}
static void scramble(uint64_t acc[8], const uint8_t *sec) {
    for (int i = 0; i < 8; ++i) {
        uint64_t a = acc[i];
        a ^= a >> 47;
        a ^= rd64(sec + 8 * i);
        a *= P32_1;
        acc[i] = a;
    }
}

static uint64_t hash_long(const uint8_t *in, size_t len, uint64_t seed) {
    uint8_t secret[192];
    if (seed == 0) {
        memcpy(secret, kSecret, 192);
    } else {
        for (int i = 0; i < 12; ++i) {
            uint64_t lo = rd64(kSecret + 16 * i) + seed;
            uint64_t hi = rd64(kSecret + 16 * i + 8) - seed;
            for (int b = 0; b < 8; ++b) {
                secret[16 * i + b] = (uint8_t)(lo >> (8 * b));
                secret[16 * i + 8 + b] = (uint8_t)(hi >> (8 * b));
            }
        }
    }
    uint64_t acc[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    const size_t nb_stripes_per_block = (192 - 64) / 8;
    const size_t block_len = 64 * nb_stripes_per_block;
    const size_t nb_blocks = (len - 1) / block_len;
    for (size_t b = 0; b < nb_blocks; ++b) {
        for (size_t s = 0; s < nb_stripes_per_block; ++s)
            accumulate_512(acc, in + b * block_len + s * 64, secret + s * 8);
        scramble(acc, secret + 192 - 64);
    }
    const size_t nb_stripes = ((len - 1) - block_len * nb_blocks) / 64;
    for (size_t s = 0; s < nb_stripes; ++s)
        accumulate_512(acc, in + nb_blocks * block_len + s * 64, secret + s * 8);
    accumulate_512(acc, in + len - 64, secret + 192 - 64 - 7);
    uint64_t r = (uint64_t)len * P64_1;
    for (int i = 0; i < 4; ++i)
        r += mul128_fold64(acc[2 * i] ^ rd64(secret + 11 + 16 * i),
                           acc[2 * i + 1] ^ rd64(secret + 11 + 16 * i + 8));
    return xxh3_avalanche(r);
}

/* ---- candidate variant 1: the XXH3 draft of xxHash v0.7.1 / v0.7.2 (Aug-Oct 2019) — what a 2019 Rust port exposing
 * `xxh3::hash64_with_seed` (crate xxh3 0.1.x; the same code later became twox-hash's xxh3 module) would have been written
 * against.  Restated from memory of xxh3.h at those tags; NO known-answer vector for it exists in this image, so it is a
 * CANDIDATE that `colorid hashcheck` can confirm or reject against a reference-built .bxi — parity unpinned.
 * Differences from the frozen v0.8 for inputs up to 128 bytes: XXH3_avalanche multiplies by PRIME64_3; the 1..16-byte paths
 * key the input with secret[0..16] +/- seed and have no bswap / rrmxmx steps.  (129..240 and long inputs: not restated —
 * k-mers are at most 128 bytes here.) */
static uint64_t v07_avalanche(uint64_t h) {
    h ^= h >> 37; h *= P64_3; h ^= h >> 32;
    return h;
}
static uint64_t v07_mix16(const uint8_t *in, const uint8_t *key, uint64_t seed) {
    uint64_t ll1 = rd64(in), ll2 = rd64(in + 8);
    return mul128_fold64(ll1 ^ (rd64(key) + seed), ll2 ^ (rd64(key + 8) - seed));
}
uint64_t orc_xxh3_v07_64_with_seed(const uint8_t *in, size_t len, uint64_t seed) {
    const uint8_t *key = kSecret;
    if (len <= 16) {
        if (len > 8) {                                             /* XXH3_len_9to16_64b */
            uint64_t ll1 = rd64(in) ^ (rd64(key) + seed);
            uint64_t ll2 = rd64(in + len - 8) ^ (rd64(key + 8) - seed);
            return v07_avalanche((uint64_t)len + (ll1 + ll2) + mul128_fold64(ll1, ll2));
        }
        if (len >= 4) {                                            /* XXH3_len_4to8_64b */
            uint32_t in1 = rd32(in), in2 = rd32(in + len - 4);
            uint64_t in64 = (uint64_t)in1 + ((uint64_t)in2 << 32);
            uint64_t keyed = in64 ^ (rd64(key) + seed);
            uint64_t mix64 = (uint64_t)len + ((keyed ^ (keyed >> 51)) * (uint64_t)P32_1);
            return v07_avalanche((mix64 ^ (mix64 >> 47)) * P64_2);
        }
        if (len > 0) {                                             /* XXH3_len_1to3_64b */
            uint8_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
            uint32_t combined = (uint32_t)c1 + ((uint32_t)c2 << 8) + ((uint32_t)c3 << 16) + ((uint32_t)len << 24);
            uint64_t keyed = (uint64_t)combined ^ ((uint64_t)rd32(key) + seed);
            return v07_avalanche(keyed * P64_1);
        }
        return 0;
    }
    if (len <= 128) {                                              /* XXH3_len_17to128_64b, nested form */
        uint64_t acc = (uint64_t)len * P64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) {
                    acc += v07_mix16(in + 48, key + 96, seed);
                    acc += v07_mix16(in + len - 64, key + 112, seed);
                }
                acc += v07_mix16(in + 32, key + 64, seed);
                acc += v07_mix16(in + len - 48, key + 80, seed);
            }
            acc += v07_mix16(in + 16, key + 32, seed);
            acc += v07_mix16(in + len - 32, key + 48, seed);
        }
        acc += v07_mix16(in, key, seed);
        acc += v07_mix16(in + len - 16, key + 16, seed);
        return v07_avalanche(acc);
    }
    return 0;   /* not restated beyond 128 bytes */
}

/* which variant orc_colorid.c's bit_index() uses: 0 = published v0.8 (default), 1 = the v0.7 draft above.  Process-wide test switch. */
static int g_variant = 0;
void orc_set_hash_variant(int v) { g_variant = v; }
int orc_get_hash_variant(void) { return g_variant; }
uint64_t orc_xxh3_published_64_with_seed(const uint8_t *in, size_t len, uint64_t seed);
uint64_t orc_xxh3_64_with_seed(const uint8_t *in, size_t len, uint64_t seed) {
    return g_variant == 1 ? orc_xxh3_v07_64_with_seed(in, len, seed) : orc_xxh3_published_64_with_seed(in, len, seed);
}

/* XXH3_64bits_withSeed, all input lengths. */
uint64_t orc_xxh3_published_64_with_seed(const uint8_t *in, size_t len, uint64_t seed) {
    const uint8_t *sec = kSecret;
    if (len == 0) return xxh64_avalanche(seed ^ (rd64(sec + 56) ^ rd64(sec + 64)));
    if (len <= 3) {
        uint32_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        uint32_t combined = (c1 << 16) | (c2 << 24) | c3 | ((uint32_t)len << 8);
        uint64_t bitflip = (uint64_t)(rd32(sec) ^ rd32(sec + 4)) + seed;
        return xxh64_avalanche((uint64_t)combined ^ bitflip);
    }
    if (len <= 8) {
        seed ^= (uint64_t)swap32((uint32_t)seed) << 32;
        uint32_t i1 = rd32(in), i2 = rd32(in + len - 4);
        uint64_t bitflip = (rd64(sec + 8) ^ rd64(sec + 16)) - seed;
        uint64_t in64 = (uint64_t)i2 + ((uint64_t)i1 << 32);
        return rrmxmx(in64 ^ bitflip, len);
    }
    if (len <= 16) {
        uint64_t bf1 = (rd64(sec + 24) ^ rd64(sec + 32)) + seed;
        uint64_t bf2 = (rd64(sec + 40) ^ rd64(sec + 48)) - seed;
        uint64_t lo = rd64(in) ^ bf1;
        uint64_t hi = rd64(in + len - 8) ^ bf2;
        uint64_t acc = len + swap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len <= 128) {
        uint64_t acc = (uint64_t)len * P64_1;
        /* pairs (front block i, back block i), i = 0..(len-1)/32 — the published nested-if
         * form adds the same terms; wrapping addition commutes. */
        size_t nb = ((len - 1) >> 5) + 1;
        for (size_t i = 0; i < nb; ++i) {
            acc += mix16(in + 16 * i, sec + 32 * i, seed);
            acc += mix16(in + len - 16 * (i + 1), sec + 32 * i + 16, seed);
        }
        return xxh3_avalanche(acc);
    }
    if (len <= 240) {
        uint64_t acc = (uint64_t)len * P64_1;
        size_t nb_rounds = len / 16;
        for (size_t i = 0; i < 8; ++i) acc += mix16(in + 16 * i, sec + 16 * i, seed);
        acc = xxh3_avalanche(acc);
        for (size_t i = 8; i < nb_rounds; ++i) acc += mix16(in + 16 * i, sec + 16 * (i - 8) + 3, seed);
        acc += mix16(in + len - 16, sec + 136 - 17, seed);
        return xxh3_avalanche(acc);
    }
    return hash_long(in, len, seed);
}
