// Host-side integer helpers (no HIP types): the exact-modulo constants and the row-stride rule.
#pragma once
#include <stdint.h>

namespace cid {

struct ModMagicHost {
    uint64_t m, magic;
    uint32_t shift, flags;  // flags: 1 = power of two, 2 = add step (same meaning as cid::ModMagic)
};

// Round-up multiply-shift constants for an exact unsigned 64-bit `x % m` (Granlund–Montgomery):
//   q = mulhi64(x, magic); if (add) q = (((x - q) >> 1) + q) >> shift; else q >>= shift;  r = x - q*m.
inline ModMagicHost make_mod_magic(uint64_t m) {
    ModMagicHost mm{m, 0, 0, 0};
    if ((m & (m - 1)) == 0) { mm.flags = 1; return mm; }  // includes m == 1
    const uint32_t fl = 63u - (uint32_t)__builtin_clzll(m);
    const unsigned __int128 num = (unsigned __int128)1 << (64 + fl);
    uint64_t prop = (uint64_t)(num / m);
    const uint64_t rem = (uint64_t)(num % m);
    const uint64_t e = m - rem;
    mm.shift = fl;
    if (e >= ((uint64_t)1 << fl)) {
        prop += prop;
        const uint64_t twice = rem + rem;
        if (twice >= m || twice < rem) prop += 1;
        mm.flags = 2;
    }
    mm.magic = prop + 1;
    return mm;
}

// u64 words per matrix row: 1 for <= 64 colours, else the next power of two >= ceil(C/64) (2..128 words); beyond
// 8192 colours ("wide" rows) a multiple of 128 words = whole KiB, covered by one wave in rs/128 steps.
inline uint32_t row_stride_words(uint32_t n_colors) {
    const uint32_t w64 = (n_colors + 63u) / 64u;
    if (w64 <= 1) return 1;
    if (w64 > 128) return (w64 + 127u) / 128u * 128u;
    uint32_t rs = 2;
    while (rs < w64) rs <<= 1;
    return rs;
}

}  // namespace cid
