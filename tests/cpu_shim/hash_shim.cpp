// CPU unit-test harness (g++) for the integer arithmetic the gfx950 kernels run: includes the SAME
// colorid_amd/csrc/cid_hash.hpp + cid_host_math.hpp the HIP build compiles, with the two HIP builtins
// replaced by their plain-C definitions.  Test infrastructure only; never linked into the product.
#include <cstring>
#include <vector>

#include "../../colorid_amd/csrc/cid_hash.hpp"
#include "../../colorid_amd/csrc/cid_host_math.hpp"

extern "C" {

// Hash `kmer` (len bytes) with seeds 0..n-1 exactly as a lane does: out of a dword image at byte offset `off`.
static cid::ModMagic g_hv{1, 0, 0, 1, cid::PMX1};   // carries the hash variant (cid::kHashV08 / kHashV07) used by every shim call
void shim_set_hash_variant(uint32_t hv) { g_hv.flags = 1u | (hv << 8); g_hv.xmul = cid::avalanche_mult_of(hv); }
void shim_hash_seeds(const uint8_t *kmer, uint32_t len, uint32_t off, uint32_t n, uint64_t *out) {
    std::vector<uint32_t> img((off + len + 16 + 3) / 4 + 4, 0xA5A5A5A5u);
    memcpy(reinterpret_cast<uint8_t *>(img.data()) + off, kmer, len);
    cid::xxh3_seeds(img.data(), off, len, n, cid::HashSel::of(g_hv), [&](uint32_t s, uint64_t h) { out[s] = h; });
}

// The packed path: an upper-case ACGT k-mer (k <= 32) -> LSB-first 2-bit code -> canonical code -> hashes of the
// canonical string.  out_canon receives the canonical ASCII string as the device would hash it.
void shim_hash_canonical_code(const uint8_t *kmer, uint32_t k, uint32_t n, uint64_t *out, uint8_t *out_canon, uint64_t *out_msb) {
    uint64_t lsb = 0;
    for (uint32_t j = 0; j < k; ++j) {
        const uint64_t c = kmer[j] == 'A' ? 0 : kmer[j] == 'C' ? 1 : kmer[j] == 'G' ? 2 : 3;
        lsb |= c << (2 * j);
    }
    uint64_t msb;
    const uint64_t canon = cid::canonical_code(lsb, k, &msb);
    *out_msb = msb;
    const cid::CodeReader r{canon};
    for (uint32_t j = 0; j < k; ++j) out_canon[j] = (uint8_t)r.rd8(j);
    cid::xxh3_seeds_from(r, k, n, cid::HashSel::of(g_hv), [&](uint32_t s, uint64_t h) { out[s] = h; });
}

// canonical ACGT k-mer (ASCII) -> minimizer as the device computes it from the 2-bit code -> ASCII, and its hashes (len m)
void shim_minimizer(const uint8_t *kmer, uint32_t k, uint32_t m, uint32_t n, uint8_t *out_mini, uint64_t *out_hash) {
    uint64_t msb = 0;
    for (uint32_t j = 0; j < k; ++j) msb = (msb << 2) | (uint64_t)(kmer[j] == 'A' ? 0 : kmer[j] == 'C' ? 1 : kmer[j] == 'G' ? 2 : 3);
    const uint64_t mini = cid::minimizer_code(msb, k, m);
    const cid::CodeReader r{cid::rev_fields(mini, m)};
    for (uint32_t j = 0; j < m; ++j) out_mini[j] = (uint8_t)r.rd8(j);
    cid::xxh3_seeds_from(r, m, n, cid::HashSel::of(g_hv), [&](uint32_t s, uint64_t h) { out_hash[s] = h; });
}

uint64_t shim_mod(uint64_t h, uint64_t m) {
    const cid::ModMagicHost mh = cid::make_mod_magic(m);
    const cid::ModMagic mm{mh.m, mh.magic, mh.shift, mh.flags, cid::PMX1};
    return cid::mod_m(h, mm);
}

uint32_t shim_row_stride_words(uint32_t n_colors) { return cid::row_stride_words(n_colors); }
}
