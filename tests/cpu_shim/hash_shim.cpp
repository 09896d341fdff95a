// CPU unit-test harness (g++) for the integer arithmetic the gfx950 kernels run: includes the SAME
// colorid_amd/csrc/cid_hash.hpp + cid_host_math.hpp the HIP build compiles, with the two HIP builtins
// replaced by their plain-C definitions.  Test infrastructure only; never linked into the product.
#include <cstring>
#include <vector>

#include "../../colorid_amd/csrc/cid_hash.hpp"
#include "../../colorid_amd/csrc/cid_host_math.hpp"

extern "C" {

// Hash `kmer` (len bytes) with seeds 0..n-1 exactly as a lane does: out of a dword image at byte offset `off`.
void shim_hash_seeds(const uint8_t *kmer, uint32_t len, uint32_t off, uint32_t n, uint64_t *out) {
    std::vector<uint32_t> img((off + len + 16 + 3) / 4 + 4, 0xA5A5A5A5u);
    memcpy(reinterpret_cast<uint8_t *>(img.data()) + off, kmer, len);
    cid::xxh3_seeds(img.data(), off, len, n, [&](uint32_t s, uint64_t h) { out[s] = h; });
}

uint64_t shim_mod(uint64_t h, uint64_t m) {
    const cid::ModMagicHost mh = cid::make_mod_magic(m);
    const cid::ModMagic mm{mh.m, mh.magic, mh.shift, mh.flags};
    return cid::mod_m(h, mm);
}

uint32_t shim_row_stride_words(uint32_t n_colors) { return cid::row_stride_words(n_colors); }
}
