// Window codes of sequences resident in HBM, shared by the k-mer set (cid_kmerset.hip) and the long-read read_id path
// (cid_readlong.hip): every k-window of a segment -> its canonical 2-bit code (or a sentinel), kmer.rs:221-243 / :87-125.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cid_kernels.hpp"
#include "cid_hash.hpp"

namespace cid {

struct Segment {  // consecutive windows of one sequence: window w starts at base_off + w*stride
    uint64_t base_off;  // offset of the first window's first base in `bases`
    uint64_t out_off;   // where the window codes go
    uint32_t n_win;     // (n_win-1)*stride + k <= kSegWindows + 31
    uint32_t stride;
};
constexpr uint32_t kSegWindows = 2048;

__device__ __forceinline__ uint8_t switch_base_dev(uint8_t c) {  // src/kmer.rs:847-863 for ACGTacgt
    const uint32_t low = c & 0x1Fu;
    return (uint8_t)(c ^ ((low == 1u || low == 0x14u) ? 0x15u : 0x04u));
}
__device__ __forceinline__ bool good_base_dev(uint32_t b) {
    const uint32_t u = b & 0xDFu;
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}
__device__ __forceinline__ uint64_t bits_at_dev(const uint32_t *w, uint32_t bit, uint32_t nbits) {
    const uint32_t i = bit >> 5, sh = bit & 31u;
    uint64_t v = (((uint64_t)w[i + 1] << 32) | w[i]) >> sh;
    if (sh) v |= (uint64_t)w[i + 2] << (64u - sh);
    return nbits >= 64 ? v : (v & ((1ull << nbits) - 1ull));
}

// mode 0: kmerize_vector (has_no_n filter; compare raw bytes, then upper-case; src/kmer.rs:104-117)
// mode 1: fastq body (has_no_n filter; raw case kept, so a lower-case base cannot be packed: flags[0] is raised)
// segs == NULL: segment sg is sequence sg of a batch of short sequences (reads: at most kSegWindows windows each) — its bases at
// seq_off[sg] (relative to `bases`' first byte, which is seq_off[0] of the batch), its codes at win_off[sg]; nothing per read comes
// from the host but the offsets it already has.
// KEYED (a set built for an index, cid_kmerset_set_target_index): every window's row0_key goes to key_out next to its code.
// redo != NULL (with segs): a lower-case base in segment sg also marks redo[seg_read[sg]].
struct KeyFor {   // the index the keys are for
    ModMagic mm;
    uint64_t scale;   // floor((2^32 - 1) * 2^32 / bloom_size)
};
// The sort key of a k-mer in such a set: monotone in the row its first hash (seed 0) selects, spread evenly over [0, 2^32 - 1) —
// never kNoKey, and distinct rows get distinct keys (bloom_size <= 2^32 - 1).  `lsb`: the canonical k-mer, LSB-first (as it is hashed).
__device__ __forceinline__ uint32_t row0_key(uint64_t lsb, uint32_t k, const KeyFor &kf) {
    uint32_t row0 = 0;
    xxh3_seeds_from(CodeReader{lsb}, k, 1, HashSel::of(kf.mm), [&](uint32_t, uint64_t h) { row0 = (uint32_t)mod_m(h, kf.mm); });
    return (uint32_t)(((uint64_t)row0 * kf.scale) >> 32);
}
__attribute__((unused)) static __global__ void k_row0_keys(const uint64_t *codes, uint32_t k, KeyFor kf, uint32_t *keys, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = row0_key(rev_fields(codes[i], k), k, kf);
}

template <bool KEYED>
__global__ __launch_bounds__(256) void k_extract_codes(const uint8_t *bases, const Segment *segs, uint32_t n_segs, uint32_t k,
                                                       int mode, uint64_t sentinel, uint64_t *out, int *flags,
                                                       const uint64_t *seq_off, const uint64_t *win_off, uint64_t base0, uint32_t *key_out, KeyFor kf,
                                                       const uint32_t *seg_read, uint8_t *redo) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t kBytes = kSegWindows + 32 + 96;                  // bases of one segment (+ slack)
    uint8_t *s_bases = smem + (size_t)wave * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
    uint32_t *s_pack = reinterpret_cast<uint32_t *>(s_bases + kBytes);
    uint32_t *s_bad = s_pack + (kBytes / 16 + 4);
    uint32_t *s_low = s_bad + (kBytes / 32 + 4);
    for (uint32_t sg = blockIdx.x * 4 + wave; sg < n_segs; sg += gridDim.x * 4) {
        Segment seg;
        if (segs) seg = segs[sg];
        else {
            const uint64_t b0 = seq_off[sg], len = seq_off[sg + 1] - b0;
            if (len < k) continue;   // (wave-uniform)
            seg = Segment{b0 - base0, win_off[sg], (uint32_t)(len - k + 1), 1u};
        }
        const uint32_t nb = (seg.n_win - 1) * seg.stride + k;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool lower = false;
        for (uint32_t i = lane; i < nb; i += 64) {
            const uint8_t b = bases[seg.base_off + i];
            s_bases[i] = b;
            lower = lower || (good_base_dev(b) && (b & 0x20u));
        }
        if (mode == 1 && __any(lower)) {
            if (lane == 0) {
                atomicOr(&flags[0], 1);
                if (redo) redo[seg_read[sg]] = 1;   // (the long-read path: the read of this segment alone takes the byte-string path)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j0 = 0; j0 * 16 < nb + 64; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint32_t code = 0, bad = 0, low = 0;
            for (uint32_t t = 0; t < 16; ++t) {
                const uint32_t i = j * 16 + t;
                const uint32_t b = i < nb ? s_bases[i] : 'N';
                const uint32_t c2 = (b >> 1) & 3u;
                code |= (c2 ^ (c2 >> 1)) << (2 * t);
                bad |= (good_base_dev(b) ? 0u : 1u) << t;
                low |= ((b >> 5) & 1u) << t;
            }
            const uint32_t bad_hi = __shfl_down(bad, 1, 64), low_hi = __shfl_down(low, 1, 64);
            if (j * 16 < nb + 64) {
                s_pack[j] = code;
                if (!(lane & 1)) { s_bad[j >> 1] = bad | (bad_hi << 16); s_low[j >> 1] = low | (low_hi << 16); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t mask = code_mask(k);
        for (uint32_t wi = lane; wi < seg.n_win; wi += 64) {
            const uint32_t w = wi * seg.stride;              // position of the window in the staged bases
            uint64_t result = sentinel;
            if (bits_at_dev(s_bad, w, k) == 0) {
                const uint64_t lsb = bits_at_dev(s_pack, 2 * w, 2 * k);
                const uint64_t lowbits = bits_at_dev(s_low, w, k);
                uint64_t msb;
                if (lowbits == 0 || lowbits == (k >= 64 ? ~0ull : ((1ull << k) - 1ull))) {
                    canonical_code(lsb, k, &msb);            // uniform case: byte order == code order
                } else {                                     // mixed case: the reference compares the raw bytes
                    bool fwd = false;                        // equal strings take the reverse-complement branch
                    for (uint32_t t = 0; t < k; ++t) {
                        const uint8_t f = s_bases[w + t], r = switch_base_dev(s_bases[w + k - 1 - t]);
                        if (f != r) { fwd = f < r; break; }
                    }
                    const uint64_t f_msb = rev_fields(lsb, k);
                    msb = fwd ? f_msb : (~lsb & mask);
                }
                result = msb;
            }
            out[seg.out_off + wi] = result;
            if constexpr (KEYED) key_out[seg.out_off + wi] = result == sentinel ? kNoKey : row0_key(rev_fields(result, k), k, kf);
        }
    }
}

__device__ __forceinline__ uint32_t read_of_window(const uint64_t *wstart, uint32_t n_reads, uint64_t w) {
    uint32_t lo = 0, hi = n_reads;  // largest r with wstart[r] <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (wstart[mid] <= w) lo = mid; else hi = mid;
    }
    return lo;
}
__attribute__((unused)) static __global__ void k_codes_to_minimizers(uint64_t *codes, uint64_t n, uint32_t k, uint32_t m, uint64_t sentinel_k, uint64_t sentinel_m) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = codes[i];
    codes[i] = c == sentinel_k ? sentinel_m : minimizer_code(c, k, m);
}

}  // namespace cid
