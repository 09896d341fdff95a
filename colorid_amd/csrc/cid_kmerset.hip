// GPU k-mer counting for `search`: replaces the reference's String-keyed FnvHashMap counting
// (src/kmer.rs:87-125 kmerize_vector, :461-510 / :581-655 the fastq bodies, :826-837 clean_map) for k <= 32:
//   windows -> 2-bit canonical codes (one u64 each) -> radix sort -> run-length = (distinct k-mer, multiplicity).
// The set stays in HBM and feeds k_search_count / k_search_perfect directly (8 bytes per k-mer instead of k).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"
#include "cid_partition.hpp"
#include "cid_windows.hpp"
#include "cid_devbuf.hpp"

namespace cid {

// windows of every sequence (0 when it is shorter than k): the input of the scan that places each read's codes
__global__ void k_seq_windows(const uint64_t *seq_off, uint64_t n_seqs, uint32_t k, uint64_t *n_win) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t len = seq_off[s + 1] - seq_off[s];
    n_win[s] = len >= k ? len - k + 1 : 0;
}
__global__ void k_fill_u32(uint32_t *p, uint32_t v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_flag_saturated(const uint32_t *counts, const uint64_t *n, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *n && counts[i] == 0xFFFFFFFFu) atomicOr(flag, 1);
}
__global__ void k_flag_gt(const uint32_t *counts, uint64_t t, uint8_t *flags, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = counts[i] > t ? 1 : 0;
}
__global__ void k_codes_to_ascii(const uint64_t *codes, uint32_t k, uint8_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = codes[i];
    for (uint32_t t = 0; t < k; ++t) out[i * k + t] = (uint8_t)"ACGT"[(c >> (2 * (k - 1 - t))) & 3u];
}
// sort key for index locality: the 128-byte line of the k-mer's first row (bucket_bits == 0), or the slice of the index it falls
// in when the index is cut into 2^bucket_bits equal slices
__global__ void k_row0_line(const uint64_t *codes, uint32_t k, ModMagic mm, uint32_t line_shift, uint32_t bucket_bits, uint32_t *keys, uint32_t *idx,
                            uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t lsb = rev_fields(codes[i], k);
    uint32_t row0 = 0;
    xxh3_seeds_from(CodeReader{lsb}, k, 1, HashSel::of(mm), [&](uint32_t, uint64_t h) { row0 = (uint32_t)mod_m(h, mm); });
    keys[i] = bucket_bits ? (uint32_t)(((uint64_t)row0 << bucket_bits) / mm.m) : row0 >> line_shift;
    if (idx) idx[i] = (uint32_t)i;
}
// the same key for the k-mers of a byte-string set (k > 32): k ASCII bytes each, hashed as they are
struct BytesReader {
    const uint8_t *b;
    __device__ __forceinline__ uint32_t rd8(uint32_t o) const { return b[o]; }
    __device__ __forceinline__ uint32_t rd32(uint32_t o) const { return rd8(o) | (rd8(o + 1) << 8) | (rd8(o + 2) << 16) | (rd8(o + 3) << 24); }
    __device__ __forceinline__ uint64_t rd64(uint32_t o) const { return (uint64_t)rd32(o) | ((uint64_t)rd32(o + 4) << 32); }
};
__global__ void k_row0_line_ascii(const uint8_t *ascii, uint32_t k, ModMagic mm, uint32_t line_shift, uint32_t bucket_bits, uint32_t *keys, uint32_t *idx,
                                  uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t row0 = 0;
    xxh3_seeds_from(BytesReader{ascii + i * k}, k, 1, HashSel::of(mm), [&](uint32_t, uint64_t h) { row0 = (uint32_t)mod_m(h, mm); });
    keys[i] = bucket_bits ? (uint32_t)(((uint64_t)row0 << bucket_bits) / mm.m) : row0 >> line_shift;
    idx[i] = (uint32_t)i;
}
__global__ void k_permute_rows(const uint8_t *rows_in, const uint32_t *counts_in, const uint32_t *idx, uint32_t k, uint8_t *rows_out, uint32_t *counts_out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = idx[i];
    for (uint32_t t = 0; t < k; ++t) rows_out[i * k + t] = rows_in[j * k + t];
    counts_out[i] = counts_in[j];
}

}  // namespace cid

struct cid_kmerset {
    cid_ctx *ctx = nullptr;
    uint32_t k = 0;
    uint64_t sentinel = 0;
    unsigned end_bit = 64;
    uint64_t *raw = nullptr;  size_t n_raw = 0, cap_raw = 0;   // window codes not yet merged
    uint64_t *codes = nullptr; uint32_t *counts = nullptr; size_t n = 0;  // distinct k-mers, ascending code unless reordered
    int *d_flags = nullptr;
    bool finalized = false;
    // built FOR an index (cid_kmerset_set_target_index): every window carries row0_key of that index (raw_key, parallel to raw) and the
    // set comes out ordered by (row0_key, code) — the order in which the search's first-row fetches share 128-byte lines
    bool targeted = false;
    cid::KeyFor key_for{};
    uint32_t *raw_key = nullptr;
    // merge the unsorted window buffer into the set beyond this many codes (2 GiB).  (8 GiB until round 3: the buffer then regrows through
    // 1.3 / 1.9 / 2.9 / 4.3 / 6.5 / 9.7 GB blocks, and those hipMallocs made a 16 M-read query's count take 0.25 s or 1.9 s from run to run)
    size_t compact_at = 1ull << 28;
    // k > 32: keys are byte strings.  The sequences stay resident until finalize (g_bases, g_segs), where every window's key is
    // described as a stretch of them, sorted on a 4-bit-per-base image (LSD radix, 16 bases per pass) and run-length counted; the
    // finished set is n x k ASCII bytes (`ascii`) + counts, and feeds the byte-string kernels.
    bool general = false;
    int g_mode = -1;
    uint8_t *g_bases = nullptr; size_t g_n = 0, g_cap = 0;
    std::vector<cid::Segment> g_segs;
    uint64_t g_windows = 0;
    uint8_t *ascii = nullptr;
};

namespace cid {   // (for the FASTQ front end, cid_fastq.hip)
uint32_t kmerset_k(const cid_kmerset *ks) { return ks->k; }
cid_ctx *kmerset_ctx(const cid_kmerset *ks) { return ks->ctx; }
}


namespace cid {
int kmerset_view(const cid_kmerset *ks, cid_ctx **ctx, const uint64_t **codes, const uint32_t **counts, uint64_t *n, uint32_t *k) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->general) return fail(CID_ERR_UNSUPPORTED, "a k_size > 32 set holds byte strings: shard it with the host-pointer group calls");
    *ctx = ks->ctx; *codes = ks->codes; *counts = ks->counts; *n = ks->n; *k = ks->k;
    return CID_OK;
}
}  // namespace cid

namespace {


// ---- the k-mer set's sort (cid_partition.hpp): MSD radix partition passes until the runs fit a workgroup's LDS, then every run is
// finished on its CU.  `a` holds the n codes (sentinels = keys with a bit at or above `top` included) and is overwritten; `b` is
// scratch of the same size.  On return *sorted (a or b) holds the *n_real real codes in ascending order — the sentinels are dropped,
// not carried.  CID_ERR_UNSUPPORTED = not this path's case (small input, wide codes): the caller sorts with rocPRIM's LSD radix sort.
static const bool kMsdSort = getenv("CID_KMERSET_MSD_SORT") ? atoi(getenv("CID_KMERSET_MSD_SORT")) != 0 : true;
int msd_sort(cid_ctx *c, hipStream_t st, uint64_t *a, uint64_t *b, size_t n, unsigned top, uint64_t **sorted, size_t *n_real) {
    using namespace cid;
    // (CID_KMERSET_MSD_MIN: the tests send small inputs through these kernels too; below a million keys the launches cost more than they save)
    const char *min_env = getenv("CID_KMERSET_MSD_MIN");
    const size_t min_n = min_env ? strtoull(min_env, nullptr, 10) : (size_t)1 << 20;
    if (!kMsdSort || top >= 64 || top < 12 || n < min_n || n < 2 || n >= (1ull << 32)) return CID_ERR_UNSUPPORTED;
    // the prefix the partition passes consume: enough bits for runs of 1300 .. 2600 keys (a workgroup sorts up to 4096 in 32 KiB of LDS)
    unsigned prefix = 1;
    while (prefix < top - 8 && ((uint64_t)n >> prefix) > 2600) ++prefix;
    const unsigned levels = (prefix + kPartBits - 1) / kPartBits;
    if (levels > 3) return CID_ERR_UNSUPPORTED;
    unsigned lbits[3] = {0, 0, 0};
    for (unsigned l = 0; l < levels; ++l) lbits[l] = l + 1 < levels ? kPartBits : prefix - kPartBits * (levels - 1);
    const uint32_t n_runs = 1u << prefix;
    uint32_t S_last = 1;                                   // segments entering the last partition level
    for (unsigned l = 0; l + 1 < levels; ++l) S_last <<= lbits[l];
    const uint32_t max_tiles = part_max_tiles((uint32_t)n, S_last);
    constexpr uint32_t kInfo = 8, kBigCap = 1024;          // info: [0] dropped (sentinels) [1] big runs [2] largest run [3] hard runs
    DevBuf<uint32_t> seg_a(c), seg_b(c), tile_base(c), table(c), info(c), hard(c);
    DevBuf<uint8_t> scan_tmp(c);
    int rc;
    if ((rc = seg_a.alloc((size_t)n_runs + 1)) || (rc = seg_b.alloc((size_t)n_runs + 1)) || (rc = tile_base.alloc((size_t)S_last + 1)) ||
        (rc = table.alloc((size_t)max_tiles * kPartBins)) || (rc = info.alloc(kInfo + kBigCap)) || (rc = hard.alloc((size_t)n_runs + 1)))
        return rc;
    size_t scan_tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tb, table.p, table.p, 0u, (size_t)max_tiles * kPartBins, rocprim::plus<uint32_t>(), st));
    if ((rc = scan_tmp.alloc(scan_tb))) return rc;
    HIP_TRY(hipMemsetAsync(info.p, 0, kInfo * 4, st));
    const uint32_t seg0[2] = {0u, (uint32_t)n};
    HIP_TRY(hipMemcpyAsync(seg_a.p, seg0, 8, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)cid::ctx_n_cu(c) * 8u;
    uint64_t *src = a, *dst = b;
    uint32_t *seg = seg_a.p, *seg_next = seg_b.p;
    uint32_t S = 1;
    unsigned consumed = 0;
    for (unsigned l = 0; l < levels; ++l) {
        const uint32_t bits = lbits[l], shift = top - consumed - bits, bins = 1u << bits;
        const uint32_t level_top = l == 0 ? top : 64u;    // the first level leaves the sentinels behind
        const size_t table_n = (size_t)part_max_tiles((uint32_t)n, S) * bins;
        HIP_TRY(hipMemsetAsync(table.p, 0, table_n * 4, st));
        hipLaunchKernelGGL(k_part_tiles, dim3(1), dim3(kPartBlock), 0, st, seg, S, tile_base.p);
        hipLaunchKernelGGL(k_part_hist, dim3(grid), dim3(kPartBlock), 0, st, src, seg, tile_base.p, S, shift, bits, level_top, table.p, info.p);
        HIP_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_tb, table.p, table.p, 0u, table_n, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, tile_base.p, S, shift, bits, level_top, table.p);
        hipLaunchKernelGGL(k_part_offsets, dim3((S * bins + 256) / 256), dim3(256), 0, st, table.p, tile_base.p, S, bits, (uint32_t)n, info.p, seg_next);
        HIP_TRY(hipGetLastError());
        std::swap(src, dst);
        std::swap(seg, seg_next);
        S *= bins;
        consumed += bits;
    }
    // src: partitioned, seg[0 .. n_runs]: the runs.  How large are they?
    hipLaunchKernelGGL(k_run_sizes, dim3((n_runs + 255) / 256), dim3(256), 0, st, seg, n_runs, 8192u, info.p + 1, info.p + kInfo, kBigCap, info.p + 2);
    uint32_t h_info[kInfo + kBigCap];
    HIP_TRY(hipMemcpyAsync(h_info, info.p, sizeof(h_info), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t dropped = h_info[0], n_big = h_info[1], largest = h_info[2];
    const size_t kept = n - dropped;
    const unsigned rest = top - consumed;                 // bits the runs still have to be sorted on
    if (n_big > kBigCap) {   // badly skewed codes (low-complexity sequence): LSD radix sort of the partitioned array, all bits
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, src, dst, kept, 0u, top, st));
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::radix_sort_keys(tmp.p, tb, src, dst, kept, 0u, top, st));
        *sorted = dst; *n_real = kept;
        return CID_OK;
    }
    // the bucket kernel takes every run it can (evenly spread keys), names the others in `hard`; the radix kernel sorts those
    hipLaunchKernelGGL(k_run_bucket_sort<8>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, info.p + 3, hard.p);
    if (largest > 2048) hipLaunchKernelGGL(k_run_bucket_sort<16>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 2049u, info.p + 3, hard.p);
    if (largest <= 2048) hipLaunchKernelGGL(k_run_sort<8>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 2048u, hard.p, info.p + 3);
    else if (largest <= 4096) hipLaunchKernelGGL(k_run_sort<16>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 4096u, hard.p, info.p + 3);
    else hipLaunchKernelGGL(k_run_sort<32>, dim3(grid / 4), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 8192u, hard.p, info.p + 3);
    HIP_TRY(hipGetLastError());
    if (n_big) {   // runs beyond 8192 keys: one LSD radix sort each, on the bits they still differ in
        std::vector<uint32_t> h_seg(2 * (size_t)n_big);
        for (uint32_t i = 0; i < n_big; ++i) HIP_TRY(hipMemcpyAsync(&h_seg[2 * i], seg + h_info[kInfo + i], 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, src, dst, kept, 0u, rest, st));   // (an upper bound: no run is larger than the array)
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb))) return rc;
        for (uint32_t i = 0; i < n_big; ++i) {
            const uint32_t lo = h_seg[2 * i], sz = h_seg[2 * i + 1] - lo;
            size_t tbi = tb;
            HIP_TRY(rocprim::radix_sort_keys(tmp.p, tbi, src + lo, dst + lo, sz, 0u, rest ? rest : 1u, st));
        }
    }
    *sorted = dst; *n_real = kept;
    return CID_OK;
}

// The same sort for a set built FOR an index: (key, code) pairs, partitioned on the key's leading bits, the runs finished in LDS on
// (rest of the key, code).  keys_a / codes_a hold the n windows (keys of kNoKey = no k-mer) and are overwritten; keys_b / codes_b
// are scratch.  On return *sorted (codes_a or codes_b) holds the *n_real real codes in (key, code) order.  The keys are evenly
// spread by construction (row0_key), so the runs all have about n >> prefix members.
struct KeyCodeLess {   // (row0_key, code) pairs, as a targeted set is ordered
    __host__ __device__ bool operator()(const rocprim::tuple<uint32_t, uint64_t> &a, const rocprim::tuple<uint32_t, uint64_t> &b) const {
        const uint32_t ka = rocprim::get<0>(a), kb = rocprim::get<0>(b);
        return ka < kb || (ka == kb && rocprim::get<1>(a) < rocprim::get<1>(b));
    }
};
int msd_sort_pair(cid_ctx *c, hipStream_t st, uint32_t *keys_a, uint64_t *codes_a, uint32_t *keys_b, uint64_t *codes_b, size_t n, unsigned code_bits,
                  uint64_t **sorted, size_t *n_real) {
    using namespace cid;
    const char *min_env = getenv("CID_KMERSET_MSD_MIN");
    const size_t min_n = min_env ? strtoull(min_env, nullptr, 10) : (size_t)1 << 20;
    if (!kMsdSort || n < min_n || n < 2 || n >= (1ull << 32)) return CID_ERR_UNSUPPORTED;
    unsigned prefix = 1;
    while (prefix < 24 && ((uint64_t)n >> prefix) > 1900) ++prefix;   // runs of ~950 .. 1900 pairs: k_run_bucket_sort_pair<8> takes up to 2048
    const unsigned levels = (prefix + kPartBits - 1) / kPartBits;
    unsigned lbits[3] = {0, 0, 0};
    for (unsigned l = 0; l < levels; ++l) lbits[l] = l + 1 < levels ? kPartBits : prefix - kPartBits * (levels - 1);
    const uint32_t n_runs = 1u << prefix;
    uint32_t S_last = 1;
    for (unsigned l = 0; l + 1 < levels; ++l) S_last <<= lbits[l];
    const uint32_t max_tiles = part_max_tiles((uint32_t)n, S_last);
    constexpr uint32_t kInfo = 8, kBigCap = 1024;          // info: [0] dropped (no k-mer) [1] big runs [2] largest run [3] hard runs
    DevBuf<uint32_t> seg_a(c), seg_b(c), tile_base(c), table(c), info(c), hard(c);
    DevBuf<uint8_t> scan_tmp(c);
    int rc;
    if ((rc = seg_a.alloc((size_t)n_runs + 1)) || (rc = seg_b.alloc((size_t)n_runs + 1)) || (rc = tile_base.alloc((size_t)S_last + 1)) ||
        (rc = table.alloc((size_t)max_tiles * kPartBins)) || (rc = info.alloc(kInfo + kBigCap)) || (rc = hard.alloc((size_t)n_runs + 1)))
        return rc;
    size_t scan_tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tb, table.p, table.p, 0u, (size_t)max_tiles * kPartBins, rocprim::plus<uint32_t>(), st));
    if ((rc = scan_tmp.alloc(scan_tb))) return rc;
    HIP_TRY(hipMemsetAsync(info.p, 0, kInfo * 4, st));
    const uint32_t seg0[2] = {0u, (uint32_t)n};
    HIP_TRY(hipMemcpyAsync(seg_a.p, seg0, 8, hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)cid::ctx_n_cu(c) * 8u;
    uint32_t *ksrc = keys_a, *kdst = keys_b;
    uint64_t *src = codes_a, *dst = codes_b;
    uint32_t *seg = seg_a.p, *seg_next = seg_b.p;
    uint32_t S = 1;
    unsigned consumed = 0;
    for (unsigned l = 0; l < levels; ++l) {
        const uint32_t bits = lbits[l], shift = 32 - consumed - bits, bins = 1u << bits;
        const size_t table_n = (size_t)part_max_tiles((uint32_t)n, S) * bins;
        HIP_TRY(hipMemsetAsync(table.p, 0, table_n * 4, st));
        hipLaunchKernelGGL(k_part_tiles, dim3(1), dim3(kPartBlock), 0, st, seg, S, tile_base.p);
        hipLaunchKernelGGL(k_part_hist_key, dim3(grid), dim3(kPartBlock), 0, st, ksrc, seg, tile_base.p, S, shift, bits, l == 0 ? 1u : 0u, table.p, info.p);
        HIP_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_tb, table.p, table.p, 0u, table_n, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(k_part_scatter_pair, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, kdst, dst, seg, tile_base.p, S, shift, bits, l == 0 ? 1u : 0u,
                           table.p);
        hipLaunchKernelGGL(k_part_offsets, dim3((S * bins + 256) / 256), dim3(256), 0, st, table.p, tile_base.p, S, bits, (uint32_t)n, info.p, seg_next);
        HIP_TRY(hipGetLastError());
        std::swap(src, dst);
        std::swap(ksrc, kdst);
        std::swap(seg, seg_next);
        S *= bins;
        consumed += bits;
    }
    hipLaunchKernelGGL(k_run_sizes, dim3((n_runs + 255) / 256), dim3(256), 0, st, seg, n_runs, 8192u, info.p + 1, info.p + kInfo, kBigCap, info.p + 2);
    uint32_t h_info[kInfo + kBigCap];
    HIP_TRY(hipMemcpyAsync(h_info, info.p, sizeof(h_info), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t dropped = h_info[0], n_big = h_info[1], largest = h_info[2];
    const size_t kept = n - dropped;
    const PairOrder ord{32u - consumed, code_bits};
    // two stable LSD sorts = the (key, code) order, for what the LDS kernels do not take: first by code, then by the key's rest
    auto lsd_pair = [&](size_t lo, size_t sz) -> int {
        size_t tb1 = 0, tb2 = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb1, src + lo, dst + lo, ksrc + lo, kdst + lo, sz, 0u, code_bits, st));
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, kdst + lo, ksrc + lo, dst + lo, src + lo, sz, 0u, ord.kbits ? ord.kbits : 1u, st));
        DevBuf<uint8_t> tmp(c);
        int rc2 = tmp.alloc(tb1 > tb2 ? tb1 : tb2);
        if (rc2) return rc2;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb1, src + lo, dst + lo, ksrc + lo, kdst + lo, sz, 0u, code_bits, st));
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb2, kdst + lo, ksrc + lo, dst + lo, src + lo, sz, 0u, ord.kbits ? ord.kbits : 1u, st));
        HIP_TRY(hipMemcpyAsync(dst + lo, src + lo, sz * 8, hipMemcpyDeviceToDevice, st));   // the result belongs in dst, like the LDS kernels'
        HIP_TRY(hipStreamSynchronize(st));   // (tmp is released on return)
        return CID_OK;
    };
    if (n_big > kBigCap) {   // one row taking a large share of the windows (low-complexity sequence): everything through the LSD sorts
        size_t tb1 = 0, tb2 = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb1, src, dst, ksrc, kdst, kept, 0u, code_bits, st));
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, kdst, ksrc, dst, src, kept, 0u, 32u, st));
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb1 > tb2 ? tb1 : tb2))) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb1, src, dst, ksrc, kdst, kept, 0u, code_bits, st));
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb2, kdst, ksrc, dst, src, kept, 0u, 32u, st));
        HIP_TRY(hipStreamSynchronize(st));
        *sorted = src; *n_real = kept;
        return CID_OK;
    }
    hipLaunchKernelGGL(k_run_bucket_sort_pair<8>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, info.p + 3, hard.p);
    if (largest > 2048) hipLaunchKernelGGL(k_run_bucket_sort_pair<16>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 2049u, info.p + 3, hard.p);
    if (largest <= 2048) hipLaunchKernelGGL(k_run_sort_pair<8>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 2048u, hard.p, info.p + 3);
    else if (largest <= 4096) hipLaunchKernelGGL(k_run_sort_pair<16>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 4096u, hard.p, info.p + 3);
    else hipLaunchKernelGGL(k_run_sort_pair<32>, dim3(grid / 4), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 8192u, hard.p, info.p + 3);
    HIP_TRY(hipGetLastError());
    if (n_big) {   // runs beyond 8192 pairs (one row's k-mers at deep coverage)
        std::vector<uint32_t> h_seg(2 * (size_t)n_big);
        for (uint32_t i = 0; i < n_big; ++i) HIP_TRY(hipMemcpyAsync(&h_seg[2 * i], seg + h_info[kInfo + i], 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t i = 0; i < n_big; ++i)
            if ((rc = lsd_pair(h_seg[2 * i], h_seg[2 * i + 1] - h_seg[2 * i]))) return rc;
    }
    *sorted = dst; *n_real = kept;
    return CID_OK;
}

// merge the raw window codes into (codes, counts): the batch is sorted and run-length counted (distinct codes + multiplicities,
// sentinel dropped); a set that already holds k-mers is then MERGED with it — two sorted lists, one pass (rocprim::merge), equal
// neighbours added (reduce_by_key) — instead of re-sorting everything it holds with every batch
int compact(cid_kmerset *ks) {
    if (ks->n_raw == 0) return CID_OK;
    hipStream_t st = cid::ctx_stream(ks->ctx);
    const size_t batch = ks->n_raw;
    if (ks->n + batch >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED,
                                                   "%zu k-mer windows and distinct k-mers in one merge (limit 2^32 - 1): add fewer sequences per set", ks->n + batch);
    DevBuf<uint64_t> uniq(ks->ctx), d_count(ks->ctx);
    DevBuf<uint32_t> agg(ks->ctx);
    int rc;
    if ((rc = uniq.alloc(batch)) || (rc = agg.alloc(batch)) || (rc = d_count.alloc(1))) return rc;
    uint64_t n_runs = 0;
    {
        DevBuf<uint64_t> sorted(ks->ctx);
        if ((rc = sorted.alloc(batch))) return rc;
        const uint64_t *in_order = sorted.p;
        size_t n_sorted = batch;
        uint64_t *msd_out = nullptr;
        if (ks->targeted) {   // (key, code) order
            DevBuf<uint32_t> key_b(ks->ctx);
            if ((rc = key_b.alloc(batch))) return rc;
            rc = msd_sort_pair(ks->ctx, st, ks->raw_key, ks->raw, key_b.p, sorted.p, batch, 2 * ks->k, &msd_out, &n_sorted);
            if (rc == CID_OK) in_order = msd_out;
            else if (rc != CID_ERR_UNSUPPORTED) return rc;
            else {   // small batches: two stable LSD sorts, by code and then by key; the windows without a k-mer (kNoKey, sentinel) sort last
                n_sorted = batch;
                size_t tb1 = 0, tb2 = 0;
                HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb1, ks->raw, sorted.p, ks->raw_key, key_b.p, batch, 0u, ks->end_bit, st));
                HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, key_b.p, ks->raw_key, sorted.p, ks->raw, batch, 0u, 32u, st));
                DevBuf<uint8_t> tmp(ks->ctx);
                if ((rc = tmp.alloc(tb1 > tb2 ? tb1 : tb2))) return rc;
                HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb1, ks->raw, sorted.p, ks->raw_key, key_b.p, batch, 0u, ks->end_bit, st));
                HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb2, key_b.p, ks->raw_key, sorted.p, ks->raw, batch, 0u, 32u, st));
                HIP_TRY(hipStreamSynchronize(st));
                in_order = ks->raw;
            }
        } else {
        // k <= 31: every real code is below bit end_bit - 1, the sentinel is that bit — the MSD sort drops it (ks->raw is overwritten)
        rc = ks->k <= 31 ? msd_sort(ks->ctx, st, ks->raw, sorted.p, batch, ks->end_bit - 1, &msd_out, &n_sorted) : CID_ERR_UNSUPPORTED;
        if (rc == CID_OK) in_order = msd_out;
        else if (rc != CID_ERR_UNSUPPORTED) return rc;
        else {
            n_sorted = batch;
            size_t tmp_bytes = 0;
            HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, ks->raw, sorted.p, batch, 0u, ks->end_bit, st));
            DevBuf<uint8_t> tmp(ks->ctx);
            if ((rc = tmp.alloc(tmp_bytes))) return rc;
            HIP_TRY(rocprim::radix_sort_keys(tmp.p, tmp_bytes, ks->raw, sorted.p, batch, 0u, ks->end_bit, st));
        }
        }
        if (n_sorted == 0) {   // nothing but invalid windows
            HIP_TRY(hipMemsetAsync(d_count.p, 0, 8, st));
        } else {
            size_t tmp2 = 0;
            HIP_TRY(rocprim::run_length_encode(nullptr, tmp2, in_order, n_sorted, uniq.p, agg.p, d_count.p, st));
            DevBuf<uint8_t> t2(ks->ctx);
            if ((rc = t2.alloc(tmp2))) return rc;
            HIP_TRY(rocprim::run_length_encode(t2.p, tmp2, in_order, n_sorted, uniq.p, agg.p, d_count.p, st));
        }
        HIP_TRY(hipMemcpyAsync(&n_runs, d_count.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n_runs > 0) {  // the sentinel (invalid windows) sorts last: drop it (the MSD sort has done so already)
            uint64_t last = 0;
            HIP_TRY(hipMemcpy(&last, uniq.p + (n_runs - 1), 8, hipMemcpyDeviceToHost));
            if (last == ks->sentinel) --n_runs;
        }
    }
    ks->n_raw = 0;
    if (ks->n == 0) {
        if (ks->codes) cid::ctx_free(ks->ctx, ks->codes);
        if (ks->counts) cid::ctx_free(ks->ctx, ks->counts);
        ks->codes = uniq.release();
        ks->counts = agg.release();
        ks->n = n_runs;
        return CID_OK;
    }
    if (n_runs == 0) return CID_OK;
    const size_t total = ks->n + n_runs;
    DevBuf<uint64_t> mk(ks->ctx), ok(ks->ctx);
    DevBuf<uint32_t> mv(ks->ctx), ov(ks->ctx);
    if ((rc = mk.alloc(total)) || (rc = mv.alloc(total)) || (rc = ok.alloc(total)) || (rc = ov.alloc(total))) return rc;
    size_t tb = 0;
    if (ks->targeted) {   // both lists are in (row0_key, code) order: merged on that pair (the keys are recomputed from the codes, not kept)
        DevBuf<uint32_t> ka(ks->ctx), kb(ks->ctx), kout(ks->ctx);
        if ((rc = ka.alloc(ks->n)) || (rc = kb.alloc(n_runs)) || (rc = kout.alloc(total))) return rc;
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->codes, ks->k, ks->key_for, ka.p, (uint64_t)ks->n);
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(n_runs)), dim3(256), 0, st, uniq.p, ks->k, ks->key_for, kb.p, (uint64_t)n_runs);
        HIP_TRY(hipGetLastError());
        auto in_a = rocprim::make_zip_iterator(rocprim::make_tuple(ka.p, ks->codes));
        auto in_b = rocprim::make_zip_iterator(rocprim::make_tuple(kb.p, uniq.p));
        auto out_k = rocprim::make_zip_iterator(rocprim::make_tuple(kout.p, mk.p));
        HIP_TRY(rocprim::merge(nullptr, tb, in_a, in_b, out_k, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, KeyCodeLess(), st));
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::merge(tmp.p, tb, in_a, in_b, out_k, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, KeyCodeLess(), st));
        HIP_TRY(hipStreamSynchronize(st));   // (ka / kb / kout are released here)
    } else {
    HIP_TRY(rocprim::merge(nullptr, tb, ks->codes, uniq.p, mk.p, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, rocprim::less<uint64_t>(), st));
    {
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::merge(tmp.p, tb, ks->codes, uniq.p, mk.p, ks->counts, agg.p, mv.p, ks->n, (size_t)n_runs, rocprim::less<uint64_t>(), st));
    }
    }
    size_t tmp2 = 0;
    HIP_TRY(rocprim::reduce_by_key(nullptr, tmp2, mk.p, mv.p, total, ok.p, ov.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
    DevBuf<uint8_t> t2(ks->ctx);
    if ((rc = t2.alloc(tmp2))) return rc;
    HIP_TRY(rocprim::reduce_by_key(t2.p, tmp2, mk.p, mv.p, total, ok.p, ov.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
    // multiplicities are u32 (the reference: usize): a sum that saturated cannot be reported faithfully
    HIP_TRY(hipMemsetAsync(ks->d_flags + 1, 0, 4, st));
    hipLaunchKernelGGL(cid::k_flag_saturated, dim3(grid_for_n(total)), dim3(256), 0, st, ov.p, d_count.p, ks->d_flags + 1);
    int sat = 0;
    uint64_t n_merged = 0;
    HIP_TRY(hipMemcpyAsync(&sat, ks->d_flags + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_merged, d_count.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sat) return fail(CID_ERR_UNSUPPORTED,
                         "a k-mer occurs more than 2^32 - 2 times: beyond the u32 multiplicities of the GPU k-mer set (count on the host)");
    cid::ctx_free(ks->ctx, ks->codes);
    cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = ok.release();
    ks->counts = ov.release();
    ks->n = n_merged;
    return CID_OK;
}

}  // namespace

// A finalized code set's contents replaced by the merge of `total` (code, count) pairs in any order (device arrays in ks's ctx; the
// caller keeps owning them): sort by code, equal codes' counts added (cid_group_kmerset: the ranges a rank receives from the others).
int cid::kmerset_assign_merged(cid_kmerset *ks, const uint64_t *d_codes_in, const uint32_t *d_counts_in, size_t total) {
    if (!ks || ks->general || !ks->finalized) return fail(CID_ERR_STATE, "kmerset_assign_merged: a finalized 2-bit-code set is needed");
    if (total >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "%zu k-mers in one rank's share of the set (limit 2^32 - 1): use more GPUs", total);
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    DevBuf<uint64_t> uniq(ks->ctx), kout(ks->ctx), d_count(ks->ctx);
    DevBuf<uint32_t> agg(ks->ctx), vout(ks->ctx);
    int rc;
    if ((rc = uniq.alloc(total)) || (rc = agg.alloc(total)) || (rc = kout.alloc(total)) || (rc = vout.alloc(total)) || (rc = d_count.alloc(1))) return rc;
    uint64_t n_runs = 0;
    if (total) {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_codes_in, kout.p, d_counts_in, vout.p, total, 0u, ks->end_bit, st));
        DevBuf<uint8_t> tmp(ks->ctx);
        if ((rc = tmp.alloc(tmp_bytes))) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, d_codes_in, kout.p, d_counts_in, vout.p, total, 0u, ks->end_bit, st));
        size_t tmp2 = 0;
        HIP_TRY(rocprim::reduce_by_key(nullptr, tmp2, kout.p, vout.p, total, uniq.p, agg.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
        DevBuf<uint8_t> t2(ks->ctx);
        if ((rc = t2.alloc(tmp2))) return rc;
        HIP_TRY(rocprim::reduce_by_key(t2.p, tmp2, kout.p, vout.p, total, uniq.p, agg.p, d_count.p, SatAdd(), rocprim::equal_to<uint64_t>(), st));
        HIP_TRY(hipMemsetAsync(ks->d_flags + 1, 0, 4, st));
        hipLaunchKernelGGL(cid::k_flag_saturated, dim3(grid_for_n(total)), dim3(256), 0, st, agg.p, d_count.p, ks->d_flags + 1);
        int sat = 0;
        HIP_TRY(hipMemcpyAsync(&sat, ks->d_flags + 1, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&n_runs, d_count.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (sat) return fail(CID_ERR_UNSUPPORTED,
                             "a k-mer occurs more than 2^32 - 2 times: beyond the u32 multiplicities of the GPU k-mer set (count on the host)");
    }
    if (ks->codes) cid::ctx_free(ks->ctx, ks->codes);
    if (ks->counts) cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = uniq.release();
    ks->counts = agg.release();
    ks->n = n_runs;
    return CID_OK;
}

// k > 32: all windows of the resident sequences -> distinct canonical byte strings + multiplicities
static int finalize_general(cid_kmerset *ks);

// ------------------------------------------------------------------------------------------------ long reads (read_id)
// Per-read distinct k-mers in first-occurrence order for reads whose k-mer set does not fit a wave's LDS:
// window codes -> one stable radix sort by code (windows are laid out read by read, so inside a run of equal codes
// the entries of one read are adjacent and ascending) -> first-occurrence flags -> exclusive scan -> ordered lists.
namespace cid {

__global__ void k_iota_u32(uint32_t *p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_first_flags(const uint64_t *sorted_codes, const uint32_t *sorted_idx, const uint64_t *wstart, uint32_t n_reads,
                              uint64_t sentinel, uint32_t *flags, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t code = sorted_codes[i];
    const uint32_t w = sorted_idx[i];
    bool first = code != sentinel;
    if (first && i > 0 && sorted_codes[i - 1] == code)
        first = read_of_window(wstart, n_reads, sorted_idx[i - 1]) != read_of_window(wstart, n_reads, w);
    flags[w] = first ? 1u : 0u;
}
__global__ void k_scatter_list(const uint64_t *codes, const uint32_t *flags, const uint32_t *pos, uint64_t *list, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) list[pos[i]] = codes[i];
}
__global__ void k_list_starts(const uint64_t *wstart, const uint32_t *pos, uint64_t *list_start, uint32_t n_reads) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) list_start[r] = pos[wstart[r]];
}

// ---- general keys: k > 32 or lower-case bases (their case is kept), so a key is a byte string.
// One thread per window: validity, canonical orientation and (for .mxi) the minimizer, all on raw bytes as the reference
// compares them.  A key is described by where its bytes sit in `bases`: entry = offset << 1 | reverse-complement flag
// (minimizers are upper-cased afterwards, kmer.rs:381); its sort image is 4 bits per base (2-bit base | lower-case << 2),
// 16 bases per word, word-major arrays; a window without a key gets all-ones words (no base encodes to 0xF).
__device__ __forceinline__ uint32_t key_byte(const uint8_t *bases, uint64_t entry, uint32_t klen, bool upper, uint32_t t) {
    const uint64_t off = entry >> 1;
    uint32_t b = (entry & 1ull) ? switch_base_dev(bases[off + klen - 1 - t]) : bases[off + t];
    if (upper && b >= 'a' && b <= 'z') b -= 32u;
    return b;
}
__global__ __launch_bounds__(256) void k_general_keys(const uint8_t *bases, const Segment *segs, uint32_t n_segs, uint64_t W, uint32_t k,
                                                      uint32_t msz, uint32_t n_words, uint64_t *keyw, uint64_t *entry, uint32_t upper_keys = 0) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    uint32_t lo = 0, hi = n_segs;   // the segment holding window w: largest s with segs[s].out_off <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].out_off <= w) lo = mid; else hi = mid;
    }
    const uint64_t pos = segs[lo].base_off + (w - segs[lo].out_off) * segs[lo].stride;
    const uint8_t *b = bases + pos;
    bool ok = true;
    for (uint32_t t = 0; t < k; ++t) ok = ok && good_base_dev(b[t]);
    if (!ok) {
        entry[w] = ~0ull;
        for (uint32_t j = 0; j < n_words; ++j) keyw[(uint64_t)j * W + w] = ~0ull;
        return;
    }
    uint32_t rc = 1;   // palindromes take the reverse-complement branch (the same string)
    for (uint32_t t = 0; t < k; ++t) {
        const uint32_t f = b[t], r = switch_base_dev(b[k - 1 - t]);
        if (f != r) { rc = f < r ? 0u : 1u; break; }
    }
    uint64_t e = (pos << 1) | rc;
    uint32_t klen = k;
    if (msz) {   // find_minimizer (kmer.rs:971-986) over the canonical string: candidates (i, reverse-complement)
        const uint64_t canon = e;
        auto cand_byte = [&](uint32_t cand, uint32_t t) -> uint32_t {
            const uint32_t i = cand & 0xFFFFu;
            return (cand >> 16) ? (uint32_t)switch_base_dev((uint8_t)key_byte(bases, canon, k, false, i + msz - 1 - t))
                                : key_byte(bases, canon, k, false, i + t);
        };
        auto less = [&](uint32_t x, uint32_t y) -> bool {
            for (uint32_t t = 0; t < msz; ++t) {
                const uint32_t bx = cand_byte(x, t), by = cand_byte(y, t);
                if (bx != by) return bx < by;
            }
            return false;
        };
        uint32_t best = 0;
        for (uint32_t i = 1; i + msz <= k; ++i) {
            if (less(i, best)) best = i;
            if (less(i | (1u << 16), best)) best = i | (1u << 16);
        }
        // the minimizer as a stretch of `bases`: canonical byte j is b[j] (rc = 0) or comp(b[k-1-j]) (rc = 1)
        const uint32_t i = best & 0xFFFFu, mrc = best >> 16;
        const uint64_t off = rc ? pos + k - i - msz : pos + i;
        e = (off << 1) | (rc ^ mrc);
        klen = msz;
    }
    entry[w] = e;
    for (uint32_t j = 0; j < n_words; ++j) {
        uint64_t word = 0;
        for (uint32_t t = 16 * j; t < 16 * j + 16 && t < klen; ++t) {
            const uint32_t c = key_byte(bases, e, klen, msz != 0 || upper_keys != 0, t);
            word |= (uint64_t)(((c >> 1) & 3u) | ((c >> 3) & 4u)) << (4u * (t & 15u));
        }
        keyw[(uint64_t)j * W + w] = word;
    }
}
__global__ void k_gather_u64(const uint64_t *src, const uint32_t *idx, uint64_t *dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_first_flags_general(const uint64_t *keyw, uint32_t n_words, const uint64_t *entry, const uint32_t *sorted_idx,
                                      const uint64_t *wstart, uint32_t n_reads, uint32_t *flags, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    const uint32_t w = sorted_idx[i];
    bool first = entry[w] != ~0ull;
    if (first && i > 0) {
        const uint32_t pw = sorted_idx[i - 1];
        bool same = true;
        for (uint32_t j = 0; j < n_words && same; ++j) same = keyw[(uint64_t)j * W + w] == keyw[(uint64_t)j * W + pw];
        if (same) first = read_of_window(wstart, n_reads, pw) != read_of_window(wstart, n_reads, w);
    }
    flags[w] = first ? 1u : 0u;
}

// ---- byte-string k-mer sets (k > 32): run boundaries over the whole sorted window list
__global__ void k_first_flags_set(const uint64_t *keyw, uint32_t n_words, const uint64_t *entry, const uint32_t *sorted_idx, uint32_t *flags,
                                  uint32_t *valid, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > W) return;
    if (i == W) { flags[i] = 0; valid[i] = 0; return; }   // slot W receives the totals
    const uint32_t w = sorted_idx[i];
    const bool ok = entry[w] != ~0ull;
    bool first = ok;
    if (first && i > 0) {
        const uint32_t pw = sorted_idx[i - 1];
        bool same = true;
        for (uint32_t j = 0; j < n_words && same; ++j) same = keyw[(uint64_t)j * W + w] == keyw[(uint64_t)j * W + pw];
        first = !same;
    }
    flags[i] = first ? 1u : 0u;
    valid[i] = ok ? 1u : 0u;
}
// run j starts at sorted position starts[j]; starts[n_runs] = number of valid windows (they sort before the invalid ones)
__global__ void k_set_starts(const uint32_t *flags, const uint32_t *pos, const uint32_t *sorted_idx, const uint64_t *entry, uint32_t *starts,
                             uint64_t *run_entry, uint64_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W && flags[i]) { starts[pos[i]] = (uint32_t)i; run_entry[pos[i]] = entry[sorted_idx[i]]; }
}
__global__ void k_run_counts(const uint32_t *starts, uint32_t n_runs, uint32_t n_valid, uint32_t *counts) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_runs) counts[j] = (j + 1 < n_runs ? starts[j + 1] : n_valid) - starts[j];
}
__global__ void k_entries_to_ascii(const uint8_t *bases, const uint64_t *run_entry, uint32_t k, uint32_t upper, uint8_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = run_entry[i];
    for (uint32_t t = 0; t < k; ++t) out[i * k + t] = (uint8_t)key_byte(bases, e, k, upper != 0, t);
}
__global__ void k_compact_rows(const uint8_t *rows_in, const uint32_t *counts_in, const uint32_t *keep, const uint32_t *pos, uint32_t k,
                               uint8_t *rows_out, uint32_t *counts_out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint64_t o = pos[i];
    for (uint32_t t = 0; t < k; ++t) rows_out[o * k + t] = rows_in[i * k + t];
    counts_out[o] = counts_in[i];
}
__global__ void k_keep_gt(const uint32_t *counts, uint64_t t, uint32_t *keep, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    keep[i] = (i < n && counts[i] > t) ? 1u : 0u;
}

// d_bases resident; host seq_off / read_seq0.  Writes report / n_kmers / status to DEVICE arrays.
int readid_long_sorted(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                       size_t n_reads, uint32_t stride_d, uint32_t start_sample, const uint8_t *route, bool clear_wide, uint32_t *d_report,
                       uint32_t *d_n_kmers, uint8_t *d_status, const StripePass &sp) {
    const uint32_t k = index_k(ix);
    hipStream_t st = ctx_stream(c);
    const uint32_t msz = index_m_size(ix);           // > 0: the sets hold minimizers of length msz
    const uint32_t key_len = msz ? msz : k;
    bool general = k > 32;                           // byte-string keys; also taken when a lower-case base shows up
    const uint64_t sentinel_k = k < 32 ? (1ull << (2 * k)) : ~0ull;
    const uint64_t sentinel = key_len < 32 ? (1ull << (2 * key_len)) : ~0ull;
    const unsigned end_bit = key_len < 32 ? 2 * key_len + 1 : 64;
    // windows are numbered read by read, mate by mate
    std::vector<uint64_t> wstart(n_reads + 1, 0);
    std::vector<uint8_t> status(n_reads, 0);
    std::vector<Segment> segs;
    const uint32_t seg_win = kSegWindows / stride_d ? kSegWindows / stride_d : 1;
    uint64_t W = 0;
    for (size_t r = 0; r < n_reads; ++r) {
        wstart[r] = W;
        if (route && !route[r]) { status[r] = 2; continue; }
        const uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
        if (s1 == s0 || seq_off[s0 + 1] - seq_off[s0] < k) { status[r] = 1; continue; }   // too_short (first mate only)
        for (uint64_t s = s0; s < s1; ++s) {
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len < k) continue;
            const uint64_t nw = (len - k) / stride_d + 1;
            for (uint64_t w0 = 0; w0 < nw; w0 += seg_win) {
                const uint32_t m = (uint32_t)(nw - w0 < seg_win ? nw - w0 : seg_win);
                segs.push_back(Segment{seq_off[s] + w0 * stride_d, W, m, stride_d});
                W += m;
            }
        }
    }
    wstart[n_reads] = W;
    if (W >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    const size_t C1 = (size_t)index_n_colors(ix) + 1;
    HIP_TRY(hipMemcpyAsync(d_status, status.data(), n_reads, hipMemcpyHostToDevice, st));
    DevBuf<uint64_t> d_wstart(c), d_codes(c), d_sorted(c), d_list(c), d_lstart(c);
    DevBuf<uint32_t> d_idx(c), d_sidx(c), d_flags(c), d_pos(c);
    DevBuf<Segment> d_segs(c);
    DevBuf<int> d_lower(c);
    int rc;
    if ((rc = d_wstart.alloc(n_reads + 1)) || (rc = d_codes.alloc(W + 1)) || (rc = d_sorted.alloc(W + 1)) || (rc = d_idx.alloc(W + 1)) ||
        (rc = d_sidx.alloc(W + 1)) || (rc = d_flags.alloc(W + 1)) || (rc = d_pos.alloc(W + 1)) || (rc = d_segs.alloc(segs.size())) ||
        (rc = d_lstart.alloc(n_reads + 1)) || (rc = d_lower.alloc(4))) return rc;
    HIP_TRY(hipMemcpyAsync(d_wstart.p, wstart.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_lower.p, 0, 16, st));
    HIP_TRY(hipMemsetAsync(d_flags.p, 0, (W + 1) * 4, st));
    if (W) {
        HIP_TRY(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice, st));
        if (!general) {
            constexpr uint32_t kBytes = kSegWindows + 32 + 96;
            const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
            unsigned grid = (unsigned)((segs.size() + 3) / 4);
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, d_bases, d_segs.p, (uint32_t)segs.size(), k, 1, sentinel_k,
                               d_codes.p, d_lower.p, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr, KeyFor{});
            int lower = 0;
            HIP_TRY(hipMemcpyAsync(&lower, d_lower.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            general = lower != 0;   // case-preserving k-mers cannot be packed in 2 bits per base
        }
        hipLaunchKernelGGL(k_iota_u32, dim3(grid_for_n(W)), dim3(256), 0, st, d_idx.p, (uint64_t)W);
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_codes.p, d_sorted.p, d_idx.p, d_sidx.p, W, 0u, 64u, st));
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb))) return rc;
        if (!general) {
            if (msz) hipLaunchKernelGGL(k_codes_to_minimizers, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, (uint64_t)W, k, msz, sentinel_k, sentinel);
            HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, d_codes.p, d_sorted.p, d_idx.p, d_sidx.p, W, 0u, end_bit, st));
            hipLaunchKernelGGL(k_first_flags, dim3(grid_for_n(W)), dim3(256), 0, st, d_sorted.p, d_sidx.p, d_wstart.p, (uint32_t)n_reads,
                               sentinel, d_flags.p, (uint64_t)W);
        } else {
            // stable LSD radix sort over the key words; d_codes ends up holding the entries the search kernel reads
            const uint32_t n_words = (key_len + 15) / 16;
            DevBuf<uint64_t> d_keyw(c), d_gath(c);
            if ((rc = d_keyw.alloc((size_t)n_words * W)) || (rc = d_gath.alloc(W))) return rc;
            hipLaunchKernelGGL(k_general_keys, dim3(grid_for_n(W)), dim3(256), 0, st, d_bases, d_segs.p, (uint32_t)segs.size(), (uint64_t)W, k, msz,
                               n_words, d_keyw.p, d_codes.p);
            uint32_t *cur = d_idx.p, *nxt = d_sidx.p;
            for (uint32_t j = 0; j < n_words; ++j) {
                hipLaunchKernelGGL(k_gather_u64, dim3(grid_for_n(W)), dim3(256), 0, st, d_keyw.p + (size_t)j * W, cur, d_gath.p, (uint64_t)W);
                HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, d_gath.p, d_sorted.p, cur, nxt, W, 0u, 64u, st));
                std::swap(cur, nxt);
            }
            hipLaunchKernelGGL(k_first_flags_general, dim3(grid_for_n(W)), dim3(256), 0, st, d_keyw.p, n_words, d_codes.p, cur, d_wstart.p,
                               (uint32_t)n_reads, d_flags.p, (uint64_t)W);
            HIP_TRY(hipStreamSynchronize(st));   // d_keyw / d_gath go out of scope
        }
    }
    size_t tb2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, d_flags.p, d_pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    DevBuf<uint8_t> tmp2(c);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, d_flags.p, d_pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    uint32_t D = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&D, d_pos.p + W, 4, hipMemcpyDeviceToHost));
    if ((rc = d_list.alloc(D))) return rc;
    if (W) hipLaunchKernelGGL(k_scatter_list, dim3(grid_for_n(W)), dim3(256), 0, st, d_codes.p, d_flags.p, d_pos.p, d_list.p, (uint64_t)W);
    hipLaunchKernelGGL(k_list_starts, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, d_wstart.p, d_pos.p, d_lstart.p,
                       (uint32_t)n_reads);
    ReadIdListParams p{};
    p.mat = index_matrix(ix); p.rs = index_rs(ix); p.w64 = (index_n_colors(ix) + 63) / 64; p.n_colors = index_n_colors(ix);
    p.n_hash = index_n_hash(ix); p.k = key_len; p.mod = index_mod(ix);
    p.list_codes = d_list.p; p.list_start = d_lstart.p; p.n_reads = n_reads; p.start_sample = start_sample;
    p.bases = general ? d_bases : nullptr; p.upper = msz != 0;
    p.hist_pad = p.rs > 128 ? 4u * p.rs : (uint32_t)((C1 + 3) & ~(size_t)3);
    if (p.rs > 128 && clear_wide && !sp.on()) HIP_TRY(hipMemsetAsync(d_report, 0, n_reads * C1 * 4, st));   // wide rows count in place
    p.zero_acc = sp.zero_acc; p.zero_in = sp.zero_in; p.zero_start = sp.zero_start;   // a colour stripe's pass: the caller zeroed the report
    p.colour_base = sp.colour_base; p.report_width = sp.report_width; p.write_nohits = sp.write_nohits;
    p.wave_bytes = (uint32_t)((4ull * kWave * p.n_hash + 4ull * p.hist_pad + 15) & ~15ull);
    if ((size_t)(kBlock / kWave) * p.wave_bytes > 160u * 1024u) return fail(CID_ERR_UNSUPPORTED, "LDS need exceeds 160 KiB");
    p.report = d_report; p.n_kmers = d_n_kmers; p.status = d_status;
    uint64_t grid = (n_reads + 3) / 4;
    if (grid > 4096) grid = 4096;
    HIP_TRY(launch_readid_list(p, (int)grid, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CID_OK;
}

}  // namespace cid

namespace cid {

hipError_t warm_kmerset() {   // see warm_readid (cid_readid.hip): this file's kernels serve k-mer counting and the sort-based read_id path
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_seq_windows));
}

}  // namespace cid

static int finalize_general(cid_kmerset *ks) {
    using namespace cid;
    cid_ctx *c = ks->ctx;
    hipStream_t st = ctx_stream(c);
    const uint64_t W = ks->g_windows;
    const uint32_t k = ks->k;
    ks->n = 0;
    if (W == 0) return CID_OK;
    if (W >= (1ull << 32) - 1) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 - 2 k-mer windows in one byte-string k-mer set");
    const uint32_t n_words = (k + 15) / 16;
    const uint32_t upper = ks->g_mode == 0 ? 1u : 0u;   // kmerize_vector upper-cases after the canonical choice (kmer.rs:104-117); fastq keeps case
    DevBuf<Segment> d_segs(c);
    DevBuf<uint64_t> keyw(c), entry(c), gath(c), sorted(c), run_entry(c);
    DevBuf<uint32_t> idx(c), idx2(c), flags(c), valid(c), pos(c), vpos(c), starts(c);
    int rc;
    if ((rc = d_segs.alloc(ks->g_segs.size())) || (rc = keyw.alloc((size_t)n_words * W)) || (rc = entry.alloc(W)) || (rc = gath.alloc(W)) ||
        (rc = sorted.alloc(W)) || (rc = idx.alloc(W)) || (rc = idx2.alloc(W)) || (rc = flags.alloc(W + 1)) || (rc = valid.alloc(W + 1)) ||
        (rc = pos.alloc(W + 1)) || (rc = vpos.alloc(W + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(d_segs.p, ks->g_segs.data(), ks->g_segs.size() * sizeof(Segment), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_general_keys, dim3(grid_for_n(W)), dim3(256), 0, st, ks->g_bases, d_segs.p, (uint32_t)ks->g_segs.size(), W, k, 0u, n_words,
                       keyw.p, entry.p, upper);
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_for_n(W)), dim3(256), 0, st, idx.p, W);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, gath.p, sorted.p, idx.p, idx2.p, W, 0u, 64u, st));
    DevBuf<uint8_t> tmp(c);
    if ((rc = tmp.alloc(tb))) return rc;
    uint32_t *cur = idx.p, *nxt = idx2.p;
    for (uint32_t j = 0; j < n_words; ++j) {   // stable LSD passes, least significant word (the key's LAST bases) first
        const uint32_t word = n_words - 1 - j;
        hipLaunchKernelGGL(k_gather_u64, dim3(grid_for_n(W)), dim3(256), 0, st, keyw.p + (size_t)word * W, cur, gath.p, W);
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, gath.p, sorted.p, cur, nxt, W, 0u, 64u, st));
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL(k_first_flags_set, dim3(grid_for_n(W + 1)), dim3(256), 0, st, keyw.p, n_words, entry.p, cur, flags.p, valid.p, W);
    size_t tb2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, flags.p, pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    DevBuf<uint8_t> tmp2(c);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, flags.p, pos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    HIP_TRY(rocprim::exclusive_scan(tmp2.p, tb2, valid.p, vpos.p, 0u, W + 1, rocprim::plus<uint32_t>(), st));
    uint32_t n_runs = 0, n_valid = 0;
    HIP_TRY(hipMemcpyAsync(&n_runs, pos.p + W, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_valid, vpos.p + W, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_runs) {
        DevBuf<uint32_t> counts(c);
        DevBuf<uint8_t> ascii(c);
        if ((rc = starts.alloc(n_runs)) || (rc = run_entry.alloc(n_runs)) || (rc = counts.alloc(n_runs)) || (rc = ascii.alloc((size_t)n_runs * k))) return rc;
        hipLaunchKernelGGL(k_set_starts, dim3(grid_for_n(W)), dim3(256), 0, st, flags.p, pos.p, cur, entry.p, starts.p, run_entry.p, W);
        hipLaunchKernelGGL(k_run_counts, dim3(grid_for_n(n_runs)), dim3(256), 0, st, starts.p, n_runs, n_valid, counts.p);
        hipLaunchKernelGGL(k_entries_to_ascii, dim3(grid_for_n(n_runs)), dim3(256), 0, st, ks->g_bases, run_entry.p, k, upper, ascii.p, (uint64_t)n_runs);
        HIP_TRY(hipStreamSynchronize(st));
        ks->ascii = ascii.release();
        ks->counts = counts.release();
    }
    ks->n = n_runs;
    HIP_TRY(hipStreamSynchronize(st));
    return CID_OK;
}

namespace cid {
int kmerset_view_ascii(const cid_kmerset *ks, cid_ctx **ctx, const uint8_t **ascii, const uint32_t **counts, uint64_t *n, uint32_t *k) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (!ks->general) return fail(CID_ERR_INVALID, "not a byte-string k-mer set");
    *ctx = ks->ctx; *ascii = ks->ascii; *counts = ks->counts; *n = ks->n; *k = ks->k;
    return CID_OK;
}
}  // namespace cid

namespace {
// the window-code kernel for a set: keyed when the set is built for an index
void launch_extract(const cid_kmerset *ks, hipStream_t st, unsigned grid, size_t shmem, const uint8_t *bases, const cid::Segment *segs, uint32_t n_segs, int mode,
                    const uint64_t *seq_off, const uint64_t *win_off, uint64_t base0) {
    if (ks->targeted)
        hipLaunchKernelGGL(cid::k_extract_codes<true>, dim3(grid), dim3(256), shmem, st, bases, segs, n_segs, ks->k, mode, ks->sentinel, ks->raw, ks->d_flags, seq_off,
                           win_off, base0, ks->raw_key, ks->key_for);
    else
        hipLaunchKernelGGL(cid::k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, bases, segs, n_segs, ks->k, mode, ks->sentinel, ks->raw, ks->d_flags, seq_off,
                           win_off, base0, (uint32_t *)nullptr, cid::KeyFor{});
}
// room for `want_n` window codes (and their keys) in the unsorted buffer, what it holds kept
int grow_raw(cid_kmerset *ks, size_t need_n, hipStream_t st) {
    if (need_n <= ks->cap_raw) return CID_OK;
    cid_ctx *c = ks->ctx;
    const size_t want = need_n * 3 / 2;
    DevBuf<uint64_t> nb(c);
    DevBuf<uint32_t> nk(c);
    int rc = nb.alloc(want);
    if (rc) return rc;
    if (ks->targeted && (rc = nk.alloc(want))) return rc;
    // on the ctx stream and waited for: a device-to-device hipMemcpy on the null stream returns before it has run, and the block
    // freed below is handed out again at once (as this call's d_bases) — the copy then read ASCII bases as codes
    if (ks->n_raw) HIP_TRY(hipMemcpyAsync(nb.p, ks->raw, ks->n_raw * 8, hipMemcpyDeviceToDevice, st));
    if (ks->n_raw && ks->targeted) HIP_TRY(hipMemcpyAsync(nk.p, ks->raw_key, ks->n_raw * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ks->raw) cid::ctx_free(c, ks->raw);
    if (ks->raw_key) cid::ctx_free(c, ks->raw_key);
    ks->raw = nb.release();
    ks->raw_key = ks->targeted ? nk.release() : nullptr;
    ks->cap_raw = want;
    return CID_OK;
}
}  // namespace

extern "C" {

int cid_kmerset_create(cid_ctx *c, uint32_t k_size, cid_kmerset **out) {
    if (!c || !out) return fail(CID_ERR_INVALID, "null ctx/out");
    *out = nullptr;
    if (k_size == 0 || k_size > cid::kMaxK) return fail(CID_ERR_UNSUPPORTED, "k_size %u outside 1..%u", k_size, cid::kMaxK);
    cid_kmerset *ks = new (std::nothrow) cid_kmerset();
    if (!ks) return fail(CID_ERR_NOMEM, "kmerset");
    ks->ctx = c; ks->k = k_size;
    ks->general = k_size > 32;   // beyond 32 bases a k-mer no longer packs into one u64: byte-string keys
    ks->sentinel = k_size < 32 ? (1ull << (2 * k_size)) : ~0ull;   // never a canonical code (T^32's canonical form is A^32)
    ks->end_bit = k_size < 32 ? 2 * k_size + 1 : 64;
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    if (hipMalloc(reinterpret_cast<void **>(&ks->d_flags), 16) != hipSuccess) { delete ks; return fail(CID_ERR_NOMEM, "flags"); }
    HIP_TRY(hipMemset(ks->d_flags, 0, 16));
    if (const char *e = getenv("CID_KMERSET_COMPACT_WINDOWS")) ks->compact_at = strtoull(e, nullptr, 10);  // tests
    *out = ks;
    return CID_OK;
}

int cid_kmerset_add_seqs(cid_kmerset *ks, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs, int mode) {
    if (!ks || !seq_off || (mode != 0 && mode != 1)) return fail(CID_ERR_INVALID, "bad argument");
    if (ks->finalized) return fail(CID_ERR_STATE, "k-mer set already finalized");
    if (n_seqs == 0) return CID_OK;
    const uint64_t total_bases = seq_off[n_seqs];
    if (total_bases && !bases) return fail(CID_ERR_INVALID, "null bases");
    if (ks->general) {   // keep the sequences resident; the windows are keyed, sorted and counted at finalize
        if (ks->g_mode >= 0 && ks->g_mode != mode) return fail(CID_ERR_INVALID, "one byte-string k-mer set takes one mode");
        ks->g_mode = mode;
        HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
        hipStream_t st = cid::ctx_stream(ks->ctx);
        for (size_t q = 0; q < n_seqs; ++q) {
            if (seq_off[q + 1] < seq_off[q]) return fail(CID_ERR_INVALID, "seq_off not monotonic");
            const uint64_t len = seq_off[q + 1] - seq_off[q];
            if (len < ks->k) continue;
            const uint64_t nw = len - ks->k + 1;
            for (uint64_t w0 = 0; w0 < nw; w0 += cid::kSegWindows) {
                const uint32_t mw = (uint32_t)(nw - w0 < cid::kSegWindows ? nw - w0 : cid::kSegWindows);
                ks->g_segs.push_back(cid::Segment{ks->g_n + seq_off[q] + w0, ks->g_windows, mw, 1});
                ks->g_windows += mw;
            }
        }
        if (ks->g_n + total_bases > ks->g_cap) {
            const size_t want = (ks->g_n + total_bases) * 3 / 2 + 256;
            DevBuf<uint8_t> nb(ks->ctx);
            int rc = nb.alloc(want);
            if (rc) return rc;
            if (ks->g_n) HIP_TRY(hipMemcpyAsync(nb.p, ks->g_bases, ks->g_n, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (ks->g_bases) cid::ctx_free(ks->ctx, ks->g_bases);
            ks->g_bases = nb.release();
            ks->g_cap = want;
        }
        if (total_bases) HIP_TRY(hipMemcpyAsync(ks->g_bases + ks->g_n, bases, total_bases, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        ks->g_n += total_bases;
        return CID_OK;
    }
    // sizes first: the windows of the batch and its longest sequence
    uint64_t n_win_total = 0, max_len = 0;
    for (size_t s = 0; s < n_seqs; ++s) {
        if (seq_off[s + 1] < seq_off[s]) return fail(CID_ERR_INVALID, "seq_off not monotonic");
        const uint64_t len = seq_off[s + 1] - seq_off[s];
        max_len = len > max_len ? len : max_len;
        n_win_total += len >= ks->k ? len - ks->k + 1 : 0;
    }
    if (n_win_total == 0) return CID_OK;
    cid_ctx *c = ks->ctx;
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    hipStream_t st = cid::ctx_stream(c);
    int rc;
    if ((rc = grow_raw(ks, ks->n_raw + n_win_total, st))) return rc;
    constexpr uint32_t kBytes = cid::kSegWindows + 32 + 96;
    const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
    DevBuf<uint8_t> d_bases(c);
    if ((rc = d_bases.alloc(total_bases))) return rc;
    int flag = 0;
    if (max_len <= cid::kSegWindows + ks->k - 1) {
        // Reads: one segment per sequence, derived on the device from the offsets (no per-read host work, no 24-byte segment record
        // per read over PCIe); the bases go up in slices of ~32 MiB on the ctx's copy stream and each slice's windows are extracted
        // while the next slice is still on the bus — what remains of the call is the PCIe time of the bases.
        DevBuf<uint64_t> d_off(c), d_win(c);
        if ((rc = d_off.alloc(n_seqs + 1)) || (rc = d_win.alloc(n_seqs + 1))) return rc;
        HIP_TRY(hipMemcpyAsync(d_off.p, seq_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(cid::k_seq_windows, dim3(grid_for_n(n_seqs)), dim3(256), 0, st, d_off.p, (uint64_t)n_seqs, ks->k, d_win.p);
        size_t tb = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_win.p, d_win.p, (uint64_t)ks->n_raw, n_seqs, rocprim::plus<uint64_t>(), st));
        DevBuf<uint8_t> tmp(c);
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, d_win.p, d_win.p, (uint64_t)ks->n_raw, n_seqs, rocprim::plus<uint64_t>(), st));
        const bool piped = st == cid::ctx_own_stream(c);     // a borrowed stream: keep everything on it
        hipStream_t cs = piped ? cid::ctx_copy_stream(c) : st;
        hipEvent_t ev_scan = cid::ctx_event(c, 0), ev_done = cid::ctx_event(c, 1);
        if (piped) { HIP_TRY(hipEventRecord(ev_scan, st)); HIP_TRY(hipStreamWaitEvent(cs, ev_scan, 0)); }   // (d_bases may still be read by an earlier kernel of the ctx stream)
        static const uint64_t slice_bytes = (uint64_t)(getenv("CID_KMERSET_SLICE_MB") ? atoi(getenv("CID_KMERSET_SLICE_MB")) : 32) << 20;   // (experiments)
        for (size_t s0 = 0; s0 < n_seqs;) {
            size_t s1 = s0;
            while (s1 < n_seqs && seq_off[s1] - seq_off[s0] < slice_bytes) ++s1;
            const uint64_t b0 = seq_off[s0], nb = seq_off[s1] - b0;
            if (nb) HIP_TRY(hipMemcpyAsync(d_bases.p + (b0 - seq_off[0]), bases + b0, nb, hipMemcpyHostToDevice, cs));
            if (piped) { HIP_TRY(hipEventRecord(ev_done, cs)); HIP_TRY(hipStreamWaitEvent(st, ev_done, 0)); }
            unsigned grid = (unsigned)((s1 - s0 + 3) / 4);
            if (grid > 8192) grid = 8192;
            launch_extract(ks, st, grid, shmem, d_bases.p, nullptr, (uint32_t)(s1 - s0), mode, d_off.p + s0, d_win.p + s0, seq_off[0]);
            HIP_TRY(hipGetLastError());
            s0 = s1;
        }
        if (mode == 1) HIP_TRY(hipMemcpyAsync(&flag, ks->d_flags, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {   // long sequences (genomes, contigs): cut into segments of kSegWindows windows on the host — a handful per megabase
        std::vector<cid::Segment> segs;
        uint64_t placed = 0;
        for (size_t s = 0; s < n_seqs; ++s) {
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len < ks->k) continue;
            const uint64_t nw = len - ks->k + 1;
            for (uint64_t w0 = 0; w0 < nw; w0 += cid::kSegWindows) {
                const uint32_t m = (uint32_t)(nw - w0 < cid::kSegWindows ? nw - w0 : cid::kSegWindows);
                segs.push_back(cid::Segment{seq_off[s] - seq_off[0] + w0, ks->n_raw + placed, m, 1});
                placed += m;
            }
        }
        DevBuf<cid::Segment> d_segs(c);
        if ((rc = d_segs.alloc(segs.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(d_bases.p, bases + seq_off[0], total_bases - seq_off[0], hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(cid::Segment), hipMemcpyHostToDevice, st));
        unsigned grid = (unsigned)((segs.size() + 3) / 4);
        if (grid > 8192) grid = 8192;
        launch_extract(ks, st, grid, shmem, d_bases.p, d_segs.p, (uint32_t)segs.size(), mode, nullptr, nullptr, 0);
        HIP_TRY(hipGetLastError());
        if (mode == 1) HIP_TRY(hipMemcpyAsync(&flag, ks->d_flags, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    ks->n_raw += n_win_total;
    if (mode == 1 && flag) return fail(CID_ERR_UNSUPPORTED, "lower-case bases in a case-preserving (fastq) k-mer count: count on the host");
    if (ks->n_raw > ks->compact_at) return compact(ks);
    return CID_OK;
}

// add_seqs for reads that are already on the device (the FASTQ front end's packed batch): d_seq_off[n_seqs + 1] are offsets into d_bases;
// every sequence at most max_len bases, which must fit one segment (reads, not genomes)
int cid_kmerset_add_seqs_dev(cid_kmerset *ks, const uint8_t *d_bases, const uint64_t *d_seq_off, size_t n_seqs, uint64_t max_len, int mode) {
    if (!ks || !d_seq_off || (mode != 0 && mode != 1)) return fail(CID_ERR_INVALID, "bad argument");
    if (ks->finalized) return fail(CID_ERR_STATE, "k-mer set already finalized");
    if (ks->general) return fail(CID_ERR_UNSUPPORTED, "cid_kmerset_add_seqs_dev takes k <= 32 (byte-string sets keep the host-pointer call)");
    if (max_len > cid::kSegWindows + ks->k - 1)
        return fail(CID_ERR_UNSUPPORTED, "cid_kmerset_add_seqs_dev takes reads of at most %u bases (longer sequences: cid_kmerset_add_seqs)", cid::kSegWindows + ks->k - 1);
    if (n_seqs == 0) return CID_OK;
    if (!d_bases) return fail(CID_ERR_INVALID, "null bases");
    cid_ctx *c = ks->ctx;
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    hipStream_t st = cid::ctx_stream(c);
    int rc;
    DevBuf<uint64_t> d_win(c);
    if ((rc = d_win.alloc(n_seqs + 1))) return rc;
    HIP_TRY(hipMemsetAsync(d_win.p + n_seqs, 0, 8, st));
    hipLaunchKernelGGL(cid::k_seq_windows, dim3(grid_for_n(n_seqs)), dim3(256), 0, st, d_seq_off, (uint64_t)n_seqs, ks->k, d_win.p);
    size_t tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_win.p, d_win.p, (uint64_t)ks->n_raw, n_seqs + 1, rocprim::plus<uint64_t>(), st));
    DevBuf<uint8_t> tmp(c);
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, d_win.p, d_win.p, (uint64_t)ks->n_raw, n_seqs + 1, rocprim::plus<uint64_t>(), st));
    uint64_t end = 0;
    HIP_TRY(hipMemcpyAsync(&end, d_win.p + n_seqs, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t n_win_total = end - ks->n_raw;
    if (n_win_total == 0) return CID_OK;
    if ((rc = grow_raw(ks, end, st))) return rc;
    constexpr uint32_t kBytes = cid::kSegWindows + 32 + 96;
    const size_t shmem = 4 * (kBytes + 4 * (kBytes / 16 + 4) + 2 * 4 * (kBytes / 32 + 4));
    unsigned grid = (unsigned)((n_seqs + 3) / 4);
    if (grid > 8192) grid = 8192;
    launch_extract(ks, st, grid, shmem, d_bases, nullptr, (uint32_t)n_seqs, mode, d_seq_off, d_win.p, 0);
    HIP_TRY(hipGetLastError());
    int flag = 0;
    if (mode == 1) HIP_TRY(hipMemcpyAsync(&flag, ks->d_flags, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ks->n_raw += n_win_total;
    if (mode == 1 && flag) return fail(CID_ERR_UNSUPPORTED, "lower-case bases in a case-preserving (fastq) k-mer count: count on the host");
    if (ks->n_raw > ks->compact_at) return compact(ks);
    return CID_OK;
}

// Build the set FOR an index (before the first sequence is added): it then comes out ordered by (first row in that index, code)
// instead of by code — the search's first-row fetches of neighbouring k-mers share 128-byte lines of the matrix — at the cost of
// four more bytes per window through the sort.  Contents, counts and every result are the same; only the order differs.
int cid_kmerset_set_target_index(cid_kmerset *ks, const cid_index *ix) {
    if (!ks || !ix) return fail(CID_ERR_INVALID, "null argument");
    if (ks->finalized || ks->n_raw || ks->n || ks->g_n) return fail(CID_ERR_STATE, "cid_kmerset_set_target_index comes before the first sequences");
    if (cid::index_k(ix) != ks->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", ks->k, cid::index_k(ix));
    if (ks->general) return CID_OK;   // byte-string sets keep their order (cid_kmerset_order_for_index groups them afterwards)
    if (getenv("CID_KMERSET_TARGET") && atoi(getenv("CID_KMERSET_TARGET")) == 0) return CID_OK;   // A/B: the code-ordered set
    const cid::ModMagic mm = cid::index_mod(ix);
    if (mm.m >= 0xFFFFFFFFull) return CID_OK;     // the key needs bloom_size < 2^32 - 1 to stay below kNoKey: such a set keeps code order
    ks->targeted = true;
    ks->key_for.mm = mm;
    ks->key_for.scale = 0xFFFFFFFF00000000ull / mm.m;
    return CID_OK;
}

int cid_kmerset_finalize(cid_kmerset *ks, uint64_t *n_distinct) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    if (ks->general) {
        if (ks->finalized) { if (n_distinct) *n_distinct = ks->n; return CID_OK; }
        const int rcg = finalize_general(ks);
        if (rcg) return rcg;
        if (ks->g_bases) { cid::ctx_free(ks->ctx, ks->g_bases); ks->g_bases = nullptr; ks->g_cap = ks->g_n = 0; }
        ks->g_segs.clear(); ks->g_segs.shrink_to_fit();
        ks->finalized = true;
        if (n_distinct) *n_distinct = ks->n;
        return CID_OK;
    }
    int rc = compact(ks);
    if (rc) return rc;
    if (ks->raw) { cid::ctx_free(ks->ctx, ks->raw); ks->raw = nullptr; ks->cap_raw = 0; }
    if (ks->raw_key) { cid::ctx_free(ks->ctx, ks->raw_key); ks->raw_key = nullptr; }
    ks->finalized = true;
    if (n_distinct) *n_distinct = ks->n;
    return CID_OK;
}

int cid_kmerset_size(const cid_kmerset *ks, uint64_t *n_distinct) {
    if (!ks || !n_distinct) return fail(CID_ERR_INVALID, "null argument");
    *n_distinct = ks->n;
    return CID_OK;
}

int cid_kmerset_count_histogram(const cid_kmerset *ks, uint32_t *values, uint64_t *n_kmers, size_t cap, size_t *n_bins) {
    if (!ks || !n_bins) return fail(CID_ERR_INVALID, "null argument");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    *n_bins = 0;
    if (ks->n == 0) return CID_OK;
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    DevBuf<uint32_t> sorted(ks->ctx), uniq(ks->ctx), runs(ks->ctx);
    DevBuf<uint64_t> d_count(ks->ctx);
    int rc;
    if ((rc = sorted.alloc(ks->n)) || (rc = uniq.alloc(ks->n)) || (rc = runs.alloc(ks->n)) || (rc = d_count.alloc(1))) return rc;
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, ks->counts, sorted.p, ks->n, 0u, 32u, st));
    DevBuf<uint8_t> tmp(ks->ctx);
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(rocprim::radix_sort_keys(tmp.p, tb, ks->counts, sorted.p, ks->n, 0u, 32u, st));
    size_t tb2 = 0;
    HIP_TRY(rocprim::run_length_encode(nullptr, tb2, sorted.p, ks->n, uniq.p, runs.p, d_count.p, st));
    DevBuf<uint8_t> tmp2(ks->ctx);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::run_length_encode(tmp2.p, tb2, sorted.p, ks->n, uniq.p, runs.p, d_count.p, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t nb = 0;
    HIP_TRY(hipMemcpy(&nb, d_count.p, 8, hipMemcpyDeviceToHost));
    *n_bins = nb;
    if (!values || !n_kmers) return CID_OK;   // size query
    if (cap < nb) return fail(CID_ERR_INVALID, "histogram needs %llu bins", (unsigned long long)nb);
    std::vector<uint32_t> r(nb);
    HIP_TRY(hipMemcpy(values, uniq.p, nb * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.data(), runs.p, nb * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < nb; ++i) n_kmers[i] = r[i];
    return CID_OK;
}

int cid_kmerset_clean(cid_kmerset *ks, uint64_t t) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->n == 0 || t == 0) return CID_OK;   // every stored k-mer has count >= 1 > 0
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    if (ks->general) {
        DevBuf<uint32_t> keep(ks->ctx), pos(ks->ctx), oc(ks->ctx);
        DevBuf<uint8_t> orows(ks->ctx), tmp(ks->ctx);
        int rc;
        if ((rc = keep.alloc(ks->n + 1)) || (rc = pos.alloc(ks->n + 1))) return rc;
        hipLaunchKernelGGL(cid::k_keep_gt, dim3(grid_for_n(ks->n + 1)), dim3(256), 0, st, ks->counts, t, keep.p, (uint64_t)ks->n);
        size_t tb = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, tb, keep.p, pos.p, 0u, ks->n + 1, rocprim::plus<uint32_t>(), st));
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, keep.p, pos.p, 0u, ks->n + 1, rocprim::plus<uint32_t>(), st));
        uint32_t kept = 0;
        HIP_TRY(hipMemcpyAsync(&kept, pos.p + ks->n, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if ((rc = orows.alloc((size_t)kept * ks->k)) || (rc = oc.alloc(kept))) return rc;
        hipLaunchKernelGGL(cid::k_compact_rows, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->counts, keep.p, pos.p, ks->k, orows.p, oc.p,
                           (uint64_t)ks->n);
        HIP_TRY(hipStreamSynchronize(st));
        cid::ctx_free(ks->ctx, ks->ascii); cid::ctx_free(ks->ctx, ks->counts);
        ks->ascii = orows.release(); ks->counts = oc.release(); ks->n = kept;
        return CID_OK;
    }
    DevBuf<uint8_t> flags(ks->ctx);
    DevBuf<uint64_t> oc(ks->ctx), d_count(ks->ctx);
    DevBuf<uint32_t> on(ks->ctx);
    int rc;
    if ((rc = flags.alloc(ks->n)) || (rc = oc.alloc(ks->n)) || (rc = on.alloc(ks->n)) || (rc = d_count.alloc(1))) return rc;
    hipLaunchKernelGGL(cid::k_flag_gt, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->counts, t, flags.p, (uint64_t)ks->n);
    size_t tb = 0;
    HIP_TRY(rocprim::select(nullptr, tb, ks->codes, flags.p, oc.p, d_count.p, ks->n, st));
    DevBuf<uint8_t> tmp(ks->ctx);
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(rocprim::select(tmp.p, tb, ks->codes, flags.p, oc.p, d_count.p, ks->n, st));
    size_t tb2 = 0;
    HIP_TRY(rocprim::select(nullptr, tb2, ks->counts, flags.p, on.p, d_count.p, ks->n, st));
    DevBuf<uint8_t> tmp2(ks->ctx);
    if ((rc = tmp2.alloc(tb2))) return rc;
    HIP_TRY(rocprim::select(tmp2.p, tb2, ks->counts, flags.p, on.p, d_count.p, ks->n, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t kept = 0;
    HIP_TRY(hipMemcpy(&kept, d_count.p, 8, hipMemcpyDeviceToHost));
    cid::ctx_free(ks->ctx, ks->codes); cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = oc.release(); ks->counts = on.release(); ks->n = kept;
    return CID_OK;
}

// Device arrays in, device arrays out (asynchronous on the ctx stream): the n k-mers (2-bit codes + multiplicities) grouped by the
// 128-byte index line of their first row (cid_ctx_tune "order_bits" > 0: by that many leading bits of its position instead).  The
// codes and the multiplicities each ride through their own stable radix sort on that key (same permutation), restricted to the
// key's significant bits: no index array, no random gather.
int cid_order_codes_for_index_dev(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, const uint32_t *d_counts, size_t n,
                                  uint64_t *d_codes_out, uint32_t *d_counts_out) {
    if (!c || !ix || (n && (!d_codes || !d_codes_out)) || (d_counts && !d_counts_out)) return fail(CID_ERR_INVALID, "null argument");
    if (n == 0) return CID_OK;
    if (n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mers");
    if (cid::index_k(ix) > 32) return fail(CID_ERR_UNSUPPORTED, "2-bit codes need k_size <= 32");
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    hipStream_t st = cid::ctx_stream(c);
    const uint32_t rs = cid::index_rs(ix);
    uint32_t line_shift = 0;
    while ((rs << line_shift) < 16) ++line_shift;   // rows per 128-byte line = 16 / rs
    const uint32_t bucket_bits = (uint32_t)cid::ctx_order_bits(c);
    const uint64_t max_key = bucket_bits ? ((1ull << bucket_bits) - 1) : ((cid::index_mod(ix).m - 1) >> line_shift);
    unsigned end_bit = 1;
    while (end_bit < 32 && (max_key >> end_bit)) ++end_bit;
    DevBuf<uint32_t> keys(c), keys2(c);
    int rc;
    if ((rc = keys.alloc(n)) || (rc = keys2.alloc(n))) return rc;
    hipLaunchKernelGGL(cid::k_row0_line, dim3(grid_for_n(n)), dim3(256), 0, st, d_codes, cid::index_k(ix), cid::index_mod(ix), line_shift, bucket_bits,
                       keys.p, (uint32_t *)nullptr, (uint64_t)n);
    size_t tb = 0, tb2 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, d_codes, d_codes_out, n, 0u, end_bit, st));
    if (d_counts) HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb2, keys.p, keys2.p, d_counts, d_counts_out, n, 0u, end_bit, st));
    DevBuf<uint8_t> tmp(c);
    if ((rc = tmp.alloc(tb > tb2 ? tb : tb2))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys.p, keys2.p, d_codes, d_codes_out, n, 0u, end_bit, st));
    if (d_counts) HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb2, keys.p, keys2.p, d_counts, d_counts_out, n, 0u, end_bit, st));
    return CID_OK;   // the scratch goes back to the ctx's block cache; later work on the same stream is ordered behind these kernels
}

int cid_kmerset_order_for_index(cid_kmerset *ks, const cid_index *ix) {
    if (!ks || !ix) return fail(CID_ERR_INVALID, "null argument");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->n == 0) return CID_OK;
    if (cid::index_k(ix) != ks->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", ks->k, cid::index_k(ix));
    if (ks->general) {   // byte strings: the same key from the ASCII k-mer, the rows permuted
        if (ks->n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mers");
        cid_ctx *c = ks->ctx;
        HIP_TRY(hipSetDevice(cid::ctx_device(c)));
        hipStream_t st = cid::ctx_stream(c);
        const uint32_t rs = cid::index_rs(ix);
        uint32_t line_shift = 0;
        while ((rs << line_shift) < 16) ++line_shift;
        const uint32_t bucket_bits = (uint32_t)cid::ctx_order_bits(c);
        const uint64_t max_key = bucket_bits ? ((1ull << bucket_bits) - 1) : ((cid::index_mod(ix).m - 1) >> line_shift);
        unsigned end_bit = 1;
        while (end_bit < 32 && (max_key >> end_bit)) ++end_bit;
        DevBuf<uint32_t> keys(c), keys2(c), idx(c), idx2(c), cnt(c);
        DevBuf<uint8_t> rows(c), tmp(c);
        int rc;
        if ((rc = keys.alloc(ks->n)) || (rc = keys2.alloc(ks->n)) || (rc = idx.alloc(ks->n)) || (rc = idx2.alloc(ks->n)) || (rc = cnt.alloc(ks->n)) ||
            (rc = rows.alloc(ks->n * ks->k)))
            return rc;
        hipLaunchKernelGGL(cid::k_row0_line_ascii, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->k, cid::index_mod(ix), line_shift, bucket_bits, keys.p,
                           idx.p, (uint64_t)ks->n);
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, idx.p, idx2.p, ks->n, 0u, end_bit, st));
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys.p, keys2.p, idx.p, idx2.p, ks->n, 0u, end_bit, st));
        hipLaunchKernelGGL(cid::k_permute_rows, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->ascii, ks->counts, idx2.p, ks->k, rows.p, cnt.p, (uint64_t)ks->n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        cid::ctx_free(c, ks->ascii); cid::ctx_free(c, ks->counts);
        ks->ascii = rows.release(); ks->counts = cnt.release();
        return CID_OK;
    }
    DevBuf<uint32_t> on(ks->ctx);
    DevBuf<uint64_t> oc(ks->ctx);
    int rc;
    if ((rc = oc.alloc(ks->n)) || (rc = on.alloc(ks->n))) return rc;
    if ((rc = cid_order_codes_for_index_dev(ks->ctx, ix, ks->codes, ks->counts, ks->n, oc.p, on.p))) return rc;
    HIP_TRY(hipStreamSynchronize(cid::ctx_stream(ks->ctx)));
    cid::ctx_free(ks->ctx, ks->codes); cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = oc.release(); ks->counts = on.release();
    return CID_OK;
}

int cid_kmerset_download(const cid_kmerset *ks, uint8_t *kmers_ascii, uint32_t *counts) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->n == 0) return CID_OK;
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    if (kmers_ascii && ks->general) {
        HIP_TRY(hipMemcpyAsync(kmers_ascii, ks->ascii, ks->n * ks->k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else if (kmers_ascii) {
        DevBuf<uint8_t> a(ks->ctx);
        int rc = a.alloc(ks->n * ks->k);
        if (rc) return rc;
        hipLaunchKernelGGL(cid::k_codes_to_ascii, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->codes, ks->k, a.p, (uint64_t)ks->n);
        HIP_TRY(hipMemcpyAsync(kmers_ascii, a.p, ks->n * ks->k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (counts) HIP_TRY(hipMemcpy(counts, ks->counts, ks->n * 4, hipMemcpyDeviceToHost));
    return CID_OK;
}

int cid_kmerset_device_arrays(const cid_kmerset *ks, void **d_codes, void **d_counts, uint64_t *n) {
    if (!ks || !d_codes || !d_counts || !n) return fail(CID_ERR_INVALID, "null argument");
    if (ks->general) return fail(CID_ERR_UNSUPPORTED, "a k_size > 32 set holds byte strings, not 2-bit codes: cid_kmerset_device_ascii");
    *d_codes = ks->codes; *d_counts = ks->counts; *n = ks->n;
    return CID_OK;
}

int cid_kmerset_device_ascii(const cid_kmerset *ks, void **d_kmers_ascii, void **d_counts, uint64_t *n) {
    if (!ks || !d_kmers_ascii || !d_counts || !n) return fail(CID_ERR_INVALID, "null argument");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (!ks->general) return fail(CID_ERR_UNSUPPORTED, "a k_size <= 32 set holds 2-bit codes: cid_kmerset_device_arrays");
    *d_kmers_ascii = ks->ascii; *d_counts = ks->counts; *n = ks->n;
    return CID_OK;
}

void cid_kmerset_destroy(cid_kmerset *ks) {
    if (!ks) return;
    (void)hipSetDevice(cid::ctx_device(ks->ctx));
    if (ks->raw) cid::ctx_free(ks->ctx, ks->raw);
    if (ks->raw_key) cid::ctx_free(ks->ctx, ks->raw_key);
    if (ks->codes) cid::ctx_free(ks->ctx, ks->codes);
    if (ks->counts) cid::ctx_free(ks->ctx, ks->counts);
    if (ks->g_bases) cid::ctx_free(ks->ctx, ks->g_bases);
    if (ks->ascii) cid::ctx_free(ks->ctx, ks->ascii);
    if (ks->d_flags) (void)hipFree(ks->d_flags);
    delete ks;
}

int cid_index_insert_kmerset(cid_index *ix, const cid_kmerset *ks, uint32_t colour) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->general) return cid::index_insert_ascii(ix, ks->ascii, ks->n, ks->k, colour);
    return cid::index_insert_codes(ix, ks->codes, ks->n, ks->k, colour);
}

int cid_search_count_set(cid_ctx *c, const cid_index *ix, const cid_kmerset *ks, uint64_t *hits, uint64_t *n_unique,
                         uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->general) return cid::search_count_ascii(c, ix, ks->ascii, ks->counts, ks->n, ks->k, hits, n_unique, sum_unique_freq, unique_colour);
    return cid::search_count_codes(c, ix, ks->codes, ks->counts, ks->n, ks->k, hits, n_unique, sum_unique_freq, unique_colour);
}


}  // extern "C"


extern "C" {

// a5 over a finalized set with everything reports::generate_report needs (reports.rs:8-48) and nothing per k-mer: hits, the number
// of unique-hit k-mers, the sum of their multiplicities (-> mean) and their mode, n_colors values each
int cid_search_count_set_report(cid_ctx *c, const cid_index *ix, const cid_kmerset *ks, uint64_t *hits, uint64_t *n_unique,
                                uint64_t *sum_unique_freq, uint64_t *mode_unique_freq) {
    if (!ks || !hits || !n_unique || !sum_unique_freq || !mode_unique_freq) return fail(CID_ERR_INVALID, "null argument");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    const uint32_t C = cid::index_n_colors(ix);
    DevBuf<uint32_t> uc(ks->ctx);
    DevBuf<uint64_t> out(ks->ctx);
    int rc;
    if ((rc = uc.alloc(ks->n)) || (rc = out.alloc((size_t)4 * C))) return rc;
    if (ks->general) rc = cid_search_count_dev(c, ix, ks->ascii, ks->counts, ks->n, out.p, out.p + C, out.p + 2 * C, uc.p);
    else rc = cid_search_count_codes_dev(c, ix, ks->codes, ks->counts, ks->n, out.p, out.p + C, out.p + 2 * C, uc.p);
    if (rc) return rc;
    if ((rc = cid::unique_freq_modes(c, uc.p, ks->counts, ks->n, C, out.p + 3 * C))) return rc;
    hipStream_t st = cid::ctx_stream(c);
    HIP_TRY(hipMemcpyAsync(hits, out.p, (size_t)C * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(n_unique, out.p + C, (size_t)C * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(sum_unique_freq, out.p + 2 * C, (size_t)C * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(mode_unique_freq, out.p + 3 * C, (size_t)C * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CID_OK;
}

int cid_search_perfect_set(cid_ctx *c, const cid_index *ix, const cid_kmerset *ks, uint32_t *and_words_le, int *any_row_missing) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->general) return cid::search_perfect_ascii(c, ix, ks->ascii, ks->n, ks->k, and_words_le, any_row_missing);
    return cid::search_perfect_codes(c, ix, ks->codes, ks->n, ks->k, and_words_le, any_row_missing);
}

}  // extern "C"
