// GPU k-mer counting for `search`: replaces the reference's String-keyed FnvHashMap counting
// (src/kmer.rs:87-125 kmerize_vector, :461-510 / :581-655 the fastq bodies, :826-837 clean_map) for k <= 32:
//   windows -> 2-bit canonical codes (one u64 each) -> radix sort -> run-length = (distinct k-mer, multiplicity).
// The set stays in HBM and feeds k_search_count / k_search_perfect directly (8 bytes per k-mer instead of k).
#include <algorithm>
#include <cstring>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"
#include "cid_objects.hpp"
#include "cid_partition.hpp"
#include "cid_rundedupe.hpp"
#include "cid_scan.hpp"
#include "cid_rle.hpp"
#include "cid_merge.hpp"
#include "cid_kmerset_obj.hpp"
#include "cid_devbuf.hpp"

namespace cid {

// windows of every sequence (0 when it is shorter than k): the input of the scan that places each read's codes
__global__ void k_seq_windows(const uint64_t *seq_off, uint64_t n_seqs, uint32_t k, uint64_t *n_win) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t len = seq_off[s + 1] - seq_off[s];
    n_win[s] = len >= k ? len - k + 1 : 0;
}
// histogram of the multiplicities (auto_cutoff's input, kmer.rs:866-942): values below kHistBins in LDS, the rest listed
constexpr uint32_t kHistBins = 4096;
__global__ __launch_bounds__(256) void k_count_hist(const uint32_t *counts, uint64_t n, uint64_t *bins, uint32_t *over, uint32_t cap_over, uint32_t *over_n) {
    __shared__ uint32_t s_bins[kHistBins];
    for (uint32_t b = threadIdx.x; b < kHistBins; b += blockDim.x) s_bins[b] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = counts[i];
        if (v < kHistBins) atomicAdd(&s_bins[v], 1u);
        else {
            const uint32_t at = atomicAdd(over_n, 1u);
            if (at < cap_over) over[at] = v;
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kHistBins; b += blockDim.x)
        if (s_bins[b]) atomicAdd(reinterpret_cast<unsigned long long *>(&bins[b]), (unsigned long long)s_bins[b]);
}
__global__ void k_codes_to_ascii(const uint64_t *codes, uint32_t k, uint8_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = codes[i];
    for (uint32_t t = 0; t < k; ++t) out[i * k + t] = (uint8_t)"ACGT"[(c >> (2 * (k - 1 - t))) & 3u];
}
}  // namespace cid


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
struct U32In {
    const uint32_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct U32Out {
    uint32_t *p;
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t) const { p[i] = (uint32_t)excl; }
};
struct U64In {
    const uint64_t *p;
    __device__ uint64_t operator()(uint64_t i) const { return p[i]; }
};
struct U64OutPlus {   // exclusive prefix + a start value, in place
    uint64_t *p;
    uint64_t init;
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t) const { p[i] = init + excl; }
};
struct KeepIn {
    const uint32_t *counts;
    uint64_t t;
    __device__ uint64_t operator()(uint64_t i) const { return counts[i] > t ? 1ull : 0ull; }
};
struct KeepOut {
    const uint64_t *codes;
    const uint32_t *counts;
    uint64_t *codes_out;
    uint32_t *counts_out;
    __device__ void operator()(uint64_t i, uint64_t excl, uint64_t keep) const {
        if (keep) { codes_out[excl] = codes[i]; counts_out[excl] = counts[i]; }
    }
};
// crowded runs (average comparisons per key in the bucket kernels beyond kCrowdedAt) go through k_run_dedupe_sort; CID_KMERSET_DEDUPE=0: straight
// to the radix kernel, at round 4's threshold of 32
int msd_sort(cid_ctx *c, hipStream_t st, uint64_t *a, uint64_t *b, size_t n, unsigned top, uint64_t **sorted, size_t *n_real) {
    using namespace cid;
    const cid_tunables &tn = c->tune;   // (the switches of this context: cid_switches.def)
    const bool kMsdSort = tn.kmerset_msd_sort, kDedupeSort = tn.kmerset_dedupe;
    const uint32_t kCrowdedAt = tn.kmerset_crowded_at >= 0 ? (uint32_t)tn.kmerset_crowded_at : (kDedupeSort ? 8u : cid::kBucketWork);
    // (CID_KMERSET_MSD_MIN: the tests send small inputs through the cold LSD sorts too.  Until round 4 batches below a million keys went
    // there by default — a dozen launches cost more than one rocPRIM sort — but the first rocPRIM call of a process loads a code object of
    // some thousand kernels, 0.2 s, which no small query earns back.)
    const size_t min_n = (size_t)tn.kmerset_msd_min;
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
    constexpr uint32_t kInfo = 8, kBigCap = 1024;          // info: [0] dropped (sentinels) [1] big runs [2] largest run [3] crowded runs (hard) [4] the radix kernel's runs (hard2) [5] runs sampled [6] of them crowded
    DevBuf<uint32_t> seg_a(c), seg_b(c), tile_base(c), table(c), info(c), hard(c), hard2(c);
    DevBuf<uint64_t> scan_state(c);
    DevBuf<PartTile> desc(c);
    int rc;
    if ((rc = desc.alloc((size_t)max_tiles + 1))) return rc;
    if ((rc = seg_a.alloc((size_t)n_runs + 1)) || (rc = seg_b.alloc((size_t)n_runs + 1)) || (rc = tile_base.alloc((size_t)S_last + 1)) ||
        (rc = table.alloc((size_t)max_tiles * kPartBins)) || (rc = info.alloc(kInfo + kBigCap)) || (rc = hard.alloc((size_t)n_runs + 1)) || (rc = hard2.alloc((size_t)n_runs + 1)))
        return rc;
    if ((rc = scan_state.alloc(scan_state_words((size_t)max_tiles * kPartBins)))) return rc;
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
        hipLaunchKernelGGL(k_part_tiles, dim3(1), dim3(kPartBlock), 0, st, seg, S, tile_base.p);
        hipLaunchKernelGGL(k_part_tile_desc, dim3((part_max_tiles((uint32_t)n, S) + 255) / 256), dim3(256), 0, st, seg, tile_base.p, S, bins, desc.p);
        hipLaunchKernelGGL(k_part_zero_tail, dim3(256), dim3(256), 0, st, table.p, tile_base.p, S, bins, (uint64_t)table_n);
        hipLaunchKernelGGL(k_part_hist, dim3(grid), dim3(kPartBlock), 0, st, (const PartTile *)desc.p, src, seg, tile_base.p, S, shift, bits, level_top, table.p, info.p);
        HIP_TRY(scan_launch(U32In{table.p}, U32Out{table.p}, table_n, scan_state.p, st));   // in place: a thread reads its elements before it writes them
        hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(kPartBlock), 0, st, (const PartTile *)desc.p, src, dst, seg, tile_base.p, S, shift, bits, level_top, table.p);
        hipLaunchKernelGGL(k_part_offsets, dim3((S * bins + 256) / 256), dim3(256), 0, st, table.p, tile_base.p, S, bits, (uint32_t)n, info.p, seg_next);
        HIP_TRY(hipGetLastError());
        std::swap(src, dst);
        std::swap(seg, seg_next);
        S *= bins;
        consumed += bits;
    }
    // src: partitioned, seg[0 .. n_runs]: the runs.  How large are they?
    hipLaunchKernelGGL(k_run_sizes, dim3((n_runs + 255) / 256), dim3(256), 0, st, seg, n_runs, 8192u, info.p + 1, info.p + kInfo, kBigCap, info.p + 2);
    const unsigned rest = top - consumed;                 // bits the runs still have to be sorted on
    if (kDedupeSort) hipLaunchKernelGGL((k_run_crowd_sample<false>), dim3(n_runs < kCrowdSample ? n_runs : kCrowdSample), dim3(kPartBlock), 0, st, nullptr, src, seg, n_runs,
                                        PairOrder{0u, 0u}, rest, kCrowdedAt, info.p + 5);
    uint32_t h_info[kInfo + kBigCap];
    HIP_TRY(hipMemcpyAsync(h_info, info.p, sizeof(h_info), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t dropped = h_info[0], n_big = h_info[1], largest = h_info[2];
    const bool crowded_batch = h_info[5] >= 8 && 2 * h_info[6] > h_info[5];
    const size_t kept = n - dropped;
    if (n_big > kBigCap) {   // badly skewed codes (low-complexity sequence): LSD radix sort of the partitioned array, all bits
        if ((rc = cold_sort_keys_u64(c, st, src, dst, kept, 0u, top))) return rc;
        *sorted = dst; *n_real = kept;
        return CID_OK;
    }
    // the bucket kernel takes every run it can (evenly spread keys), names the others in `hard`; the radix kernel sorts those
    // ... and k_run_dedupe_sort (cid_rundedupe.hpp) the crowded ones first: where the crowding is copies of the same k-mers (coverage) it
    // finishes them, the rest it names in hard2 for the radix kernel.  info[4] = runs in hard2.  A batch whose sampled runs are mostly
    // crowded (info[5], [6]: reads of an isolate) skips the bucket kernels: every run goes through k_run_dedupe_sort.
    const uint32_t *radix_list = hard.p, *radix_n = info.p + 3;
    const bool all_dedupe = kDedupeSort && crowded_batch;
    if (!all_dedupe) {
        hipLaunchKernelGGL(k_run_bucket_sort<8>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, info.p + 3, hard.p, kCrowdedAt);
        if (largest > 2048) hipLaunchKernelGGL(k_run_bucket_sort<16>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 2049u, info.p + 3, hard.p, kCrowdedAt);
    }
    if (kDedupeSort) {
        const uint32_t *list = all_dedupe ? nullptr : hard.p;
        hipLaunchKernelGGL((k_run_dedupe_sort<8, false>), dim3(grid), dim3(kPartBlock), 0, st, nullptr, src, dst, seg, n_runs, PairOrder{0u, 0u}, rest, 1u, 2048u,
                           largest <= 2048 ? 1u : 0u, list, info.p + 3, info.p + 4, hard2.p);
        if (largest > 2048) hipLaunchKernelGGL((k_run_dedupe_sort<15, false>), dim3(grid / 2), dim3(kPartBlock), 0, st, nullptr, src, dst, seg, n_runs, PairOrder{0u, 0u}, rest,
                                               2049u, 3840u, 1u, list, info.p + 3, info.p + 4, hard2.p);
        radix_list = hard2.p; radix_n = info.p + 4;
    }
    if (largest <= 2048) hipLaunchKernelGGL(k_run_sort<8>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 2048u, radix_list, radix_n);
    else if (largest <= 4096) hipLaunchKernelGGL(k_run_sort<16>, dim3(grid), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 4096u, radix_list, radix_n);
    else hipLaunchKernelGGL(k_run_sort<32>, dim3(grid / 4), dim3(kPartBlock), 0, st, src, dst, seg, n_runs, rest, 1u, 8192u, radix_list, radix_n);
    HIP_TRY(hipGetLastError());
    if (n_big) {   // runs beyond 8192 keys: one LSD radix sort each, on the bits they still differ in
        std::vector<uint32_t> h_seg(2 * (size_t)n_big);
        for (uint32_t i = 0; i < n_big; ++i) HIP_TRY(hipMemcpyAsync(&h_seg[2 * i], seg + h_info[kInfo + i], 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t i = 0; i < n_big; ++i) {
            const uint32_t lo = h_seg[2 * i], sz = h_seg[2 * i + 1] - lo;
            if ((rc = cold_sort_keys_u64(c, st, src + lo, dst + lo, sz, 0u, rest ? rest : 1u))) return rc;
        }
    }
    *sorted = dst; *n_real = kept;
    return CID_OK;
}

// The same sort for a set built FOR an index: (key, code) pairs, partitioned on the key's leading bits, the runs finished in LDS on
// (rest of the key, code).  keys_a / codes_a hold the n windows (keys of kNoKey = no k-mer) and are overwritten; keys_b / codes_b
// are scratch.  On return *sorted (codes_a or codes_b) holds the *n_real real codes in (key, code) order.  The keys are evenly
// spread by construction (row0_key), so the runs all have about n >> prefix members.
int msd_sort_pair(cid_ctx *c, hipStream_t st, uint32_t *keys_a, uint64_t *codes_a, uint32_t *keys_b, uint64_t *codes_b, size_t n, unsigned code_bits,
                  uint64_t **sorted, size_t *n_real) {
    using namespace cid;
    const cid_tunables &tn = c->tune;   // (the switches of this context: cid_switches.def)
    const bool kMsdSort = tn.kmerset_msd_sort, kDedupeSort = tn.kmerset_dedupe;
    const uint32_t kCrowdedAt = tn.kmerset_crowded_at >= 0 ? (uint32_t)tn.kmerset_crowded_at : (kDedupeSort ? 8u : cid::kBucketWork);
    const size_t min_n = (size_t)tn.kmerset_msd_min;
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
    constexpr uint32_t kInfo = 8, kBigCap = 1024;          // info: [0] dropped (no k-mer) [1] big runs [2] largest run [3] crowded runs (hard) [4] the radix kernel's runs (hard2) [5] runs sampled [6] of them crowded
    DevBuf<uint32_t> seg_a(c), seg_b(c), tile_base(c), table(c), info(c), hard(c), hard2(c);
    DevBuf<uint64_t> scan_state(c);
    DevBuf<PartTile> desc(c);
    int rc;
    if ((rc = desc.alloc((size_t)max_tiles + 1))) return rc;
    if ((rc = seg_a.alloc((size_t)n_runs + 1)) || (rc = seg_b.alloc((size_t)n_runs + 1)) || (rc = tile_base.alloc((size_t)S_last + 1)) ||
        (rc = table.alloc((size_t)max_tiles * kPartBins)) || (rc = info.alloc(kInfo + kBigCap)) || (rc = hard.alloc((size_t)n_runs + 1)) || (rc = hard2.alloc((size_t)n_runs + 1)))
        return rc;
    if ((rc = scan_state.alloc(scan_state_words((size_t)max_tiles * kPartBins)))) return rc;
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
        hipLaunchKernelGGL(k_part_tiles, dim3(1), dim3(kPartBlock), 0, st, seg, S, tile_base.p);
        hipLaunchKernelGGL(k_part_tile_desc, dim3((part_max_tiles((uint32_t)n, S) + 255) / 256), dim3(256), 0, st, seg, tile_base.p, S, bins, desc.p);
        hipLaunchKernelGGL(k_part_zero_tail, dim3(256), dim3(256), 0, st, table.p, tile_base.p, S, bins, (uint64_t)table_n);
        hipLaunchKernelGGL(k_part_hist_key, dim3(grid), dim3(kPartBlock), 0, st, (const PartTile *)desc.p, ksrc, seg, tile_base.p, S, shift, bits, l == 0 ? 1u : 0u, table.p, info.p);
        HIP_TRY(scan_launch(U32In{table.p}, U32Out{table.p}, table_n, scan_state.p, st));   // in place: a thread reads its elements before it writes them
        hipLaunchKernelGGL(k_part_scatter_pair, dim3(grid), dim3(kPartBlock), 0, st, (const PartTile *)desc.p, ksrc, src, kdst, dst, seg, tile_base.p, S, shift, bits, l == 0 ? 1u : 0u,
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
    const PairOrder ord{32u - consumed, code_bits};
    if (kDedupeSort) hipLaunchKernelGGL((k_run_crowd_sample<true>), dim3(n_runs < kCrowdSample ? n_runs : kCrowdSample), dim3(kPartBlock), 0, st, ksrc, src, seg, n_runs, ord, 0u,
                                        kCrowdedAt, info.p + 5);
    uint32_t h_info[kInfo + kBigCap];
    HIP_TRY(hipMemcpyAsync(h_info, info.p, sizeof(h_info), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t dropped = h_info[0], n_big = h_info[1], largest = h_info[2];
    const bool crowded_batch = h_info[5] >= 8 && 2 * h_info[6] > h_info[5];
    const size_t kept = n - dropped;
    // two stable LSD sorts = the (key, code) order, for what the LDS kernels do not take: first by code, then by the key's rest
    auto lsd_pair = [&](size_t lo, size_t sz) -> int {
        int rc2;
        if ((rc2 = cold_sort_pairs_u64_u32(c, st, src + lo, dst + lo, ksrc + lo, kdst + lo, sz, 0u, code_bits))) return rc2;
        if ((rc2 = cold_sort_pairs_u32_u64(c, st, kdst + lo, ksrc + lo, dst + lo, src + lo, sz, 0u, ord.kbits ? ord.kbits : 1u))) return rc2;
        HIP_TRY(hipMemcpyAsync(dst + lo, src + lo, sz * 8, hipMemcpyDeviceToDevice, st));   // the result belongs in dst, like the LDS kernels'
        return CID_OK;
    };
    if (n_big > kBigCap) {   // one row taking a large share of the windows (low-complexity sequence): everything through the LSD sorts
        if ((rc = cold_sort_pairs_u64_u32(c, st, src, dst, ksrc, kdst, kept, 0u, code_bits))) return rc;
        if ((rc = cold_sort_pairs_u32_u64(c, st, kdst, ksrc, dst, src, kept, 0u, 32u))) return rc;
        *sorted = src; *n_real = kept;
        return CID_OK;
    }
    const uint32_t *radix_list = hard.p, *radix_n = info.p + 3;
    const bool all_dedupe = kDedupeSort && crowded_batch;   // (see msd_sort; a pair's table slot is 16 bytes: 3072 pairs at three workgroups per CU)
    if (!all_dedupe) {
        hipLaunchKernelGGL(k_run_bucket_sort_pair<8>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, info.p + 3, hard.p, kCrowdedAt);
        if (largest > 2048) hipLaunchKernelGGL(k_run_bucket_sort_pair<16>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 2049u, info.p + 3, hard.p, kCrowdedAt);
    }
    if (kDedupeSort) {
        const uint32_t *list = all_dedupe ? nullptr : hard.p;
        hipLaunchKernelGGL((k_run_dedupe_sort<8, true>), dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 0u, 1u, 2048u, largest <= 2048 ? 1u : 0u, list,
                           info.p + 3, info.p + 4, hard2.p);
        if (largest > 2048) hipLaunchKernelGGL((k_run_dedupe_sort<12, true>), dim3(grid / 2), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 0u, 2049u, 3072u, 1u, list,
                                               info.p + 3, info.p + 4, hard2.p);
        radix_list = hard2.p; radix_n = info.p + 4;
    }
    if (largest <= 2048) hipLaunchKernelGGL(k_run_sort_pair<8>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 2048u, radix_list, radix_n);
    else if (largest <= 4096) hipLaunchKernelGGL(k_run_sort_pair<16>, dim3(grid), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 4096u, radix_list, radix_n);
    else hipLaunchKernelGGL(k_run_sort_pair<32>, dim3(grid / 4), dim3(kPartBlock), 0, st, ksrc, src, dst, seg, n_runs, ord, 1u, 8192u, radix_list, radix_n);
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
// sentinel dropped; cid_rle.hpp); a set that already holds k-mers is then MERGED with it — two sorted lists, one pass, equal
// neighbours added (kmerset_merge_batch) — instead of re-sorting everything it holds with every batch
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
                if ((rc = cid::cold_sort_pairs_u64_u32(ks->ctx, st, ks->raw, sorted.p, ks->raw_key, key_b.p, batch, 0u, ks->end_bit))) return rc;
                if ((rc = cid::cold_sort_pairs_u32_u64(ks->ctx, st, key_b.p, ks->raw_key, sorted.p, ks->raw, batch, 0u, 32u))) return rc;
                in_order = ks->raw;
            }
        } else {
        // k <= 31: every real code is below bit end_bit - 1, the sentinel is that bit — the MSD sort drops it (ks->raw is overwritten)
        rc = ks->k <= 31 ? msd_sort(ks->ctx, st, ks->raw, sorted.p, batch, ks->end_bit - 1, &msd_out, &n_sorted) : CID_ERR_UNSUPPORTED;
        if (rc == CID_OK) in_order = msd_out;
        else if (rc != CID_ERR_UNSUPPORTED) return rc;
        else {
            n_sorted = batch;
            if ((rc = cid::cold_sort_keys_u64(ks->ctx, st, ks->raw, sorted.p, batch, 0u, ks->end_bit))) return rc;
        }
        }
        if (n_sorted == 0) {   // nothing but invalid windows
            HIP_TRY(hipMemsetAsync(d_count.p, 0, 8, st));
        } else {
            DevBuf<uint64_t> rle_state(ks->ctx);
            DevBuf<uint32_t> rle_tiles(ks->ctx);
            if ((rc = rle_state.alloc(cid::rle_tiles(n_sorted) + 2)) || (rc = rle_tiles.alloc(3 * cid::rle_tiles(n_sorted)))) return rc;
            HIP_TRY(cid::rle_launch(in_order, (uint32_t)n_sorted, uniq.p, agg.p, rle_state.p, rle_tiles.p, d_count.p, st));
        }
        HIP_TRY(hipMemcpyAsync(&n_runs, d_count.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n_runs > 0 && in_order != msd_out) {  // the sentinel (invalid windows) sorts last: drop it (the MSD sort has done so already)
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
    // the batch joins the set: two sorted lists merged in one pass, equal k-mers' multiplicities added (cid_merge.hpp; rocPRIM's merge +
    // reduce_by_key until round 5 — kmerset_merge_batch in cid_kmerset_cold.hip, kept as CID_KMERSET_COLD_MERGE=1 for A/B runs)
    if (ks->ctx->tune.kmerset_cold_merge) return cid::kmerset_merge_batch(ks, uniq.p, agg.p, n_runs);
    const size_t total = ks->n + n_runs;
    const uint32_t tiles = cid::merge_tiles(ks->n, n_runs);
    DevBuf<uint64_t> ok(ks->ctx), split(ks->ctx), mstate(ks->ctx);
    DevBuf<uint32_t> ov(ks->ctx), ka(ks->ctx), kb(ks->ctx);
    if ((rc = ok.alloc(total)) || (rc = ov.alloc(total)) || (rc = split.alloc(2 * ((size_t)tiles + 1))) || (rc = mstate.alloc((size_t)tiles + 2))) return rc;
    if (ks->targeted) {   // both lists are in (row0_key, code) order: merged on that pair (the keys are recomputed from the codes, not kept)
        if ((rc = ka.alloc(ks->n)) || (rc = kb.alloc(n_runs))) return rc;
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(ks->n)), dim3(256), 0, st, ks->codes, ks->k, ks->key_for, ka.p, (uint64_t)ks->n);
        hipLaunchKernelGGL(cid::k_row0_keys, dim3(grid_for_n(n_runs)), dim3(256), 0, st, uniq.p, ks->k, ks->key_for, kb.p, (uint64_t)n_runs);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(ks->d_flags + 1, 0, 4, st));
    HIP_TRY(cid::merge_launch(ks->targeted ? ka.p : nullptr, ks->codes, ks->counts, ks->n, ks->targeted ? kb.p : nullptr, uniq.p, agg.p, n_runs, ok.p, ov.p, split.p,
                              mstate.p, ks->d_flags + 1, st));
    int sat = 0;
    uint64_t n_merged = 0;
    HIP_TRY(hipMemcpyAsync(&sat, ks->d_flags + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_merged, mstate.p + tiles + 1, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // multiplicities are u32 (the reference: usize): a sum that saturated cannot be reported faithfully
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

namespace cid {

hipError_t warm_kmerset() {   // see warm_readid (cid_readid.hip): this file's kernels serve k-mer counting and the sort-based read_id path
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_seq_windows));
}

}  // namespace cid

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
                           win_off, base0, ks->raw_key, ks->key_for, (const uint32_t *)nullptr, (uint8_t *)nullptr);
    else
        hipLaunchKernelGGL(cid::k_extract_codes<false>, dim3(grid), dim3(256), shmem, st, bases, segs, n_segs, ks->k, mode, ks->sentinel, ks->raw, ks->d_flags, seq_off,
                           win_off, base0, (uint32_t *)nullptr, cid::KeyFor{}, (const uint32_t *)nullptr, (uint8_t *)nullptr);
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
    if (c->tune.kmerset_compact_at) ks->compact_at = (size_t)c->tune.kmerset_compact_at;  // tests (CID_KMERSET_COMPACT_WINDOWS)
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
        DevBuf<uint64_t> scan_state(c);
        if ((rc = scan_state.alloc(cid::scan_state_words(n_seqs)))) return rc;
        HIP_TRY(cid::scan_launch(U64In{d_win.p}, U64OutPlus{d_win.p, (uint64_t)ks->n_raw}, n_seqs, scan_state.p, st));   // where each read's codes go
        const bool piped = cid::ctx_copy_stream(c) != nullptr;   // (also beside a borrowed stream: the scratch is the ctx's own, events order the two streams)
        hipStream_t cs = piped ? cid::ctx_copy_stream(c) : st;
        hipEvent_t ev_scan = cid::ctx_event(c, 0), ev_done = cid::ctx_event(c, 1);
        if (piped) { HIP_TRY(hipEventRecord(ev_scan, st)); HIP_TRY(hipStreamWaitEvent(cs, ev_scan, 0)); }   // (d_bases may still be read by an earlier kernel of the ctx stream)
        const uint64_t slice_bytes = (uint64_t)(c->tune.kmerset_slice_mb > 0 ? c->tune.kmerset_slice_mb : 32) << 20;   // (experiments: CID_KMERSET_SLICE_MB)
        // (A ring of pinned slots filled by 2-8 helper threads, the DMA engine taking a slot as soon as it is full, was tried in round 5: 4.6-4.9 ms
        // for the 150 MB of a million reads against 3.85 ms for these copies straight out of the caller's pageable memory — the host's memcpy
        // is the slower pipe; profiles/HISTORY.md.)
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
    DevBuf<uint64_t> scan_state(c);
    if ((rc = scan_state.alloc(cid::scan_state_words(n_seqs + 1)))) return rc;
    HIP_TRY(cid::scan_launch(U64In{d_win.p}, U64OutPlus{d_win.p, (uint64_t)ks->n_raw}, n_seqs + 1, scan_state.p, st));
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
    if (!ks->ctx->tune.kmerset_target) return CID_OK;   // A/B: the code-ordered set (CID_KMERSET_TARGET=0)
    const cid::ModMagic mm = cid::index_mod(ix);
    if (mm.m >= 0xFFFFFFFFull) return CID_OK;     // the key needs bloom_size < 2^32 - 1 to stay below kNoKey: such a set keeps code order
    // a small index: the keys take only bloom_size values, the partition's runs collapse to a few crowded ones (one LSD sort each, cold path),
    // and an index of a megabyte sits in L2 whatever the order of the queries — such a set keeps code order too
    if (mm.m < (1ull << 20) && !ks->ctx->tune.kmerset_target_small) return CID_OK;
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
        const int rcg = cid::kmerset_finalize_general(ks);
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
    // multiplicities below kHistBins are counted in LDS, per workgroup, and flushed; the few beyond are listed and counted on the host
    DevBuf<uint64_t> bins(ks->ctx);
    DevBuf<uint32_t> over(ks->ctx), over_n(ks->ctx);
    int rc;
    size_t cap_over = ks->n < (1u << 20) ? ks->n : (size_t)1 << 20;
    if ((rc = bins.alloc(cid::kHistBins)) || (rc = over_n.alloc(1))) return rc;
    std::vector<uint64_t> h_bins(cid::kHistBins);
    std::vector<uint32_t> h_over;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = over.alloc(cap_over))) return rc;
        HIP_TRY(hipMemsetAsync(bins.p, 0, cid::kHistBins * 8, st));
        HIP_TRY(hipMemsetAsync(over_n.p, 0, 4, st));
        hipLaunchKernelGGL(cid::k_count_hist, dim3((unsigned)cid::ctx_n_cu(ks->ctx) * 4u), dim3(256), 0, st, ks->counts, (uint64_t)ks->n, bins.p, over.p, (uint32_t)cap_over,
                           over_n.p);
        HIP_TRY(hipGetLastError());
        uint32_t n_over = 0;
        HIP_TRY(hipMemcpyAsync(h_bins.data(), bins.p, cid::kHistBins * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&n_over, over_n.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n_over > cap_over) {   // (only a set of mostly very frequent k-mers: once more with room for all of them)
            cid::ctx_free(ks->ctx, over.release());
            cap_over = ks->n;
            continue;
        }
        h_over.resize(n_over);
        if (n_over) HIP_TRY(hipMemcpy(h_over.data(), over.p, (size_t)n_over * 4, hipMemcpyDeviceToHost));
        break;
    }
    std::sort(h_over.begin(), h_over.end());
    std::vector<uint32_t> vals;
    std::vector<uint64_t> cnts;
    for (uint32_t v = 0; v < cid::kHistBins; ++v)
        if (h_bins[v]) { vals.push_back(v); cnts.push_back(h_bins[v]); }
    for (size_t i = 0; i < h_over.size();) {
        size_t j = i;
        while (j < h_over.size() && h_over[j] == h_over[i]) ++j;
        vals.push_back(h_over[i]); cnts.push_back(j - i);
        i = j;
    }
    const uint64_t nb = vals.size();
    *n_bins = nb;
    if (!values || !n_kmers) return CID_OK;   // size query
    if (cap < nb) return fail(CID_ERR_INVALID, "histogram needs %llu bins", (unsigned long long)nb);
    for (uint64_t i = 0; i < nb; ++i) { values[i] = vals[i]; n_kmers[i] = cnts[i]; }
    return CID_OK;
}

int cid_kmerset_clean(cid_kmerset *ks, uint64_t t) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->n == 0 || t == 0) return CID_OK;   // every stored k-mer has count >= 1 > 0
    HIP_TRY(hipSetDevice(cid::ctx_device(ks->ctx)));
    hipStream_t st = cid::ctx_stream(ks->ctx);
    if (ks->general) return cid::kmerset_clean_general(ks, t);
    DevBuf<uint64_t> oc(ks->ctx), scan_state(ks->ctx);
    DevBuf<uint32_t> on(ks->ctx);
    int rc;
    if ((rc = oc.alloc(ks->n)) || (rc = on.alloc(ks->n)) || (rc = scan_state.alloc(cid::scan_state_words(ks->n)))) return rc;
    // one pass: a k-mer counted more than t times learns its place from the scan and moves there with its count (clean_map, kmer.rs:826-837)
    HIP_TRY(cid::scan_launch(KeepIn{ks->counts, t}, KeepOut{ks->codes, ks->counts, oc.p, on.p}, ks->n, scan_state.p, st));
    uint64_t kept = 0;
    HIP_TRY(hipMemcpyAsync(&kept, scan_state.p + cid::scan_tiles(ks->n) + 1, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    cid::ctx_free(ks->ctx, ks->codes); cid::ctx_free(ks->ctx, ks->counts);
    ks->codes = oc.release(); ks->counts = on.release(); ks->n = kept;
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
    if ((rc = uc.alloc(ks->n)) || (rc = out.alloc((size_t)4 * C + 2))) return rc;
    if (ks->general) rc = cid_search_count_dev(c, ix, ks->ascii, ks->counts, ks->n, out.p, out.p + C, out.p + 2 * C, uc.p);
    else rc = cid_search_count_codes_dev(c, ix, ks->codes, ks->counts, ks->n, out.p, out.p + C, out.p + 2 * C, uc.p);
    if (rc) return rc;
    // ONE copy, ONE wait: the three counter arrays, the modes as they stand and the number of multiplicities the mode table did not hold
    // come back together; only when there are such k-mers (deep coverage) a second step counts them in.  (Round 4: two waits, four copies.)
    cid::ModeWork w;
    struct ModeGuard {   // whatever way this function is left, the mode table's arrays go back to the ctx's block cache
        cid_ctx *c; cid::ModeWork *w; uint64_t *modes; bool done = false;
        ~ModeGuard() { if (!done) (void)cid::unique_freq_modes_finish(c, w, 0, modes); }
    } guard{c, &w, out.p + 3 * C};
    if ((rc = cid::unique_freq_modes_begin(c, uc.p, ks->counts, ks->n, C, out.p + 3 * C, &w))) return rc;
    hipStream_t st = cid::ctx_stream(c);
    std::vector<uint64_t> h((size_t)4 * C + 1);
    HIP_TRY(hipMemcpyAsync(out.p + 4 * C, w.ovf_count, 8, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h.data(), out.p, ((size_t)4 * C + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long n_ovf = h[(size_t)4 * C];
    guard.done = true;
    if ((rc = cid::unique_freq_modes_finish(c, &w, n_ovf, out.p + 3 * C))) return rc;
    if (n_ovf) {
        HIP_TRY(hipMemcpyAsync(h.data() + 3 * (size_t)C, out.p + 3 * C, (size_t)C * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    memcpy(hits, h.data(), (size_t)C * 8);
    memcpy(n_unique, h.data() + C, (size_t)C * 8);
    memcpy(sum_unique_freq, h.data() + 2 * (size_t)C, (size_t)C * 8);
    memcpy(mode_unique_freq, h.data() + 3 * (size_t)C, (size_t)C * 8);
    return CID_OK;
}

int cid_search_perfect_set(cid_ctx *c, const cid_index *ix, const cid_kmerset *ks, uint32_t *and_words_le, int *any_row_missing) {
    if (!ks) return fail(CID_ERR_INVALID, "null set");
    if (!ks->finalized) return fail(CID_ERR_STATE, "k-mer set not finalized");
    if (ks->general) return cid::search_perfect_ascii(c, ix, ks->ascii, ks->n, ks->k, and_words_le, any_row_missing);
    return cid::search_perfect_codes(c, ix, ks->codes, ks->n, ks->k, and_words_le, any_row_missing);
}

}  // extern "C"

