// Device-side tails of the searches and of read_id that are NOT the k-mer set (cid_kmerset.hip) — a translation unit of their own, so
// that a command loads only the code object it uses (the k-mer set's, with its rocPRIM sorts, is 18 MB: 0.2 s of start-up):
//   * the mode of the unique-hit k-mer frequencies per colour (reports.rs:65-77) and its mergeable histogram form;
//   * rows of the index -> .bxi records (bigsi.rs / the build's save step);
//   * dense read_id report rows -> sparse (colour, count) entries per read.
#include <cstring>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"
#include "cid_scan.hpp"
#include "cid_devbuf.hpp"

// ------------------------------------------------------------------------------------------------ modes of the unique-hit frequencies
// reports.rs:65-77 on the vectors batch_search_pe.rs:75-82 fills: per colour, the most frequent multiplicity among the k-mers that
// hit exactly that colour (ties -> the smallest value; the reference's tie is HashMap order).  Done on the device so that neither
// the per-k-mer unique colours nor the multiplicities (4 + 4 bytes per k-mer) have to cross PCIe for the report.
//   small multiplicities (f < FL): a table of (colour, f) counts, privatised per workgroup in LDS (mode_cell), flushed with global atomics;
//   the rest: appended as (colour << 32 | f) keys, then sorted and run-length counted;
//   every (colour, f, count) cell proposes count << 32 | ~f to an atomicMax per colour: highest count wins, ties go to the smaller f.
namespace cid {

// Cell (c, f) lives at f * Cp + ((c + f) & (Cp - 1)), Cp = the power of two at or above max(C, 64): a wave's 64 atomics spread over the LDS
// banks when the colours differ (every multiplicity 1: reads of a metagenome) AND when the multiplicities differ (one colour: an isolate at
// coverage) — colour-major cells put the first case into ONE bank, 64 atomics one after the other (0.88 ms per 120 M k-mers; 0.3 since).
// A wave whose lanes all name the same cell adds its count once.
constexpr uint32_t kModeBlock = 512, kModePer = 4;
__device__ __forceinline__ uint32_t mode_cell(uint32_t c, uint32_t f, uint32_t cp_log) { return (f << cp_log) + ((c + f) & ((1u << cp_log) - 1u)); }

__global__ __launch_bounds__(kModeBlock) void k_mode_hist(const uint32_t *uc, const uint32_t *freq, uint64_t n, uint32_t cp_log, uint32_t FL, uint64_t per_block,
                                                          uint32_t *table, uint64_t *ovf, unsigned long long *ovf_count) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t *cells = reinterpret_cast<uint32_t *>(smem);
    const uint32_t n_cells = FL << cp_log;
    for (uint32_t i = threadIdx.x; i < n_cells; i += kModeBlock) cells[i] = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * per_block;
    const uint64_t i1 = i0 + per_block < n ? i0 + per_block : n;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = i0; base < i1; base += kModeBlock * kModePer) {   // block-uniform trip count: the ballots below see whole waves
        uint32_t c[kModePer], f[kModePer];
#pragma unroll
        for (uint32_t j = 0; j < kModePer; ++j) {   // the loads of the step leave together
            const uint64_t i = base + j * kModeBlock + threadIdx.x;
            c[j] = i < i1 ? uc[i] : 0xFFFFFFFFu;
            f[j] = (i < i1 && freq) ? freq[i] : 1u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kModePer; ++j) {
            const bool hit = c[j] != 0xFFFFFFFFu, small = hit && f[j] < FL, big = hit && !small;
            const uint32_t cell = small ? mode_cell(c[j], f[j], cp_log) : 0xFFFFFFFFu;
            const uint64_t ms = __ballot(small);
            if (ms) {
                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)cell, (int)__builtin_ctzll(ms));
                const uint64_t same = __ballot(small && cell == first);
                if (same == ms) {   // one cell for the whole wave
                    if (lane == (int)__builtin_ctzll(ms)) atomicAdd(&cells[first], (uint32_t)__popcll(ms));
                } else if (small) atomicAdd(&cells[cell], 1u);
            }
            const uint64_t m = __ballot(big);
            if (m) {
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd(ovf_count, (unsigned long long)__popcll(m));
                at = (unsigned long long)__shfl((long long)at, 0, 64);
                if (big) ovf[at + (uint64_t)__popcll(m & ((1ull << lane) - 1ull))] = ((uint64_t)c[j] << 32) | f[j];
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_cells; i += kModeBlock)
        if (cells[i]) atomicAdd(&table[i], cells[i]);
}
__global__ void k_mode_pick_table(const uint32_t *table, uint32_t C, uint32_t cp_log, uint32_t FL, unsigned long long *best) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (FL << cp_log)) return;
    const uint32_t v = table[i];
    if (!v) return;
    const uint32_t f = i >> cp_log, c = ((i & ((1u << cp_log) - 1u)) - f) & ((1u << cp_log) - 1u);
    if (c < C) atomicMax(&best[c], ((unsigned long long)v << 32) | (0xFFFFFFFFu - f));
}
__global__ void k_mode_pick_runs(const uint64_t *keys, const uint32_t *runs, const uint64_t *n_runs, unsigned long long *best) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_runs) return;
    const uint64_t k = keys[i];
    atomicMax(&best[k >> 32], ((unsigned long long)runs[i] << 32) | (0xFFFFFFFFu - (uint32_t)k));
}
__global__ void k_mode_final(const unsigned long long *best, uint32_t C, uint64_t *modes) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) modes[c] = best[c] ? (uint64_t)(0xFFFFFFFFu - (uint32_t)best[c]) : 0ull;
}

// begin: the table pass and the modes as they stand WITHOUT the multiplicities of FL and more (those are listed in w->ovf, their number
// lands in w->ovf_count[0]); finish: counts the listed ones in (a sort: rocPRIM) when there are any, and frees the work arrays.  Both are
// asynchronous: a caller fetches the number of listed entries together with its other results and waits ONCE when there are none.
int unique_freq_modes_begin(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, uint32_t C, uint64_t *d_modes, ModeWork *w) {
    hipStream_t st = ctx_stream(c);
    *w = ModeWork{};
    w->C = C;
    if (n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mers");
    uint32_t cp_log = 6;
    while (cp_log < 31 && (1u << cp_log) < C) ++cp_log;       // colours padded to a power of two (mode_cell)
    uint32_t FL = 64;
    while (FL > 1 && ((uint64_t)FL << cp_log) > 16384) FL >>= 1;   // the per-workgroup table: at most 64 KiB of LDS
    if (((uint64_t)FL << cp_log) > 16384) FL = 0;                  // very many colours: every (colour, f) goes through the sort
    const size_t n_cells = FL ? (size_t)FL << cp_log : 1;
    DevBuf<uint32_t> table(c);
    DevBuf<uint64_t> ovf(c);
    DevBuf<unsigned long long> best(c), ovf_count(c);
    int rc;
    if ((rc = table.alloc(n_cells)) || (rc = ovf.alloc(n ? n : 1)) || (rc = best.alloc(C)) || (rc = ovf_count.alloc(2))) return rc;
    HIP_TRY(hipMemsetAsync(table.p, 0, n_cells * 4, st));
    HIP_TRY(hipMemsetAsync(best.p, 0, (size_t)C * 8, st));
    HIP_TRY(hipMemsetAsync(ovf_count.p, 0, 16, st));
    if (n) {
        const unsigned blocks = 1024;
        constexpr uint64_t step = (uint64_t)kModeBlock * kModePer;
        uint64_t per_block = (n + blocks - 1) / blocks;
        per_block = (per_block + step - 1) / step * step;
        const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
        const size_t shmem = FL ? n_cells * 4 : 0;
        if (shmem > 64 * 1024) return fail(CID_ERR_UNSUPPORTED, "mode table");
        hipLaunchKernelGGL(k_mode_hist, dim3(grid), dim3(kModeBlock), shmem, st, d_uc, d_freq, n, cp_log, FL, per_block, table.p, ovf.p, ovf_count.p);
        if (FL) hipLaunchKernelGGL(k_mode_pick_table, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, table.p, C, cp_log, FL, best.p);
    }
    hipLaunchKernelGGL(k_mode_final, dim3((C + 255) / 256), dim3(256), 0, st, best.p, C, d_modes);
    HIP_TRY(hipGetLastError());
    w->ovf = ovf.release(); w->best = best.release(); w->ovf_count = ovf_count.release();
    return CID_OK;   // (the table goes back to the ctx's block cache: later work on the stream is ordered behind its readers)
}

int unique_freq_modes_finish(cid_ctx *c, ModeWork *w, unsigned long long n_ovf, uint64_t *d_modes) {
    hipStream_t st = ctx_stream(c);
    int rc = CID_OK;
    if (n_ovf) {
        DevBuf<uint32_t> runs(c);
        DevBuf<uint64_t> keys_sorted(c), keys_u(c), n_runs(c);
        if ((rc = keys_sorted.alloc(n_ovf)) || (rc = keys_u.alloc(n_ovf)) || (rc = runs.alloc(n_ovf)) || (rc = n_runs.alloc(1))) goto out;
        // (cid_kmerset_cold.hip: the rocPRIM unit is loaded only when a query has multiplicities beyond the table)
        if ((rc = cold_sort_keys_u64(c, st, w->ovf, keys_sorted.p, (size_t)n_ovf, 0u, 64u)) ||
            (rc = cold_run_length_u64(c, st, keys_sorted.p, (size_t)n_ovf, keys_u.p, runs.p, n_runs.p))) goto out;
        hipLaunchKernelGGL(k_mode_pick_runs, dim3(grid_for_n(n_ovf)), dim3(256), 0, st, keys_u.p, runs.p, n_runs.p, w->best);
        hipLaunchKernelGGL(k_mode_final, dim3((w->C + 255) / 256), dim3(256), 0, st, w->best, w->C, d_modes);
        if (hipGetLastError() != hipSuccess) rc = fail(CID_ERR_HIP, "mode kernels");
    }
out:
    if (w->ovf) ctx_free(c, w->ovf);
    if (w->best) ctx_free(c, w->best);
    if (w->ovf_count) ctx_free(c, w->ovf_count);
    *w = ModeWork{};
    return rc;
}

int unique_freq_modes(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, uint32_t C, uint64_t *d_modes) {
    ModeWork w;
    int rc = unique_freq_modes_begin(c, d_uc, d_freq, n, C, d_modes, &w);
    if (rc) { (void)unique_freq_modes_finish(c, &w, 0, d_modes); return rc; }
    unsigned long long n_ovf = 0;
    hipError_t e = hipMemcpyAsync(&n_ovf, w.ovf_count, 8, hipMemcpyDeviceToHost, ctx_stream(c));
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_stream(c));
    if (e != hipSuccess) {   // (w's arrays go back to the block cache on this way out too)
        (void)unique_freq_modes_finish(c, &w, 0, d_modes);
        return fail(CID_ERR_HIP, "unique_freq_modes: %s", hipGetErrorString(e));
    }
    return unique_freq_modes_finish(c, &w, n_ovf, d_modes);
}

}  // namespace cid

// ------------------------------------------------------------------------------------------------ rows -> .bxi records
// save_bigsi (bigsi.rs:51-57 / build.rs:123-127): the non-zero rows of a row range, in ascending order, as the bincode
// records the file holds — { u64 row ; u64 W32 ; W32 x u32 ; u64 n_colors } — formatted on the device so that the host only
// writes the bytes.
namespace cid {

__global__ void k_row_nonzero(const uint64_t *mat, uint32_t rs, uint64_t row_begin, uint64_t n, uint32_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { flags[i] = 0; return; }   // slot n receives the total
    const uint64_t *row = mat + (row_begin + i) * rs;
    uint64_t any = 0;
    for (uint32_t w = 0; w < rs; ++w) any |= row[w];
    flags[i] = any ? 1u : 0u;
}
__global__ void k_emit_records(const uint32_t *mat32, uint32_t rs, uint32_t w32, uint32_t n_colors, uint64_t row_begin, uint64_t n,
                               const uint32_t *flags, const uint32_t *pos, uint32_t *out32) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint64_t row = row_begin + i;
    uint32_t *rec = out32 + (uint64_t)pos[i] * (6ull + w32);
    rec[0] = (uint32_t)row; rec[1] = (uint32_t)(row >> 32);
    rec[2] = w32; rec[3] = 0;
    const uint32_t *src = mat32 + row * (2ull * rs);
    for (uint32_t w = 0; w < w32; ++w) rec[4 + w] = src[w];
    rec[4 + w32] = n_colors; rec[5 + w32] = 0;
}

// host buffer `records` holds up to n_rows records; *n_records = the number written
int index_get_records(cid_ctx *c, const cid_index *ix, uint64_t row_begin, uint64_t n_rows, uint8_t *records, uint64_t *n_records) {
    *n_records = 0;
    if (n_rows == 0) return CID_OK;
    if (n_rows >= (1ull << 32)) return fail(CID_ERR_INVALID, "at most 2^32-1 rows per call");
    hipStream_t st = ctx_stream(c);
    const uint32_t w32 = (index_n_colors(ix) + 31) / 32, rs = index_rs(ix);
    const size_t rec = 24 + 4ull * w32;
    DevBuf<uint32_t> flags(c), pos(c);
    DevBuf<uint8_t> out(c), tmp(c);
    int rc;
    if ((rc = flags.alloc(n_rows + 1)) || (rc = pos.alloc(n_rows + 1)) || (rc = out.alloc(n_rows * rec))) return rc;
    hipLaunchKernelGGL(k_row_nonzero, dim3(grid_for_n(n_rows + 1)), dim3(256), 0, st, index_matrix(ix), rs, row_begin, n_rows, flags.p);
    DevBuf<uint64_t> scan_state(c);
    if ((rc = scan_state.alloc(scan_state_words(n_rows + 1)))) return rc;
    HIP_TRY(scan_launch(ScanInU32{flags.p}, ScanOutU32{pos.p}, n_rows + 1, scan_state.p, st));
    hipLaunchKernelGGL(k_emit_records, dim3(grid_for_n(n_rows)), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(index_matrix(ix)), rs, w32,
                       index_n_colors(ix), row_begin, n_rows, flags.p, pos.p, reinterpret_cast<uint32_t *>(out.p));
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, pos.p + n_rows, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (total) HIP_TRY(hipMemcpy(records, out.p, (size_t)total * rec, hipMemcpyDeviceToHost));
    *n_records = total;
    return CID_OK;
}

}  // namespace cid

// ------------------------------------------------------------------------------------------------ sparse read_id reports
// A report row has n_colors+1 counters but only a handful are non-zero: compact the dense rows (left in HBM by
// k_readid) into per-read (colour, count) lists in ascending colour order, so that only those cross PCIe.
namespace cid {

__global__ __launch_bounds__(256) void k_row_nnz(const uint32_t *report, uint32_t width, uint64_t n_rows, uint32_t *nnz) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t *r = report + row * width;
    uint32_t n = 0;
    for (uint32_t c0 = 0; c0 < width; c0 += 64) {
        const uint32_t c = c0 + lane;
        n += (uint32_t)__popcll(__ballot(c < width && r[c] != 0));
    }
    if (lane == 0) nnz[row] = n;
}
__global__ __launch_bounds__(256) void k_row_compact(const uint32_t *report, uint32_t width, uint64_t n_rows, const uint64_t *row_start,
                                                     uint32_t *colours, uint32_t *counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t *r = report + row * width;
    uint64_t out = row_start[row];
    for (uint32_t c0 = 0; c0 < width; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < width ? r[c] : 0u;
        const uint64_t m = __ballot(v != 0);
        if (v) {
            const uint64_t o = out + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
            colours[o] = c;
            counts[o] = v;
        }
        out += (uint64_t)__popcll(m);
    }
}

// d_report: n_rows x width dense counters (device).  Leaves row_start (u64[n_rows+1]) and the entry arrays in the buffers
// it allocates; the caller owns (and frees) them.
int compact_report(cid_ctx *c, const uint32_t *d_report, uint32_t width, uint64_t n_rows, uint64_t **d_row_start, uint32_t **d_colours,
                   uint32_t **d_counts, uint64_t *n_entries) {
    hipStream_t st = ctx_stream(c);
    DevBuf<uint32_t> nnz(c), col(c), cnt(c);
    DevBuf<uint64_t> start(c);
    int rc;
    if ((rc = nnz.alloc(n_rows + 1)) || (rc = start.alloc(n_rows + 1))) return rc;
    HIP_TRY(hipMemsetAsync(nnz.p, 0, (n_rows + 1) * 4, st));
    if (n_rows) hipLaunchKernelGGL(k_row_nnz, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, d_report, width, n_rows, nnz.p);
    DevBuf<uint64_t> scan_state(c);
    if ((rc = scan_state.alloc(scan_state_words(n_rows + 1)))) return rc;
    HIP_TRY(scan_launch(ScanInU32{nnz.p}, ScanOutU64{start.p, 0ull}, n_rows + 1, scan_state.p, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t total = 0;
    HIP_TRY(hipMemcpy(&total, start.p + n_rows, 8, hipMemcpyDeviceToHost));
    if ((rc = col.alloc(total)) || (rc = cnt.alloc(total))) return rc;
    if (n_rows) hipLaunchKernelGGL(k_row_compact, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, d_report, width, n_rows, start.p, col.p, cnt.p);
    HIP_TRY(hipStreamSynchronize(st));
    *d_row_start = start.release(); *d_colours = col.release(); *d_counts = cnt.release(); *n_entries = total;
    return CID_OK;
}

hipError_t warm_reports() {   // see warm_readid (cid_readid.hip)
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_row_nnz));
}

}  // namespace cid

namespace cid {
__global__ void k_colour_freq_keys(const uint32_t *uc, const uint32_t *freq, uint64_t n, uint64_t *keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = uc[i];
    keys[i] = c == 0xFFFFFFFFu ? ~0ull : (((uint64_t)c << 32) | (freq ? freq[i] : 1u));
}
// The (colour, multiplicity) histogram of the k-mers that hit exactly one colour, as sorted keys colour << 32 | multiplicity with
// the number of k-mers each: what the mode of reports.rs:65-77 is taken from, in a form that ADDS over disjoint parts of a k-mer set
// (a mode does not) — cid_group_search_count_parts_report merges the ranks' histograms.  Host vectors; synchronous.
int unique_freq_hist(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, std::vector<uint64_t> &keys, std::vector<uint32_t> &counts) {
    keys.clear(); counts.clear();
    if (n == 0) return CID_OK;
    if (n >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 - 1 k-mers in one part");
    HIP_TRY(hipSetDevice(ctx_device(c)));
    hipStream_t st = ctx_stream(c);
    DevBuf<uint64_t> kin(c), kout(c), uniq(c), d_count(c);
    DevBuf<uint32_t> runs(c);
    int rc;
    if ((rc = kin.alloc(n)) || (rc = kout.alloc(n)) || (rc = uniq.alloc(n)) || (rc = runs.alloc(n)) || (rc = d_count.alloc(1))) return rc;
    hipLaunchKernelGGL(k_colour_freq_keys, dim3(grid_for_n(n)), dim3(256), 0, st, d_uc, d_freq, n, kin.p);
    if ((rc = cold_sort_keys_u64(c, st, kin.p, kout.p, n, 0u, 64u)) || (rc = cold_run_length_u64(c, st, kout.p, n, uniq.p, runs.p, d_count.p))) return rc;
    uint64_t nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, d_count.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    keys.resize(nb); counts.resize(nb);
    HIP_TRY(hipMemcpyAsync(keys.data(), uniq.p, nb * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(counts.data(), runs.p, nb * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (nb && keys.back() == ~0ull) { keys.pop_back(); counts.pop_back(); }   // the k-mers without a unique colour
    return CID_OK;
}
}  // namespace cid

extern "C" {

int cid_unique_freq_modes_dev(cid_ctx *c, const uint32_t *d_unique_colour, const uint32_t *d_freq, size_t n_kmers, uint32_t n_colors, uint64_t *d_modes) {
    if (!c || !d_modes || n_colors == 0 || (n_kmers && !d_unique_colour)) return fail(CID_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(cid::ctx_device(c)));
    return cid::unique_freq_modes(c, d_unique_colour, d_freq, n_kmers, n_colors, d_modes);
}

}  // extern "C"
