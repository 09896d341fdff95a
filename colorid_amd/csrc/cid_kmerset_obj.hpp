// The object behind the ABI's cid_kmerset handle and the entry points that connect its two translation units: cid_kmerset.hip — the
// steady-state path (window codes, the set's own MSD sort, run-length count: this repository's kernels only) — and
// cid_kmerset_cold.hip — what a query meets rarely or never (byte-string sets, badly skewed inputs, the A/B copy of the batch merge,
// reordering after the fact, round 1's long-read sort), built on rocPRIM, whose code object of some thousand kernels is loaded only
// when one of these is called.
#pragma once
#include <vector>

#include "cid_internal.hpp"
#include "cid_windows.hpp"

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

namespace cid {
// ---- cold paths (cid_kmerset_cold.hip).  All run on `st`; scratch comes from the ctx's block cache.
// (cold_sort_* / cold_run_length_u64: cid_internal.hpp)
// a finalized-so-far set (ks->codes / counts, ks->n > 0) merged with a sorted, run-length counted batch (uniq / agg, n_runs > 0; the
// caller keeps owning them): two sorted lists, one pass, equal neighbours added; waits for the stream
int kmerset_merge_batch(cid_kmerset *ks, const uint64_t *uniq, const uint32_t *agg, uint64_t n_runs);
int kmerset_finalize_general(cid_kmerset *ks);          // k > 32: all windows of the resident sequences -> distinct byte strings + counts
int kmerset_clean_general(cid_kmerset *ks, uint64_t t);  // k > 32: keep the k-mers counted more than t times
}  // namespace cid
