// The objects behind the ABI's opaque handles, shared by the translation units that implement them (cid_api_*.hip, cid_group*.hip).
#pragma once
#include <vector>

#include "cid_internal.hpp"

namespace cid {
namespace slots {
enum Slot { S_KMERS = 0, S_FREQ, S_OUT, S_UC, S_ROWIDS, S_WORDS, S_MISC, S_BASES, S_SEQOFF, S_READ0, S_REPORT, S_NK, S_ROUTE, S_REDO, S_QUEUE,
           S_ZSTART, S_COUNT };
}
}  // namespace cid

// Per-context tunables (cid_ctx_tune): measurement and test switches, read once from the environment when the ctx is made.
// Defaults are the shipped configuration; none of them changes a result.
struct cid_tunables {
    int search_unroll = 2;            // k_search_count on 64- / 128-byte rows: sub-passes whose row loads are issued together (2: -1.5 %)
    bool readid_packed_table = true;  // k_readid: 8-byte k-mer-set slots where they buy a sixth wave per SIMD (paired reads, k <= 31)
    int readid_blocks_per_cu = 64;    // k_readid: a batch is cut into about this many workgroups per CU (each takes a stretch of reads)
    int order_bits = 0;               // cid_kmerset_order_for_index: 0 = by the first row's 128-byte line, b = by its b leading bits
    long fastq_refuse_at_step = -1;   // tests: cid_fastq_classify_begin refuses its n-th step (as it does a stretch it cannot take); -1 = never
    long readid_long_from = -1;       // reads of at least this many bases take the long-read path (-1: the shipped rule, readid_route)
    bool readid_long_deal = true;     // long reads of four buckets and more: their windows dealt to the buckets once (false: every bucket re-reads the read)
    bool readid_long_lds = true;      // long reads: per-read sets by LDS hash tables (cid_readlong.hip); false = round 1's global radix sort
    bool readid_long_fuse = true;     // long reads of one hash table: codes + table in one kernel (k_long_fused); false = k_extract_codes + k_long_first_flags
    // the two measured-and-rejected schedulings of k_search_count; only a `make TUNE=1` build contains their kernels
    bool search_persist = false;      // persistent grid, one work queue per XCD
    bool search_mixed = false;        // 32-byte rows: each k-mer's last row through the scalar cache
};

struct cid_ctx {
    int device = 0;
    cid_tunables tune;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *slot[cid::slots::S_COUNT] = {};
    size_t slot_bytes[cid::slots::S_COUNT] = {};
    // result of the last cid_readid_count_sparse, fetched by cid_readid_sparse_fetch
    uint64_t *sp_start = nullptr; uint32_t *sp_col = nullptr, *sp_cnt = nullptr;
    uint64_t sp_rows = 0, sp_entries = 0;
    // scratch block cache (cid::ctx_alloc / ctx_free)
    struct Block { void *p; size_t bytes; bool used; };
    std::vector<Block> blocks;
    size_t idle_bytes = 0;
    // pinned host arena: the host-pointer read_id calls stage their (small) arrays through it — see cid::pin_reserve
    uint8_t *pin = nullptr;
    size_t pin_bytes = 0;
    // a cid_bgzf_inflate_start waiting for its _finish
    struct {
        size_t n_members = 0, text_bytes = 0, pin_text = 0, pin_status = 0;
        void *d_out = nullptr, *d_st = nullptr;
        bool open = false, staged = false;
    } inflate;
    // second stream + events for the host-pointer entry points: the H2D copy of chunk i+1 runs beside the kernel of chunk i
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    // the FASTQ front end's side streams (two for inflate launches, on a queue of the highest priority; text up; results down): made once
    // per context — a priority queue costs ~15 ms to create — by cid_warmup(CID_WARM_FASTQ) or the first cid_fastq_create
    hipStream_t side_streams[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct cid_index {
    cid_ctx *ctx = nullptr;
    uint64_t m = 0;
    uint32_t n_hash = 0, k = 0, n_colors = 0, w32 = 0, w64 = 0, rs = 0;
    uint32_t m_size = 0;  // > 0: minimizer (.mxi) index
    uint64_t *mat = nullptr;
    bool finalized = false;
    cid::ModMagic mod{};
};

namespace cid {
// grow-only per-role device buffers of a ctx
int slot_reserve(cid_ctx *c, int s, size_t bytes, void **out);
// A pinned host buffer of the ctx (grow-only).  A copy between device and the CALLER's pageable memory makes the runtime pin and
// unpin the caller's pages — a fixed cost per array of ~0.4 ms once other threads of the process fault pages at the same time (the
// CLI's inflating and packing threads: a read_id call of 50 000 reads took 5 ms instead of 1.1 ms).  Arrays of a few MB are cheaper
// copied through this arena.  NULL when the request is larger than kPinMax: the caller then lets the runtime handle its memory.
constexpr size_t kPinMax = 64u << 20;
uint8_t *pin_reserve(cid_ctx *c, size_t bytes, size_t cap = kPinMax);
// a5 launch on device-resident inputs/outputs (zeroes the counters first); asynchronous on the ctx stream
int search_count_launch(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint64_t *d_codes, const uint32_t *d_freq,
                        size_t n_kmers, uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq, uint32_t *d_unique_colour,
                        bool zero_counters = true);
// a4 launch: d_and (rs words) and d_missing (int) are preset here; asynchronous
int search_perfect_launch(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint64_t *d_codes, size_t n_kmers, uint64_t *d_and,
                          int *d_missing);
int search_count_host_input(cid_ctx *c, const cid_index *ix, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers, bool want_unique,
                            uint32_t *unique_colour, uint64_t **d_counters);
int check_ready(const cid_ctx *c, const cid_index *ix);
int index_put_records_slice(cid_index *ix, const uint8_t *records, size_t n_records, uint32_t n_colors_total, uint32_t colour_base);
int check_not_mini(const cid_index *ix);
}  // namespace cid
