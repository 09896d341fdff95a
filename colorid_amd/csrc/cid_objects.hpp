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

// Per-context tunables (cid_ctx_tune): measurement and test switches, read ONCE from the environment when the ctx is made
// (cid_api_ctx.hip: read_switches — the only place of the library that calls getenv).  The list, with what each does: cid_switches.def.
// Defaults are the shipped configuration; none of them changes a result.
struct cid_tunables {
#define CID_SWITCH_B(name, dflt) bool name = dflt != 0;
#define CID_SWITCH_I(name, dflt) int name = dflt;
#define CID_SWITCH_L(name, dflt) long name = dflt;
#define CID_SWITCH_U(name, dflt) unsigned long long name = dflt##ull;
#define CID_SWITCH_S(name, dflt)
#define CID_SWITCH(name, env, kind, dflt, doc) CID_SWITCH_##kind(name, dflt)
#include "cid_switches.def"
#undef CID_SWITCH
#undef CID_SWITCH_B
#undef CID_SWITCH_I
#undef CID_SWITCH_L
#undef CID_SWITCH_U
#undef CID_SWITCH_S
    int reduce_mode = -1;             // COLORID_REDUCE: -1 auto, 0 host, 1 rccl
    bool stripe_reduce_peer = false;  // COLORID_STRIPE_REDUCE=peer
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
