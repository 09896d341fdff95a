// C ABI of libcolorid_hip.so, part 4: per-read classification counts — the body of read_id_mt_pe::parallel_vec before kmer_poll_plus
// (src/read_id_mt_pe.rs:300-331; search_index_classic :66-102, search_index :104-165), whole indices and colour stripes, dense and
// sparse reports.  Kernels: cid_readid.hip (reads that fit a wave's LDS) and cid_kmerset.hip (the sort-based long-read path).
#include "cid_api_common.hpp"

using cid::check_ready;
using cid::fail;
using cid::slot_reserve;
using namespace cid::slots;

extern "C" {

// LDS layout of k_readid (bytes_kernel = false) or k_readid_bytes for reads of at most max_bytes bases / max_win windows;
// returns the bytes one wave needs (the kernels carve the same regions in the same order)
static size_t readid_layout(const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win,
                            bool bytes_kernel, cid::ReadIdParams *pp, int slot_kind = 0 /* 0: key + index, 1: one u64, 2: one u32 position */) {
    // (pp == NULL: only the size is wanted — the routing asks once per read, and clearing a parameter block per question cost a
    // millisecond per hundred thousand reads)
    const bool packed_table = slot_kind == 1;
    if (max_bytes > (1ull << 30) || max_win > (1ull << 30)) return ~(size_t)0;
    const bool wide = ix->rs > 128;
    const uint32_t bases_cap = (uint32_t)((max_bytes + 16 + 15) & ~15ull);
    uint32_t win_cap = (uint32_t)((max_win + 3) & ~3ull);
    if (win_cap < 4) win_cap = 4;
    const uint32_t hist_pad = wide ? 4u * ix->rs : ((ix->n_colors + 1 + 3) & ~3u);   // wide rows: AND word + sampled-colour set
    uint32_t table_slots = 64;
    while (table_slots < win_cap + win_cap / 2) table_slots <<= 1;
    size_t slot_bytes = 12;
    uint32_t idx_bits = 0;
    if (packed_table) {   // one u64 per slot: code << idx_bits | window index
        uint32_t ib = 1;
        while ((1ull << ib) <= win_cap) ++ib;
        if (2u * ix->k + ib > 63u) return ~(size_t)0;
        idx_bits = ib;
        slot_bytes = 8;
    }
    if (slot_kind == 2) slot_bytes = 4;
    const size_t chunk_rows = 4ull * cid::kWave * ix->n_hash;                     // one chunk's row numbers
    const size_t rall_bytes = wide ? 0 : 4ull * win_cap * ix->n_hash;             // rows of the read's distinct k-mers (wide rows search chunk by chunk)
    // k_readid keeps no byte image of the read, but the raw bases of the NEXT one (LDS-DMA, one 16-byte piece per lane)
    const uint32_t stage_bytes = bytes_kernel ? 0u : std::min<uint32_t>(bases_cap, 16u * cid::kWave);
    size_t wave_bytes = (bytes_kernel ? (size_t)bases_cap : (size_t)stage_bytes) + rall_bytes;
    if (bytes_kernel)   // histogram, tags, window infos, k-mer image (+ minimizer image and the distinct minimizer strings)
        wave_bytes += 4ull * hist_pad + chunk_rows + 8ull * win_cap + cid::kmer_img_bytes(ix->k) +
                      (ix->m_size ? cid::kmer_img_bytes(ix->m_size) + (((size_t)win_cap * ix->m_size + 15) & ~15ull) : 0);
    else if (wide)      // chunk rows, histogram, hash table keys + indices, 2-bit bases, bad-base bits
        wave_bytes += chunk_rows + 4ull * hist_pad + 12ull * table_slots + 4ull * (bases_cap / 16 + 4) + 4ull * (bases_cap / 32 + 4);
    else                // the histogram shares the hash table's region (k_readid)
        wave_bytes += (CID_READID_ALIAS ? std::max<size_t>(slot_bytes * table_slots, 4ull * hist_pad) : slot_bytes * table_slots + 4ull * hist_pad) +
                      4ull * (bases_cap / 16 + 4) + 4ull * (bases_cap / 32 + 4);
    wave_bytes = (wave_bytes + 15) & ~15ull;
    if (pp) {
        cid::ReadIdParams &p = *pp;
        p = cid::ReadIdParams{};
        p.mat = ix->mat; p.rs = ix->rs; p.w64 = ix->w64; p.n_colors = ix->n_colors; p.n_hash = ix->n_hash; p.k = ix->k;
        p.mod = ix->mod;
        p.stride_d = stride_d; p.start_sample = start_sample;
        p.m_size = ix->m_size;
        p.bases_cap = bases_cap; p.win_cap = win_cap; p.hist_pad = hist_pad; p.table_slots = table_slots; p.idx_bits = idx_bits;
        p.slot4 = slot_kind == 2 ? 1u : 0u;
        p.stage_bytes = stage_bytes;
        p.wave_bytes = (uint32_t)(wave_bytes < 0xFFFFFFF0ull ? wave_bytes : 0xFFFFFFF0ull);
    }
    return wave_bytes;
}
// what a read needs of the LDS kernels: k <= 32 reads may end up in either of them
static size_t readid_need(const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win) {
    const size_t b = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, true, nullptr);
    if (ix->k > 32) return b;
    const size_t a = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, false, nullptr);
    return a > b ? a : b;
}

constexpr size_t kLdsBytes = 160u * 1024u;
// device scratch for dense read_id report rows per launch: cid_readid_count slices larger batches, the sparse form refuses them
// (cid_ctx_tune "dense_report_bytes" / CID_DENSE_REPORT_BYTES: 2 GiB)
// k_readid keeps a read's set in one wave's LDS.  With fewer than six waves per CU its gathers are no longer hidden and the long-read
// path (cid_readlong.hip) is faster: 150 Mbases resident, configs[2]'s index, ms per call k_readid / long path (tools/exp_readlen_route.py,
// profiles/r06_readlen_route.jsonl): 300 bases 6.5 / 9.5, 600 bases 6.5 / 8.0, 800 bases (five waves) 8.7 / 7.7, 1000 bases (four waves)
// 12.9 / 7.5, 2000 bases 18.3 / 7.1.  (Round 5's long path took 10.4 at 800 bases: the rule then was "fewer than five waves"; round 1's
// sorting path took 20 ms whatever the length: "fewer than two".)
constexpr size_t kLdsReadBytesMax = kLdsBytes / 6;

static int readid_params(const cid_ctx *c, const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win,
                         bool bytes_kernel, cid::ReadIdParams &p, int &waves, bool striped = false) {
    size_t wave_bytes = 0, best = 0;
    auto choose = [&](int slot_kind) {
        wave_bytes = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, bytes_kernel, &p, slot_kind);
        // waves per workgroup: whatever puts the most waves on a CU (160 KiB of LDS, at most 8 workgroups of this size... 32 waves)
        waves = 1;
        best = 0;
        if (wave_bytes == ~(size_t)0) return;
        for (int w = 4; w >= 1; --w) {
            if ((size_t)w * wave_bytes > kLdsBytes) continue;
            size_t blocks = kLdsBytes / ((size_t)w * wave_bytes);
            if (blocks > 32u / (size_t)w) blocks = 32u / (size_t)w;
            if (blocks * (size_t)w > best) { best = blocks * (size_t)w; waves = w; }
        }
    };
    // the 8-byte-per-slot set is built for the six-waves-per-SIMD kernel only (whole k-mers, published hash, rows <= 1 KiB, no stripe
    // passes): taken when the 12-byte slots leave fewer than six waves per SIMD and the 8-byte ones reach them (paired 150-bp reads,
    // k <= 27: 10.6 -> 10.1 ms per million pairs, tools/exp_readid_table.py)
    const bool can_pack = !bytes_kernel && c->tune.readid_packed_table && ix->rs <= 128 && !ix->m_size && ix->k <= 31 &&
                          ((ix->mod.flags >> 8) & 0xFFu) == CID_HASH_XXH3_V08 && !striped;
    // ... and where the code leaves no room for the index (k = 28..32) the slot holds a position only (4 bytes; the probe reads the k-mer back)
    const bool can_slot4 = !bytes_kernel && c->tune.readid_packed_table && ix->rs <= 128 && !ix->m_size && ix->k <= 32 &&
                           ((ix->mod.flags >> 8) & 0xFFu) == CID_HASH_XXH3_V08 && !striped;
    choose(0);
    if (best < 24 && (can_pack || can_slot4)) {   // (where six waves per SIMD fit anyway the 12-byte slots are marginally faster: 6.02 vs 6.07 ms single-end)
        const size_t classic = best;
        if (can_pack) choose(1);
        if ((!can_pack || wave_bytes == ~(size_t)0 || best < 24) && can_slot4) choose(2);
        if (best <= classic || best <= 20) choose(0);   // worth it only with more waves than the 96-VGPR (5 per SIMD) build runs
    }
    if (wave_bytes > kLdsBytes)
        return fail(CID_ERR_UNSUPPORTED, "a read(-pair) of %llu bases / %llu windows needs %zu B of LDS per wave (> 160 KiB): "
                    "use the host-pointer calls, which route such reads through the sort-based path", (unsigned long long)max_bytes,
                    (unsigned long long)max_win, wave_bytes);
    return CID_OK;
}

using StripeArgs = cid::StripePass;

static int readid_dev_impl(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off,
                           const uint64_t *d_read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                           uint64_t max_read_bytes, uint64_t max_read_windows, const uint8_t *d_skip, bool clear_wide, uint32_t *d_report,
                           uint32_t *d_n_kmers, uint8_t *d_status, const StripeArgs &sa = StripeArgs()) {
    if (n_reads >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 reads in one batch");
    cid::ReadIdParams pb, pp;
    int waves_b, waves_p = 0;
    int rc = readid_params(c, ix, stride_d, start_sample, max_read_bytes, max_read_windows, true, pb, waves_b);
    if (rc) return rc;
    const bool packable = ix->k <= 32;
    const bool striped = sa.zero_acc || sa.zero_in;
    if (packable && (rc = readid_params(c, ix, stride_d, start_sample, max_read_bytes, max_read_windows, false, pp, waves_p, striped))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    auto fill = [&](cid::ReadIdParams &p, int waves) {
        p.bases = d_bases; p.seq_off = d_seq_off; p.read_seq0 = d_read_seq0; p.n_reads = n_reads;
        p.report = d_report; p.n_kmers = d_n_kmers; p.status = d_status; p.skip = d_skip;
        p.zero_acc = sa.zero_acc; p.zero_in = sa.zero_in; p.zero_start = sa.zero_start;
        p.colour_base = sa.colour_base; p.report_width = sa.report_width; p.write_nohits = sa.write_nohits;
        uint64_t rpb = n_reads / ((uint64_t)c->n_cu * (uint64_t)c->tune.readid_blocks_per_cu);   // (16 -> 64 per CU: -3 %, the tail of the grid)
        if (rpb < (uint64_t)waves) rpb = waves;
        if (rpb > 256) rpb = 256;
        p.reads_per_block = (uint32_t)rpb;
    };
    if (ix->rs > 128 && clear_wide)   // wide rows count in place
        HIP_TRY(hipMemsetAsync(d_report, 0, n_reads * ((size_t)ix->n_colors + 1) * 4, c->stream));
    if (!d_skip) {   // device-pointer callers state the maxima: reads beyond them are marked and left alone (k_readid_check_caps)
        void *d_sk;
        rc = slot_reserve(c, S_ROUTE, n_reads, &d_sk); if (rc) return rc;
        cid::ReadIdParams pc = pb;   // bases_cap / win_cap are the same for both kernels' layouts
        fill(pc, waves_b);
        pc.report_width = (sa.zero_acc || sa.zero_in) ? 0u : ix->n_colors + 1;   // striped passes only add to rows the caller zeroed
        HIP_TRY(cid::launch_readid_check_caps(pc, (uint8_t *)d_sk, c->stream));
        d_skip = (const uint8_t *)d_sk;
    }
    if (packable) {
        // k_readid takes every read it can pack; the ones with lower-case bases come back in the redo list for k_readid_bytes
        void *d_redo;
        rc = slot_reserve(c, S_REDO, 16 + 4 * n_reads, &d_redo); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_redo, 0, 16, c->stream));
        fill(pp, waves_p);
        pp.redo_count = (uint32_t *)d_redo; pp.redo_list = (uint32_t *)d_redo + 4;
        HIP_TRY(cid::launch_readid(pp, waves_p, c->stream));
        fill(pb, waves_b);
        pb.redo_count = pp.redo_count; pb.redo_list = pp.redo_list;
        uint64_t grid = (n_reads + waves_b - 1) / waves_b;
        if (grid > (uint64_t)c->n_cu * 4) grid = (uint64_t)c->n_cu * 4;
        HIP_TRY(cid::launch_readid_bytes(pb, waves_b, (int)grid, c->stream));
    } else {
        fill(pb, waves_b);
        uint64_t grid = (n_reads + waves_b - 1) / waves_b;
        if (grid > (uint64_t)c->n_cu * 64) grid = (uint64_t)c->n_cu * 64;
        HIP_TRY(cid::launch_readid_bytes(pb, waves_b, (int)grid, c->stream));
    }
    return CID_OK;
}

// Which kernel takes which read of a batch.  Reads of at least long_from(...) bases take the long-read path (cid_readlong.hip), the others
// the LDS kernels.  The rule is a threshold on a read's bases alone, so that the device applies it to offsets that never exist on the
// host (cid::long_route_launch): the smallest read whose set — with the windows a read of that many bases has, (bases - k) / stride + 1 —
// would leave k_readid fewer than six waves per CU (kLdsReadBytesMax).  cid_ctx_tune "readid_long_from" = L: every read of at least L
// bases does (measurements).
static uint64_t long_from(const cid_ctx *c, const cid_index *ix, uint32_t stride_d, uint32_t start_sample) {
    if (c->tune.readid_long_from >= 0) return (uint64_t)c->tune.readid_long_from;
    auto need = [&](uint64_t b) { return readid_need(ix, stride_d, start_sample, b, b >= ix->k ? (b - ix->k) / stride_d + 1 : 0); };
    uint64_t lo = 0, hi = 1ull << 21;   // need(lo) fits, need(hi) does not
    if (need(hi) <= kLdsReadBytesMax) return ~0ull;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) / 2;
        if (need(mid) <= kLdsReadBytesMax) lo = mid; else hi = mid;
    }
    return hi;
}
// validates host offsets; the longest read in bases and in windows
struct ReadRoute {
    uint64_t max_bytes = 0, max_win = 0;
    bool any_long = false;
};
static int readid_route(const cid_ctx *c, const cid_index *ix, const uint64_t *seq_off, size_t n_seqs, const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d,
                        uint32_t start_sample, ReadRoute &rr) {
    // (this loop runs on the caller's thread before anything is launched: a million reads of 150 bases took 2.1 ms in it — beside 5.2 ms of
    // kernel — while every read paid a 64-bit division by a stride that is 1 unless -d says otherwise)
    uint64_t max_bytes = 0, max_win = 0;
    const uint64_t k = ix->k;
    auto walk = [&](auto windows_of) -> int {
        for (size_t r = 0; r < n_reads; ++r) {   // (both bounds before seq_off is read through them)
            if (read_seq0[r + 1] < read_seq0[r]) return fail(CID_ERR_INVALID, "read_seq0 not monotonic at read %zu", r);
            if (read_seq0[r + 1] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs at read %zu", r);
            const uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
            uint64_t win = 0;
            for (uint64_t s = s0; s < s1; ++s) {
                if (seq_off[s + 1] < seq_off[s]) return fail(CID_ERR_INVALID, "seq_off not monotonic at seq %llu", (unsigned long long)s);
                const uint64_t len = seq_off[s + 1] - seq_off[s];
                if (len >= k) win += windows_of(len - k);
            }
            const uint64_t bytes = s1 > s0 ? seq_off[s1] - seq_off[s0] : 0;
            if (bytes > max_bytes) max_bytes = bytes;
            if (win > max_win) max_win = win;
        }
        return CID_OK;
    };
    const int rc_walk = stride_d == 1 ? walk([](uint64_t x) { return x + 1; }) : walk([stride_d](uint64_t x) { return x / stride_d + 1; });
    if (rc_walk) return rc_walk;
    rr.max_bytes = max_bytes; rr.max_win = max_win;
    rr.any_long = max_bytes >= long_from(c, ix, stride_d, start_sample);
    return CID_OK;
}

// A batch with long reads, everything on the device: the route (d_route: S_ROUTE), the long-read path for its reads, the LDS kernels for the
// rest.  cap_*: the maxima a device-pointer caller stated (reads beyond them: status 3), ~0 = none.  h_*: the offsets on the host, or NULL.
static int readid_routed(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0, size_t n_reads,
                         uint32_t stride_d, uint32_t start_sample, uint64_t cap_bytes, uint64_t cap_win, bool clear_wide, uint32_t *d_report, uint32_t *d_n_kmers,
                         uint8_t *d_status, const StripeArgs &sa, const uint64_t *h_seq_off, const uint64_t *h_read_seq0) {
    int rc;
    void *d_route;
    if ((rc = slot_reserve(c, S_ROUTE, n_reads + 16 + 16, &d_route))) return rc;
    uint32_t *d_stats = reinterpret_cast<uint32_t *>((uint8_t *)d_route + ((n_reads + 15) & ~(size_t)15));
    if ((rc = cid::long_route_launch(c, d_seq_off, d_read_seq0, n_reads, ix->k, stride_d, long_from(c, ix, stride_d, start_sample), cap_bytes, cap_win,
                                     (uint8_t *)d_route, d_stats)))
        return rc;
    const bool striped = sa.on();
    const size_t C1 = (size_t)ix->n_colors + 1;
    // wide rows count in place, and both kernels add into the same report: cleared once, here
    if (ix->rs > 128 && clear_wide && !striped) HIP_TRY(hipMemsetAsync(d_report, 0, n_reads * C1 * 4, c->stream));
    uint32_t stats[4] = {0, 0, 0, 0};
    // first: it writes a status for every read (2 = the other kernels')
    if ((rc = cid::readid_long(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, (const uint8_t *)d_route, false, d_report, d_n_kmers,
                               d_status, sa, h_seq_off, h_read_seq0, d_stats, stats)))
        return rc;
    if (stats[1])   // the reads of the LDS kernels, sized by the longest of THEM
        if ((rc = readid_dev_impl(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, stats[2], stats[3] ? stats[3] : 1,
                                  (const uint8_t *)d_route, false, d_report, d_n_kmers, d_status, sa)))
            return rc;
    if (cap_bytes != ~0ull || cap_win != ~0ull)
        if ((rc = cid::long_beyond_launch(c, (const uint8_t *)d_route, n_reads, striped ? 0u : (uint32_t)C1, d_report, d_n_kmers, d_status))) return rc;
    return CID_OK;
}

int cid_readid_count_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off,
                         const uint64_t *d_read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                         uint64_t max_read_bytes, uint64_t max_read_windows, uint32_t *d_report, uint32_t *d_n_kmers,
                         uint8_t *d_status) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (n_reads == 0) return CID_OK;
    if (!d_bases || !d_seq_off || !d_read_seq0 || !d_report || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (max_read_bytes >= long_from(c, ix, stride_d, start_sample))   // reads of any length: the long ones through cid_readlong.hip, in the same call
        return readid_routed(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, max_read_bytes, max_read_windows, true, d_report,
                             d_n_kmers, d_status, StripeArgs(), nullptr, nullptr);
    return readid_dev_impl(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, max_read_bytes, max_read_windows, nullptr,
                           true, d_report, d_n_kmers, d_status);
}

// read_id over colour stripes (src/read_id_mt_pe.rs:66-165 with the absent-row stop decided over ALL colours).
// Pass 1, once per stripe: d_zero_acc[read * max_read_windows + q] &= the seeds whose row is all-zero in this stripe, for the
// read's q-th distinct k-mer (first-occurrence order).  Between the passes the caller ANDs the arrays of different GPUs.
// Pass 2, once per stripe: the ordered count; a k-mer is "absent" iff its accumulated mask is non-zero.
__global__ void k_mask_starts(uint64_t *zero_start, uint64_t n_reads, uint64_t stride) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) zero_start[r] = r * stride;
}
// the device-pointer pair: masks laid out [read][max_read_windows]
static int readid_stripe_common(cid_ctx *c, const cid_index *ix, const void *d_bases, const void *d_seq_off, const void *d_read_seq0,
                                size_t n_reads, uint64_t max_read_bytes, uint64_t max_read_windows, const uint64_t **d_zero_start) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!d_bases || !d_seq_off || !d_read_seq0) return fail(CID_ERR_INVALID, "null argument");
    if (max_read_windows == 0 || max_read_windows > (1u << 20)) return fail(CID_ERR_INVALID, "max_read_windows out of range");
    if (readid_need(ix, 1, 0, max_read_bytes, max_read_windows) > kLdsBytes)
        return fail(CID_ERR_UNSUPPORTED,
                    "reads of %llu bases do not fit a wave's LDS: cid_readid_stripe_zero / _count route such reads through the sort-based path",
                    (unsigned long long)max_read_bytes);
    *d_zero_start = nullptr;
    if (n_reads == 0) return CID_OK;
    HIP_TRY(hipSetDevice(c->device));
    void *d_zs;
    rc = slot_reserve(c, S_ZSTART, n_reads * 8, &d_zs); if (rc) return rc;
    hipLaunchKernelGGL(k_mask_starts, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, c->stream, (uint64_t *)d_zs, (uint64_t)n_reads, max_read_windows);
    HIP_TRY(hipGetLastError());
    *d_zero_start = (const uint64_t *)d_zs;
    return CID_OK;
}

int cid_readid_stripe_zero_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0,
                               size_t n_reads, uint32_t stride_d, uint64_t max_read_bytes, uint64_t max_read_windows, uint32_t *d_zero_acc,
                               uint32_t *d_n_kmers, uint8_t *d_status) {
    const uint64_t *d_zs;
    int rc = readid_stripe_common(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, max_read_bytes, max_read_windows, &d_zs);
    if (rc) return rc;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (n_reads == 0) return CID_OK;
    if (!d_zero_acc || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    StripeArgs sa;
    sa.zero_acc = d_zero_acc; sa.zero_start = d_zs; sa.report_width = ix->n_colors + 1;
    return readid_dev_impl(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, 0, max_read_bytes, max_read_windows, nullptr, false,
                           reinterpret_cast<uint32_t *>(d_zero_acc) /* never written in this pass */, d_n_kmers, d_status, sa);
}

int cid_readid_stripe_count_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0,
                                size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint64_t max_read_bytes, uint64_t max_read_windows,
                                uint32_t colour_base, uint32_t n_colors_total, int write_nohits, const uint32_t *d_zero_acc, uint32_t *d_report,
                                uint32_t *d_n_kmers, uint8_t *d_status) {
    const uint64_t *d_zs;
    int rc = readid_stripe_common(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, max_read_bytes, max_read_windows, &d_zs);
    if (rc) return rc;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if ((uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) outside %u colours", colour_base,
                                                                           ix->n_colors, n_colors_total);
    if (n_reads == 0) return CID_OK;
    if (!d_zero_acc || !d_report || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    StripeArgs sa;
    sa.zero_in = d_zero_acc; sa.zero_start = d_zs; sa.colour_base = colour_base; sa.report_width = n_colors_total + 1;
    sa.write_nohits = write_nohits ? 1u : 0u;
    return readid_dev_impl(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, max_read_bytes, max_read_windows, nullptr, false,
                           d_report, d_n_kmers, d_status, sa);
}

// The two stripe passes for ANY read length and stripe width: d_bases resident, offsets on the host.  Per stripe the reads are routed
// between the LDS kernels and the sort-based path exactly as cid_readid_count routes them (the mask of a read's q-th distinct k-mer
// sits at the same word whichever kernel writes it, so different stripes may route a read differently).  Masks: one word per
// window, read r's at [prefix of the windows of reads 0..r-1] (cid_readid_stripe_mask_words words in all).
static int stripe_mask_starts(uint32_t k, uint32_t stride_d, const uint64_t *seq_off, uint64_t n_seqs, const uint64_t *read_seq0, size_t n_reads,
                              std::vector<uint64_t> &zs) {
    zs.assign(n_reads + 1, 0);
    for (size_t r = 0; r < n_reads; ++r) {
        if (read_seq0[r + 1] < read_seq0[r]) return fail(CID_ERR_INVALID, "read_seq0 not monotonic at read %zu", r);
        if (read_seq0[r + 1] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs at read %zu", r);
        uint64_t win = 0;
        for (uint64_t s = read_seq0[r]; s < read_seq0[r + 1]; ++s) {
            if (seq_off[s + 1] < seq_off[s]) return fail(CID_ERR_INVALID, "seq_off not monotonic at seq %llu", (unsigned long long)s);
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len >= k) win += (len - k) / stride_d + 1;
        }
        zs[r + 1] = zs[r] + win;
    }
    return CID_OK;
}

int cid_readid_stripe_mask_words(uint32_t k_size, uint32_t stride_d, const uint64_t *seq_off, const uint64_t *read_seq0, size_t n_reads, uint64_t *n_words) {
    if (!seq_off || !read_seq0 || !n_words) return fail(CID_ERR_INVALID, "null argument");
    if (stride_d == 0 || k_size == 0) return fail(CID_ERR_INVALID, "k_size and stride_d must be >= 1");
    std::vector<uint64_t> zs;
    const int rc = stripe_mask_starts(k_size, stride_d, seq_off, ~0ull /* the caller vouches for seq_off's length */, read_seq0, n_reads, zs);
    if (rc) return rc;
    *n_words = zs[n_reads] + 1;   // never empty
    return CID_OK;
}

static int readid_stripe_pass(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs, const uint64_t *read_seq0,
                              size_t n_reads, uint32_t stride_d, uint32_t start_sample, StripeArgs sa, uint32_t *d_report, uint32_t *d_n_kmers,
                              uint8_t *d_status) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!seq_off || !read_seq0) return fail(CID_ERR_INVALID, "null argument");
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (n_reads == 0) return CID_OK;
    if (!d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    if (read_seq0[n_reads] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs");
    if (seq_off[n_seqs] && !d_bases) return fail(CID_ERR_INVALID, "null bases");
    ReadRoute rr;
    if ((rc = readid_route(c, ix, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, rr))) return rc;
    std::vector<uint64_t> zs;
    if ((rc = stripe_mask_starts(ix->k, stride_d, seq_off, n_seqs, read_seq0, n_reads, zs))) return rc;
    if (zs[n_reads] >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    HIP_TRY(hipSetDevice(c->device));
    void *d_so, *d_r0, *d_zs;
    rc = slot_reserve(c, S_SEQOFF, (n_seqs + 1) * 8, &d_so); if (rc) return rc;
    rc = slot_reserve(c, S_READ0, (n_reads + 1) * 8, &d_r0); if (rc) return rc;
    rc = slot_reserve(c, S_ZSTART, (n_reads + 1) * 8, &d_zs); if (rc) return rc;
    {   // the three offset arrays through the pinned arena when they fit (cid::pin_reserve)
        const size_t b0 = (n_seqs + 1) * 8, b1 = (n_reads + 1) * 8;
        const uint8_t *so_src = reinterpret_cast<const uint8_t *>(seq_off), *r0_src = reinterpret_cast<const uint8_t *>(read_seq0),
                      *zs_src = reinterpret_cast<const uint8_t *>(zs.data());
        if (uint8_t *pin = cid::pin_reserve(c, b0 + 2 * b1 + 64)) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            memcpy(pin, seq_off, b0); memcpy(pin + b0, read_seq0, b1); memcpy(pin + b0 + b1, zs.data(), b1);
            so_src = pin; r0_src = pin + b0; zs_src = pin + b0 + b1;
        }
        HIP_TRY(hipMemcpyAsync(d_so, so_src, b0, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_r0, r0_src, b1, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_zs, zs_src, b1, hipMemcpyHostToDevice, c->stream));
    }
    sa.zero_start = (const uint64_t *)d_zs;
    if (rr.any_long)   // (the mask of a read's q-th distinct k-mer sits at the same word whichever kernel writes it)
        rc = readid_routed(c, ix, d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample, ~0ull, ~0ull, false, d_report,
                           d_n_kmers, d_status, sa, seq_off, read_seq0);
    else
        rc = readid_dev_impl(c, ix, d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample, rr.max_bytes,
                             rr.max_win ? rr.max_win : 1, nullptr, false, d_report, d_n_kmers, d_status, sa);
    HIP_TRY(hipStreamSynchronize(c->stream));   // the host vectors leave scope
    return rc;
}

int cid_readid_stripe_zero(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs, const uint64_t *read_seq0,
                           size_t n_reads, uint32_t stride_d, uint32_t *d_zero_acc, uint32_t *d_n_kmers, uint8_t *d_status) {
    if (n_reads && !d_zero_acc) return fail(CID_ERR_INVALID, "null argument");
    StripeArgs sa;
    sa.zero_acc = d_zero_acc; sa.report_width = ix ? ix->n_colors + 1 : 0;
    return readid_stripe_pass(c, ix, d_bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, 0, sa,
                              reinterpret_cast<uint32_t *>(d_zero_acc) /* never written in this pass */, d_n_kmers, d_status);
}

int cid_readid_stripe_count(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs, const uint64_t *read_seq0,
                            size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t colour_base, uint32_t n_colors_total, int write_nohits,
                            const uint32_t *d_zero_acc, uint32_t *d_report, uint32_t *d_n_kmers, uint8_t *d_status) {
    if (n_reads && (!d_zero_acc || !d_report)) return fail(CID_ERR_INVALID, "null argument");
    if (ix && (uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) outside %u colours", colour_base,
        ix->n_colors, n_colors_total);
    StripeArgs sa;
    sa.zero_in = d_zero_acc; sa.colour_base = colour_base; sa.report_width = n_colors_total + 1; sa.write_nohits = write_nohits ? 1u : 0u;
    return readid_stripe_pass(c, ix, d_bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, sa, d_report, d_n_kmers, d_status);
}

// bases resident, offsets on the host: routes every read between the LDS kernels and the long-read path, uploads the offsets,
// runs the kernels into the caller's device arrays.  `d_so` / `d_r0` non-null: the offsets are on the device already (the caller
// uploaded them together with the bases).
static int readid_resident(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs,
                           const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, void *d_so, void *d_r0,
                           uint32_t *d_rep, uint32_t *d_nk, uint8_t *d_status) {
    int rc;
    ReadRoute rr;
    if ((rc = readid_route(c, ix, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, rr))) return rc;
    if (!d_so) {
        rc = slot_reserve(c, S_SEQOFF, (n_seqs + 1) * 8, &d_so); if (rc) return rc;
        rc = slot_reserve(c, S_READ0, (n_reads + 1) * 8, &d_r0); if (rc) return rc;
        const size_t b0 = (n_seqs + 1) * 8, b1 = (n_reads + 1) * 8;
        const uint8_t *so_src = reinterpret_cast<const uint8_t *>(seq_off), *r0_src = reinterpret_cast<const uint8_t *>(read_seq0);
        if (uint8_t *pin = cid::pin_reserve(c, b0 + b1 + 64)) {
            HIP_TRY(hipStreamSynchronize(c->stream));   // (the arena may still feed the previous call's copies)
            memcpy(pin, seq_off, b0); memcpy(pin + b0, read_seq0, b1);
            so_src = pin; r0_src = pin + b0;
        }
        HIP_TRY(hipMemcpyAsync(d_so, so_src, b0, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_r0, r0_src, b1, hipMemcpyHostToDevice, c->stream));
    }
    if (rr.any_long)   // the device routes every read between the long-read path and the LDS kernels and makes the former's work lists
        return readid_routed(c, ix, d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample, ~0ull, ~0ull, true, d_rep, d_nk,
                             d_status, StripeArgs(), seq_off, read_seq0);
    return readid_dev_impl(c, ix, d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample, rr.max_bytes,
                           rr.max_win ? rr.max_win : 1, nullptr, true, d_rep, d_nk, d_status);
}

int cid_readid_count_resident(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs,
                              const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t *d_report,
                              uint32_t *d_n_kmers, uint8_t *d_status) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!seq_off || !read_seq0) return fail(CID_ERR_INVALID, "null argument");
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (n_reads == 0) return CID_OK;
    if (!d_report || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    if (read_seq0[n_reads] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs");
    if (seq_off[n_seqs] && !d_bases) return fail(CID_ERR_INVALID, "null bases");
    HIP_TRY(hipSetDevice(c->device));
    return readid_resident(c, ix, d_bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, nullptr, nullptr, d_report, d_n_kmers,
                           d_status);
}

// uploads the batch, runs the LDS kernels or the long-read path; leaves report / n_kmers / status in the ctx's device scratch
static int readid_to_device(cid_ctx *c, const cid_index *ix, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                            const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t **d_report_out,
                            uint32_t **d_nk_out, uint8_t **d_status_out) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!seq_off || !read_seq0) return fail(CID_ERR_INVALID, "null argument");
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (read_seq0[n_reads] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs");
    const uint64_t total_bases = seq_off[n_seqs];
    if (total_bases && !bases) return fail(CID_ERR_INVALID, "null bases");
    HIP_TRY(hipSetDevice(c->device));
    void *d_bases, *d_so, *d_r0, *d_rep, *d_nk;
    const size_t C1 = (size_t)ix->n_colors + 1;
    rc = slot_reserve(c, S_BASES, total_bases, &d_bases); if (rc) return rc;
    rc = slot_reserve(c, S_SEQOFF, (n_seqs + 1) * 8, &d_so); if (rc) return rc;
    rc = slot_reserve(c, S_READ0, (n_reads + 1) * 8, &d_r0); if (rc) return rc;
    rc = slot_reserve(c, S_REPORT, n_reads * C1 * 4, &d_rep); if (rc) return rc;
    rc = slot_reserve(c, S_NK, n_reads * 4 + n_reads + 16, &d_nk); if (rc) return rc;
    {   // the batch goes through the ctx's pinned arena when it fits (cid::pin_reserve); the arena's tail is left for the results
        // Bases that lie in page-locked memory already (cid_pinned_alloc: the CLI's batches of long reads) travel from where they are: staged,
        // 48 MB of them were 10 ms of memcpy on the calling thread before 2 ms on the bus.
        bool bases_locked = false;
        if (total_bases >= ((size_t)4 << 20)) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, bases) == hipSuccess) bases_locked = at.type == hipMemoryTypeHost;
            else (void)hipGetLastError();   // (memory the runtime has never seen: not an error of this call)
        }
        const size_t staged_bases = bases_locked ? 0 : total_bases;
        const size_t b_so = (staged_bases + 15) & ~(size_t)15, b_r0 = b_so + (n_seqs + 1) * 8, b_end = b_r0 + (n_reads + 1) * 8;
        uint8_t *pin = cid::pin_reserve(c, b_end + n_reads * 5 + 64);
        if (pin) {
            HIP_TRY(hipStreamSynchronize(c->stream));   // (the arena may still feed the previous call's copies)
            if (staged_bases) memcpy(pin, bases, total_bases);
            memcpy(pin + b_so, seq_off, (n_seqs + 1) * 8);
            memcpy(pin + b_r0, read_seq0, (n_reads + 1) * 8);
            if (total_bases) HIP_TRY(hipMemcpyAsync(d_bases, bases_locked ? bases : pin, total_bases, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_so, pin + b_so, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_r0, pin + b_r0, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
        } else {
            if (total_bases) HIP_TRY(hipMemcpyAsync(d_bases, bases, total_bases, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_so, seq_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_r0, read_seq0, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
        }
    }
    uint8_t *d_status = (uint8_t *)d_nk + n_reads * 4;
    rc = readid_resident(c, ix, (const uint8_t *)d_bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, d_so, d_r0, (uint32_t *)d_rep,
                         (uint32_t *)d_nk, d_status);
    if (rc) return rc;
    *d_report_out = (uint32_t *)d_rep; *d_nk_out = (uint32_t *)d_nk; *d_status_out = d_status;
    return CID_OK;
}

int cid_readid_count(cid_ctx *c, const cid_index *ix, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                     const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                     uint32_t *report, uint32_t *n_kmers, uint8_t *status) {
    if (n_reads == 0) return check_ready(c, ix);
    if (!report || !n_kmers || !status) return fail(CID_ERR_INVALID, "null argument");
    if (!c || !ix || !seq_off || !read_seq0) return fail(CID_ERR_INVALID, "null argument");
    // A dense report row has n_colors+1 counters (4 GB per million reads at 1024 colours): the batch is worked through in
    // slices whose rows fit kDenseReportBytes of device scratch; a read's row does not depend on its neighbours.
    const size_t C1 = (size_t)ix->n_colors + 1;
    size_t per = (size_t)c->tune.dense_report_bytes / (C1 * 4);
    if (per == 0) per = 1;
    std::vector<uint64_t> so, r0v;
    for (size_t r0 = 0; r0 < n_reads; r0 += per) {
        const size_t nr = n_reads - r0 < per ? n_reads - r0 : per;
        const uint64_t *so_p = seq_off, *r0_p = read_seq0;
        const uint8_t *bases_p = bases;
        size_t ns = n_seqs;
        if (nr != n_reads) {   // rebase the slice: its own seq_off / read_seq0 starting at 0
            if (read_seq0[r0 + nr] > n_seqs || read_seq0[r0] > read_seq0[r0 + nr]) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs");
            const uint64_t s0 = read_seq0[r0], s1 = read_seq0[r0 + nr];
            ns = (size_t)(s1 - s0);
            so.resize(ns + 1);
            for (size_t i = 0; i <= ns; ++i) {
                if (seq_off[s0 + i] < seq_off[s0]) return fail(CID_ERR_INVALID, "seq_off not monotonic at seq %llu", (unsigned long long)(s0 + i));
                so[i] = seq_off[s0 + i] - seq_off[s0];
            }
            r0v.resize(nr + 1);
            for (size_t i = 0; i <= nr; ++i) r0v[i] = read_seq0[r0 + i] - s0;
            so_p = so.data(); r0_p = r0v.data();
            bases_p = bases ? bases + seq_off[s0] : nullptr;
        }
        uint32_t *d_rep, *d_nk;
        uint8_t *d_st;
        int rc = readid_to_device(c, ix, bases_p, so_p, ns, r0_p, nr, stride_d, start_sample, &d_rep, &d_nk, &d_st);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(report + r0 * C1, d_rep, nr * C1 * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(n_kmers + r0, d_nk, nr * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(status + r0, d_st, nr, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CID_OK;
}

int cid_readid_count_sparse(cid_ctx *c, const cid_index *ix, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                            const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                            uint32_t *n_kmers, uint8_t *status, uint64_t *n_entries) {
    if (!n_entries) return fail(CID_ERR_INVALID, "null argument");
    *n_entries = 0;
    if (n_reads == 0) { int rc0 = check_ready(c, ix); if (rc0 == CID_OK) { c->sp_rows = 0; c->sp_entries = 0; } return rc0; }
    if (!n_kmers || !status) return fail(CID_ERR_INVALID, "null argument");
    if (ix && (double)n_reads * ((double)ix->n_colors + 1.0) * 4.0 > 64.0 * (double)(1ull << 30))
        return fail(CID_ERR_UNSUPPORTED, "%zu reads x %u colours need more than 64 GiB of dense report rows on the device: use smaller batches",
                    n_reads, ix->n_colors);
    uint32_t *d_rep, *d_nk;
    uint8_t *d_st;
    int rc = readid_to_device(c, ix, bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, &d_rep, &d_nk, &d_st);
    if (rc) return rc;
    cid::ctx_free(c, c->sp_start); c->sp_start = nullptr;
    cid::ctx_free(c, c->sp_col); c->sp_col = nullptr;
    cid::ctx_free(c, c->sp_cnt); c->sp_cnt = nullptr;
    rc = cid::compact_report(c, d_rep, ix->n_colors + 1, n_reads, &c->sp_start, &c->sp_col, &c->sp_cnt, &c->sp_entries);
    if (rc) return rc;
    c->sp_rows = n_reads;
    if (uint8_t *pin = cid::pin_reserve(c, n_reads * 5 + 64)) {   // (inputs are on the device by now: the arena is free again)
        HIP_TRY(hipMemcpyAsync(pin, d_nk, n_reads * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(pin + n_reads * 4, d_st, n_reads, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(n_kmers, pin, n_reads * 4);
        memcpy(status, pin + n_reads * 4, n_reads);
    } else {
        HIP_TRY(hipMemcpyAsync(n_kmers, d_nk, n_reads * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(status, d_st, n_reads, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *n_entries = c->sp_entries;
    return CID_OK;
}

int cid_readid_sparse_fetch(cid_ctx *c, uint64_t *row_start, uint32_t *colours, uint32_t *counts) {
    if (!c || !row_start) return fail(CID_ERR_INVALID, "null argument");
    if (c->sp_rows == 0) { row_start[0] = 0; return CID_OK; }
    if (c->sp_entries && (!colours || !counts)) return fail(CID_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t b_rs = (c->sp_rows + 1) * 8, b_e = c->sp_entries * 4;
    if (uint8_t *pin = cid::pin_reserve(c, b_rs + 2 * b_e + 64)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpyAsync(pin, c->sp_start, b_rs, hipMemcpyDeviceToHost, c->stream));
        if (b_e) {
            HIP_TRY(hipMemcpyAsync(pin + b_rs, c->sp_col, b_e, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(pin + b_rs + b_e, c->sp_cnt, b_e, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(row_start, pin, b_rs);
        if (b_e) { memcpy(colours, pin + b_rs, b_e); memcpy(counts, pin + b_rs + b_e, b_e); }
        return CID_OK;
    }
    HIP_TRY(hipMemcpy(row_start, c->sp_start, (c->sp_rows + 1) * 8, hipMemcpyDeviceToHost));
    if (c->sp_entries) {
        HIP_TRY(hipMemcpy(colours, c->sp_col, c->sp_entries * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(counts, c->sp_cnt, c->sp_entries * 4, hipMemcpyDeviceToHost));
    }
    return CID_OK;
}
}  // extern "C"
