// C ABI of libcolorid_hip.so, part 3: the searches — a5 proportional (src/batch_search_pe.rs:45-84, :125-164) and a4 perfect
// (src/perfect_search.rs:25-52, :83-110), whole indices and colour stripes, device-pointer and host-pointer forms.
// Kernels: cid_search.hip.
#include "cid_api_common.hpp"

using cid::aligned16;
using cid::check_not_mini;
using cid::check_ready;
using cid::fail;
using cid::pick_tiles_per_block;
using cid::slot_reserve;
using namespace cid::slots;

namespace {

// bytes per chunk of the pipelined host-pointer calls (H2D of chunk i+1 beside the kernel of chunk i)
// (cid_ctx_tune "upload_chunk_bytes" / CID_UPLOAD_CHUNK_BYTES: 256 MiB of k-mers per upload chunk)

int fill_search_params(const cid_ctx *c, const cid_index *ix, cid::SearchParams &p) {
    memset(&p, 0, sizeof(p));
    p.mat = ix->mat;
    p.rs = ix->rs;
    p.w64 = ix->w64;
    p.n_colors = ix->n_colors;
    p.n_hash = ix->n_hash;
    p.k = ix->k;
    p.c_pad = (ix->n_colors + 1u) & ~1u;
    if (p.c_pad < 2) p.c_pad = 2;
    p.wave_bytes = cid::kmer_img_bytes(ix->k) + cid::kWave * ix->n_hash * 4u + 2u * cid::kWave * 4u;   // image, hash rows, per-k-mer results of the tile
    p.wave_bytes = (p.wave_bytes + 15u) & ~15u;
    if (ix->rs > 128) {  // wide rows: no block histogram; the perfect search keeps a per-wave AND accumulator of rs words
        p.c_pad = 0;
        p.wave_bytes += 8u * ix->rs;
    }
    p.mod = ix->mod;
    p.unroll = (uint32_t)c->tune.search_unroll;
    if (cid::search_smem_bytes(p) > 160u * 1024u)
        return fail(CID_ERR_UNSUPPORTED, "LDS need %zu B exceeds 160 KiB (n_colors=%u k=%u n_hash=%u)",
                    cid::search_smem_bytes(p), ix->n_colors, ix->k, ix->n_hash);
    return CID_OK;
}
}  // namespace

int cid::search_count_launch(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint64_t *d_codes, const uint32_t *d_freq,
                             size_t n_kmers, uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq,
                             uint32_t *d_unique_colour, bool zero_counters) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if ((rc = check_not_mini(ix))) return rc;
    if (!d_hits || (n_kmers && !d_kmers && !d_codes)) return fail(CID_ERR_INVALID, "null argument");
    if (d_kmers && !aligned16(d_kmers)) return fail(CID_ERR_INVALID, "d_kmers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    cid::SearchParams p;
    rc = fill_search_params(c, ix, p);
    if (rc) return rc;
    p.kmers = d_kmers; p.codes = d_codes; p.freq = d_freq; p.n_kmers = n_kmers;
    p.hits = d_hits; p.n_unique = d_n_unique; p.sum_unique_freq = d_sum_unique_freq; p.unique_colour = d_unique_colour;
    p.want_unique = (d_n_unique || d_sum_unique_freq || d_unique_colour) ? 1u : 0u;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
#ifdef CID_TUNE_BUILD
    p.mixed = c->tune.search_mixed ? 1u : 0u;
    if (c->tune.search_persist && ix->rs <= 128 && n_kmers >= (1u << 16)) {   // persistent grid, one work queue per XCD (cid_search.hip)
        void *d_q;
        rc = slot_reserve(c, S_QUEUE, 8 * 128, &d_q); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_q, 0, 8 * 128, c->stream));
        p.queues = (uint32_t *)d_q;
        p.persist_grid = c->n_cu * cid::search_count_blocks_per_cu(p);
        if (p.persist_grid <= 0) p.queues = nullptr;
    }
#endif
    const size_t cb = (size_t)ix->n_colors * 8;
    if (zero_counters) {
        HIP_TRY(hipMemsetAsync(d_hits, 0, cb, c->stream));
        if (d_n_unique) HIP_TRY(hipMemsetAsync(d_n_unique, 0, cb, c->stream));
        if (d_sum_unique_freq) HIP_TRY(hipMemsetAsync(d_sum_unique_freq, 0, cb, c->stream));
    }
    HIP_TRY(cid::launch_search_count(p, c->stream));
    return CID_OK;
}
using cid::search_count_launch;
extern "C" {

int cid_search_count_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint32_t *d_freq, size_t n_kmers,
                         uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq, uint32_t *d_unique_colour) {
    return search_count_launch(c, ix, d_kmers, nullptr, d_freq, n_kmers, d_hits, d_n_unique, d_sum_unique_freq, d_unique_colour);
}

// One colour stripe of a wider index: per-colour hits are final; per-k-mer popcounts and unique
// candidates accumulate across the stripes' calls and are resolved by cid_search_unique_finalize_dev.
int cid_search_count_stripe_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint64_t *d_codes, size_t n_kmers,
                                uint32_t colour_base, uint64_t *d_hits, uint32_t *d_fact) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if ((rc = check_not_mini(ix))) return rc;
    if (!d_hits || !d_fact || (n_kmers && !d_kmers && !d_codes)) return fail(CID_ERR_INVALID, "null argument");
    if (d_kmers && !aligned16(d_kmers)) return fail(CID_ERR_INVALID, "d_kmers must be 16-byte aligned");
    if (d_codes && ix->k > 32) return fail(CID_ERR_UNSUPPORTED, "2-bit codes need k_size <= 32");
    HIP_TRY(hipSetDevice(c->device));
    cid::SearchParams p;
    rc = fill_search_params(c, ix, p);
    if (rc) return rc;
    p.kmers = d_kmers; p.codes = d_codes; p.n_kmers = n_kmers; p.hits = d_hits;
    p.colour_base = colour_base; p.fact = d_fact;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
    HIP_TRY(hipMemsetAsync(d_hits, 0, (size_t)ix->n_colors * 8, c->stream));
    HIP_TRY(cid::launch_search_count(p, c->stream));
    return CID_OK;
}

int cid_search_unique_finalize_dev(cid_ctx *c, const uint32_t *d_fact, const uint32_t *d_freq,
                                   size_t n_kmers, uint32_t n_colors_total, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq,
                                   uint32_t *d_unique_colour) {
    if (!c || (n_kmers && !d_fact) || n_colors_total == 0 || n_colors_total > (1u << 20)) return fail(CID_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(cid::launch_unique_finalize(d_fact, d_freq, n_kmers, n_colors_total, d_n_unique, d_sum_unique_freq,
                                        d_unique_colour, c->stream));
    return CID_OK;
}

// Perfect search on one stripe: the stripe's AND words are final; d_zero_acc[n_kmers] (preset to all-ones) collects,
// per k-mer, the seeds whose row is all-zero in every stripe so far — any bit left at the end means "row absent".
int cid_search_perfect_stripe_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_kmers, const uint64_t *d_codes, size_t n_kmers,
                                  uint64_t *d_and_words, uint32_t *d_zero_acc) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if ((rc = check_not_mini(ix))) return rc;
    if (!d_and_words || !d_zero_acc || (n_kmers && !d_kmers && !d_codes)) return fail(CID_ERR_INVALID, "null argument");
    if (d_codes && ix->k > 32) return fail(CID_ERR_UNSUPPORTED, "2-bit codes need k_size <= 32");
    if (d_kmers && !aligned16(d_kmers)) return fail(CID_ERR_INVALID, "d_kmers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    void *d_scratch;
    rc = slot_reserve(c, S_MISC, 16, &d_scratch);
    if (rc) return rc;
    cid::SearchParams p;
    rc = fill_search_params(c, ix, p);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_and_words, 0xFF, (size_t)ix->rs * 8, c->stream));
    p.kmers = d_kmers; p.codes = d_codes; p.n_kmers = n_kmers; p.and_words = d_and_words; p.missing = (int *)d_scratch;
    p.zero_acc = d_zero_acc;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
    HIP_TRY(cid::launch_search_perfect(p, c->stream));
    return CID_OK;
}
int cid_search_count_codes_dev(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, const uint32_t *d_freq, size_t n_kmers,
                               uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq, uint32_t *d_unique_colour) {
    if (ix && ix->k > 32) return fail(CID_ERR_UNSUPPORTED, "2-bit codes need k_size <= 32");
    return search_count_launch(c, ix, nullptr, d_codes, d_freq, n_kmers, d_hits, d_n_unique, d_sum_unique_freq, d_unique_colour);
}

// host results for k-mers (ASCII `d_k` or codes `d_codes`) that are already on the device
static int search_count_to_host(cid_ctx *c, const cid_index *ix, const uint8_t *d_k, const uint64_t *d_codes, const uint32_t *d_f,
                                size_t n_kmers, uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    const size_t C = ix->n_colors;
    void *d_out, *d_uc = nullptr;
    int rc = slot_reserve(c, S_OUT, 3 * C * 8, &d_out);
    if (rc) return rc;
    if (unique_colour) { rc = slot_reserve(c, S_UC, n_kmers * 4, &d_uc); if (rc) return rc; }
    uint64_t *o = (uint64_t *)d_out;
    rc = search_count_launch(c, ix, d_k, d_codes, d_f, n_kmers, o, n_unique ? o + C : nullptr, sum_unique_freq ? o + 2 * C : nullptr,
                             (uint32_t *)d_uc);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(hits, o, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_unique) HIP_TRY(hipMemcpyAsync(n_unique, o + C, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (sum_unique_freq) HIP_TRY(hipMemcpyAsync(sum_unique_freq, o + 2 * C, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (unique_colour) HIP_TRY(hipMemcpyAsync(unique_colour, d_uc, n_kmers * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

// Host-pointer form.  The batch goes through in chunks: the H2D copy of chunk i+1 (copy stream) runs beside the kernel of chunk i
// (ctx stream), and the per-k-mer results of chunk i-1 come back while both run; counters accumulate on the device over the
// chunks.  What is left is the PCIe time of 31+4 bytes in and 4 bytes out per k-mer.
}  // extern "C"
// host k-mers in, per-k-mer results out to the host, the 3*C counters (hits | n_unique | sum_unique_freq) left on the device in
// *d_counters (the ctx's S_OUT slot); returns with both streams drained
int cid::search_count_host_input(cid_ctx *c, const cid_index *ix, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers, bool want_unique,
                                 uint32_t *unique_colour, uint64_t **d_counters) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (n_kmers && !kmers) return fail(CID_ERR_INVALID, "null argument");
    uint64_t hits_dummy = 0;
    uint64_t *hits = &hits_dummy, *n_unique = want_unique ? &hits_dummy : nullptr, *sum_unique_freq = n_unique;
    HIP_TRY(hipSetDevice(c->device));
    const size_t C = ix->n_colors, k = ix->k;
    size_t chunk = (size_t)c->tune.upload_chunk_bytes / (k + 8);
    chunk = (chunk + 63) & ~(size_t)63;          // chunks start on a tile boundary: 64*k bytes keep the 16-byte alignment of the k-mer array
    if (chunk >= n_kmers || c->stream != c->own_stream) chunk = n_kmers ? n_kmers : 1;   // a borrowed stream: keep everything on it
    void *d_k, *d_f = nullptr, *d_out, *d_uc = nullptr;
    const size_t two = chunk < n_kmers ? 2 : 1;
    rc = slot_reserve(c, S_KMERS, two * chunk * k, &d_k); if (rc) return rc;
    if (freq) { rc = slot_reserve(c, S_FREQ, two * chunk * 4, &d_f); if (rc) return rc; }
    rc = slot_reserve(c, S_OUT, 3 * C * 8, &d_out); if (rc) return rc;
    if (unique_colour) { rc = slot_reserve(c, S_UC, two * chunk * 4, &d_uc); if (rc) return rc; }
    uint64_t *o = (uint64_t *)d_out;
    HIP_TRY(hipMemsetAsync(o, 0, 3 * C * 8, c->stream));
    const bool piped = two == 2;
    hipStream_t cs = piped ? c->copy_stream : c->stream;
    size_t prev_first = 0, prev_n = 0;
    int prev_b = 0;
    size_t i = 0;
    for (size_t first = 0; first < n_kmers || first == 0; first += chunk, ++i) {
        const size_t nk = n_kmers - first < chunk ? n_kmers - first : chunk;
        const int b = (int)(i & 1);
        uint8_t *dk = (uint8_t *)d_k + (size_t)b * chunk * k;
        uint32_t *df = d_f ? (uint32_t *)d_f + (size_t)b * chunk : nullptr;
        uint32_t *du = d_uc ? (uint32_t *)d_uc + (size_t)b * chunk : nullptr;
        if (piped && i >= 2) HIP_TRY(hipStreamWaitEvent(cs, c->ev_done[b], 0));     // buffer b's previous kernel has consumed it
        if (nk) HIP_TRY(hipMemcpyAsync(dk, kmers + first * k, nk * k, hipMemcpyHostToDevice, cs));
        if (nk && freq) HIP_TRY(hipMemcpyAsync(df, freq + first, nk * 4, hipMemcpyHostToDevice, cs));
        if (piped) {
            HIP_TRY(hipEventRecord(c->ev_copied[b], cs));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
        }
        rc = search_count_launch(c, ix, dk, nullptr, df, nk, o, n_unique ? o + C : nullptr, sum_unique_freq ? o + 2 * C : nullptr, du, false);
        if (rc) return rc;
        if (piped) HIP_TRY(hipEventRecord(c->ev_done[b], c->stream));
        // the previous chunk's per-k-mer results: its kernel finished while this chunk was copied in
        if (unique_colour && prev_n) {
            if (piped) HIP_TRY(hipStreamWaitEvent(cs, c->ev_done[prev_b], 0));
            HIP_TRY(hipMemcpyAsync(unique_colour + prev_first, (uint32_t *)d_uc + (size_t)prev_b * chunk, prev_n * 4, hipMemcpyDeviceToHost, cs));
        }
        prev_first = first; prev_n = nk; prev_b = b;
        if (n_kmers == 0) break;
    }
    if (unique_colour && prev_n)
        HIP_TRY(hipMemcpyAsync(unique_colour + prev_first, (uint32_t *)d_uc + (size_t)prev_b * chunk, prev_n * 4, hipMemcpyDeviceToHost, c->stream));
    if (piped) HIP_TRY(hipStreamSynchronize(cs));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hits;
    *d_counters = o;
    return CID_OK;
}
extern "C" {

int cid_search_count(cid_ctx *c, const cid_index *ix, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                     uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    uint64_t *o = nullptr;
    int rc = cid::search_count_host_input(c, ix, kmers, freq, n_kmers, n_unique || sum_unique_freq || unique_colour, unique_colour, &o);
    if (rc) return rc;
    const size_t C = ix->n_colors;
    HIP_TRY(hipMemcpyAsync(hits, o, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_unique) HIP_TRY(hipMemcpyAsync(n_unique, o + C, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (sum_unique_freq) HIP_TRY(hipMemcpyAsync(sum_unique_freq, o + 2 * C, C * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

// ------------------------------------------------------------------------------------------------ a4

}  // extern "C"
int cid::search_perfect_launch(cid_ctx *c, const cid_index *ix, const uint8_t *d_k, const uint64_t *d_codes, size_t n_kmers, uint64_t *d_and,
                               int *d_missing) {
    int rc = check_not_mini(ix);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_and, 0xFF, (size_t)ix->rs * 8, c->stream));
    HIP_TRY(hipMemsetAsync(d_missing, 0, 16, c->stream));
    cid::SearchParams p;
    rc = fill_search_params(c, ix, p);
    if (rc) return rc;
    p.kmers = d_k; p.codes = d_codes; p.n_kmers = n_kmers; p.and_words = d_and; p.missing = d_missing;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
    HIP_TRY(cid::launch_search_perfect(p, c->stream));
    return CID_OK;
}
extern "C" {

static int search_perfect_to_host(cid_ctx *c, const cid_index *ix, const uint8_t *d_k, const uint64_t *d_codes, size_t n_kmers,
                                  uint32_t *and_words_le, int *any_row_missing) {
    void *d_out;
    int rc = slot_reserve(c, S_MISC, (size_t)ix->rs * 8 + 16, &d_out);
    if (rc) return rc;
    uint64_t *d_and = (uint64_t *)d_out;
    int *d_missing = (int *)(d_and + ix->rs);
    rc = cid::search_perfect_launch(c, ix, d_k, d_codes, n_kmers, d_and, d_missing);
    if (rc) return rc;
    std::vector<uint64_t> h(ix->rs);
    int missing = 0;
    HIP_TRY(hipMemcpyAsync(h.data(), d_and, (size_t)ix->rs * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&missing, d_missing, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *any_row_missing = missing ? 1 : 0;
    for (uint32_t w = 0; w < ix->w32; ++w) {
        const uint32_t v = (uint32_t)(h[w / 2] >> (32 * (w & 1)));
        and_words_le[w] = missing ? 0u : v;
    }
    return CID_OK;
}

int cid_search_perfect(cid_ctx *c, const cid_index *ix, const uint8_t *kmers, size_t n_kmers, uint32_t *and_words_le,
                       int *any_row_missing) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    if (n_kmers == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    HIP_TRY(hipSetDevice(c->device));
    void *d_k;
    rc = slot_reserve(c, S_KMERS, n_kmers * ix->k, &d_k);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_k, kmers, n_kmers * ix->k, hipMemcpyHostToDevice, c->stream));
    return search_perfect_to_host(c, ix, (const uint8_t *)d_k, nullptr, n_kmers, and_words_le, any_row_missing);
}

extern "C++" {
namespace cid {
int search_count_ascii(cid_ctx *c, const cid_index *ix, const uint8_t *d_ascii, const uint32_t *d_counts, size_t n, uint32_t k,
                       uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    HIP_TRY(hipSetDevice(c->device));
    return search_count_to_host(c, ix, d_ascii, nullptr, d_counts, n, hits, n_unique, sum_unique_freq, unique_colour);
}
int search_perfect_ascii(cid_ctx *c, const cid_index *ix, const uint8_t *d_ascii, size_t n, uint32_t k, uint32_t *and_words_le, int *any_row_missing) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing) return fail(CID_ERR_INVALID, "null argument");
    if (n == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    HIP_TRY(hipSetDevice(c->device));
    return search_perfect_to_host(c, ix, d_ascii, nullptr, n, and_words_le, any_row_missing);
}
int search_count_codes(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, const uint32_t *d_counts, size_t n, uint32_t k,
                       uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    HIP_TRY(hipSetDevice(c->device));
    return search_count_to_host(c, ix, nullptr, d_codes, d_counts, n, hits, n_unique, sum_unique_freq, unique_colour);
}
int search_perfect_codes(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, size_t n, uint32_t k, uint32_t *and_words_le,
                         int *any_row_missing) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing) return fail(CID_ERR_INVALID, "null argument");
    if (n == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    HIP_TRY(hipSetDevice(c->device));
    return search_perfect_to_host(c, ix, nullptr, d_codes, n, and_words_le, any_row_missing);
}
}  // namespace cid
}  // extern "C++"

}  // extern "C"
