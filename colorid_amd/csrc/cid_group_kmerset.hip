// Canonical k-mer counting over the GPUs of a group (SURVEY.md §8e.1 caveat: the reference counts DISTINCT k-mers over the whole
// query — "shard after global dedup, or shard by k-mer so that duplicates meet on one GPU"; §8f.1).  With one cid_kmerset on rank 0
// the counting of a read set is the serial part of `colorid search --gpus N` (12 ms per million reads against 10 ms / N of search)
// and the set has to fit one GPU.  Here every rank counts the windows of its share of the sequences, then the code space is cut
// into N ranges at splitters taken from the ranks' own quantiles, every rank sends each range to its owner (peer copies over xGMI:
// the one exchange step of this path, 12 bytes per locally-distinct k-mer) and merges what it receives (radix sort + reduce-by-key).
// Afterwards rank r holds the r-th range of the global set, ascending: the ranks' parts laid end to end ARE the set in the order a
// single-GPU cid_kmerset has.  The searches over the parts need no further exchange than the 3*C counters.
// k <= 32 (2-bit codes); larger k and case-keeping inputs with lower-case bases stay on one GPU / the host as before.
#include "cid_group.hpp"

#include <algorithm>
#include <new>

using cid::fail;
using namespace cid::slots;
using namespace cidg;

#define HIP_TRY(expr) CIDG_HIP_TRY(expr)

struct cid_group_kmerset {
    cid_group *g = nullptr;
    uint32_t k = 0;
    std::vector<cid_kmerset *> part;
    bool finalized = false;
};

namespace {

// out[i] = codes[(2 i + 1) n / (2 S)], i < S: S evenly spaced quantiles of an ascending array
__global__ void k_sample_quantiles(const uint64_t *codes, uint64_t n, uint32_t S, uint64_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) out[i] = codes[(uint64_t)((2ull * i + 1ull) * n / (2ull * S))];
}
// bounds[j] = first index with codes[index] >= splitters[j] (lower bound), j < n_split
__global__ void k_lower_bounds(const uint64_t *codes, uint64_t n, const uint64_t *splitters, uint32_t n_split, uint64_t *bounds) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_split) return;
    const uint64_t key = splitters[j];
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (codes[mid] < key) lo = mid + 1; else hi = mid;
    }
    bounds[j] = lo;
}

struct View { cid_ctx *c; const uint64_t *codes; const uint32_t *counts; uint64_t n; };

int view_of(const cid_kmerset *ks, View &v) {
    uint32_t k;
    return cid::kmerset_view(ks, &v.c, &v.codes, &v.counts, &v.n, &k);
}

int check_set(const cid_group_kmerset *s, bool want_final) {
    if (!s) return fail(CID_ERR_INVALID, "null group k-mer set");
    if (want_final && !s->finalized) return fail(CID_ERR_STATE, "group k-mer set not finalized");
    if (!want_final && s->finalized) return fail(CID_ERR_STATE, "group k-mer set already finalized");
    return CID_OK;
}

}  // namespace

extern "C" {

int cid_group_kmerset_create(cid_group *g, uint32_t k_size, cid_group_kmerset **out) {
    if (!g || !out) return fail(CID_ERR_INVALID, "null argument");
    *out = nullptr;
    if (k_size == 0 || k_size > 32) return fail(CID_ERR_UNSUPPORTED,
                                                "the group k-mer set packs k-mers into 2-bit codes: k_size 1..32 (count larger k on one GPU)");
    cid_group_kmerset *s = new (std::nothrow) cid_group_kmerset();
    if (!s) return fail(CID_ERR_NOMEM, "group k-mer set");
    s->g = g; s->k = k_size;
    s->part.assign(g->ctx.size(), nullptr);
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        const int rc = cid_kmerset_create(g->ctx[r], k_size, &s->part[r]);
        if (rc) { const std::string m = cid_last_error(); cid_group_kmerset_destroy(s); return fail(rc, "rank %zu: %s", r, m.c_str()); }
    }
    *out = s;
    return CID_OK;
}

void cid_group_kmerset_destroy(cid_group_kmerset *s) {
    if (!s) return;
    for (cid_kmerset *p : s->part) cid_kmerset_destroy(p);
    delete s;
}

// the sequences of one call are dealt to the ranks in contiguous, balanced shards; every rank extracts and (when its buffer fills)
// merges the windows of its own shard.  CID_ERR_UNSUPPORTED (a lower-case base under mode 1) from any rank fails the call.
int cid_group_kmerset_add_seqs(cid_group_kmerset *s, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs, int mode) {
    int rc = check_set(s, false);
    if (rc) return rc;
    if (n_seqs == 0) return CID_OK;
    if (!seq_off || (seq_off[n_seqs] && !bases)) return fail(CID_ERR_INVALID, "null argument");
    const int n = (int)s->part.size();
    std::vector<int> rcs(n, CID_OK);
    rc = for_each_rank(s->g, [&](int r) -> int {
        size_t lo, hi;
        shard_bounds(n_seqs, r, n, &lo, &hi);
        if (hi == lo) return CID_OK;
        std::vector<uint64_t> so(hi - lo + 1);
        for (size_t i = 0; i <= hi - lo; ++i) {
            if (seq_off[lo + i] < seq_off[lo]) return fail(CID_ERR_INVALID, "seq_off not monotonic");
            so[i] = seq_off[lo + i] - seq_off[lo];
        }
        rcs[r] = cid_kmerset_add_seqs(s->part[r], bases + seq_off[lo], so.data(), hi - lo, mode);
        return rcs[r];
    });
    for (int r = 0; r < n; ++r)
        if (rcs[r] == CID_ERR_UNSUPPORTED) return CID_ERR_UNSUPPORTED;   // (the message is the rank's)
    return rc;
}

int cid_group_kmerset_finalize(cid_group_kmerset *s, uint64_t *n_distinct) {
    if (!s) return fail(CID_ERR_INVALID, "null group k-mer set");
    cid_group *g = s->g;
    const int n = (int)s->part.size();
    if (!s->finalized) {
        int rc = for_each_rank(g, [&](int r) { return cid_kmerset_finalize(s->part[r], nullptr); });   // local dedup
        if (rc) return rc;
        if (n > 1) {
            std::vector<View> v(n);
            for (int r = 0; r < n; ++r) if ((rc = view_of(s->part[r], v[r]))) return rc;
            // splitters: 1024 quantiles of every rank's own (sorted) set, pooled and cut into n equal shares
            constexpr uint32_t S = 1024;
            std::vector<uint64_t> pool;
            for (int r = 0; r < n; ++r) {
                if (v[r].n == 0) continue;
                cid_ctx *c = v[r].c;
                HIP_TRY(hipSetDevice(c->device));
                void *d_q;
                if ((rc = cid::slot_reserve(c, S_MISC, S * 8 + 64, &d_q))) return rc;
                hipLaunchKernelGGL(k_sample_quantiles, dim3((S + 255) / 256), dim3(256), 0, c->stream, v[r].codes, v[r].n, S, (uint64_t *)d_q);
                HIP_TRY(hipGetLastError());
                const size_t at = pool.size();
                pool.resize(at + S);
                HIP_TRY(hipMemcpyAsync(pool.data() + at, d_q, S * 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            std::sort(pool.begin(), pool.end());
            std::vector<uint64_t> split(n - 1, ~0ull);   // range j = [split[j-1], split[j]); an empty pool leaves everything on rank 0
            for (int j = 0; j + 1 < n && !pool.empty(); ++j) split[j] = pool[(size_t)(j + 1) * pool.size() / (size_t)n];
            // every rank: where its set crosses the splitters
            std::vector<std::vector<uint64_t>> bound(n, std::vector<uint64_t>(n + 1, 0));
            rc = for_each_rank(g, [&](int r) -> int {
                cid_ctx *c = v[r].c;
                bound[r][n] = v[r].n;
                if (v[r].n == 0) return CID_OK;
                HIP_TRY(hipSetDevice(c->device));
                void *d_m;
                const int e = cid::slot_reserve(c, S_MISC, (size_t)(n - 1) * 16 + 64, &d_m); if (e) return e;
                uint64_t *d_split = (uint64_t *)d_m, *d_bound = d_split + (n - 1);
                HIP_TRY(hipMemcpyAsync(d_split, split.data(), (size_t)(n - 1) * 8, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(k_lower_bounds, dim3(1), dim3(64), 0, c->stream, v[r].codes, v[r].n, d_split, (uint32_t)(n - 1), d_bound);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(bound[r].data() + 1, d_bound, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                return CID_OK;
            });
            if (rc) return rc;
            // the exchange: rank j collects range j of every rank (its own included) into one buffer, then merges
            std::vector<void *> in_codes(n, nullptr), in_counts(n, nullptr);
            std::vector<size_t> total(n, 0);
            rc = for_each_rank(g, [&](int j) -> int {
                cid_ctx *c = g->ctx[j];
                HIP_TRY(hipSetDevice(c->device));
                for (int r = 0; r < n; ++r) total[j] += (size_t)(bound[r][j + 1] - bound[r][j]);
                int e = cid::ctx_alloc(c, (total[j] ? total[j] : 1) * 8, &in_codes[j]); if (e) return e;
                e = cid::ctx_alloc(c, (total[j] ? total[j] : 1) * 4, &in_counts[j]); if (e) return e;
                size_t at = 0;
                for (int r = 0; r < n; ++r) {
                    const size_t len = (size_t)(bound[r][j + 1] - bound[r][j]);
                    if (!len) continue;
                    const uint64_t *sc = v[r].codes + bound[r][j];
                    const uint32_t *sn = v[r].counts + bound[r][j];
                    if (v[r].c->device == c->device) {
                        HIP_TRY(hipMemcpyAsync((uint64_t *)in_codes[j] + at, sc, len * 8, hipMemcpyDeviceToDevice, c->stream));
                        HIP_TRY(hipMemcpyAsync((uint32_t *)in_counts[j] + at, sn, len * 4, hipMemcpyDeviceToDevice, c->stream));
                    } else {
                        HIP_TRY(hipMemcpyPeerAsync((uint64_t *)in_codes[j] + at, c->device, sc, v[r].c->device, len * 8, c->stream));
                        HIP_TRY(hipMemcpyPeerAsync((uint32_t *)in_counts[j] + at, c->device, sn, v[r].c->device, len * 4, c->stream));
                    }
                    at += len;
                }
                HIP_TRY(hipStreamSynchronize(c->stream));
                return CID_OK;
            });
            // (all copies have landed before any rank's old arrays are replaced)
            if (rc == CID_OK)
                rc = for_each_rank(g, [&](int j) { return cid::kmerset_assign_merged(s->part[j], (const uint64_t *)in_codes[j],
                    (const uint32_t *)in_counts[j], total[j]); });
            for (int j = 0; j < n; ++j) {
                if (in_codes[j]) cid::ctx_free(g->ctx[j], in_codes[j]);
                if (in_counts[j]) cid::ctx_free(g->ctx[j], in_counts[j]);
            }
            if (rc) return rc;
        }
        s->finalized = true;
    }
    if (n_distinct) return cid_group_kmerset_size(s, n_distinct);
    return CID_OK;
}

int cid_group_kmerset_size(const cid_group_kmerset *s, uint64_t *n_distinct) {
    if (!s || !n_distinct) return fail(CID_ERR_INVALID, "null argument");
    *n_distinct = 0;
    for (const cid_kmerset *p : s->part) {
        uint64_t np = 0;
        const int rc = cid_kmerset_size(p, &np);
        if (rc) return rc;
        *n_distinct += np;
    }
    return CID_OK;
}

int cid_group_kmerset_part_sizes(const cid_group_kmerset *s, uint64_t *sizes) {
    if (!s || !sizes) return fail(CID_ERR_INVALID, "null argument");
    for (size_t r = 0; r < s->part.size(); ++r) {
        const int rc = cid_kmerset_size(s->part[r], &sizes[r]);
        if (rc) return rc;
    }
    return CID_OK;
}

// (multiplicity, number of k-mers) pairs over the whole set, ascending multiplicity: the ranks' histograms added up
int cid_group_kmerset_count_histogram(const cid_group_kmerset *s, uint32_t *multiplicity, uint64_t *n_kmers, size_t cap, size_t *n_bins) {
    int rc = check_set(s, true);
    if (rc) return rc;
    if (!n_bins) return fail(CID_ERR_INVALID, "null argument");
    std::vector<std::pair<uint32_t, uint64_t>> all;
    for (const cid_kmerset *p : s->part) {
        size_t nb = 0;
        if ((rc = cid_kmerset_count_histogram(p, nullptr, nullptr, 0, &nb))) return rc;
        std::vector<uint32_t> m(nb);
        std::vector<uint64_t> c(nb);
        if (nb && (rc = cid_kmerset_count_histogram(p, m.data(), c.data(), nb, &nb))) return rc;
        for (size_t i = 0; i < nb; ++i) all.emplace_back(m[i], c[i]);
    }
    std::sort(all.begin(), all.end());
    std::vector<std::pair<uint32_t, uint64_t>> merged;
    for (const auto &e : all) {
        if (!merged.empty() && merged.back().first == e.first) merged.back().second += e.second;
        else merged.push_back(e);
    }
    *n_bins = merged.size();
    if (!multiplicity || !n_kmers) return CID_OK;
    if (cap < merged.size()) return fail(CID_ERR_INVALID, "histogram needs %zu bins", merged.size());
    for (size_t i = 0; i < merged.size(); ++i) { multiplicity[i] = merged[i].first; n_kmers[i] = merged[i].second; }
    return CID_OK;
}

int cid_group_kmerset_clean(cid_group_kmerset *s, uint64_t t) {
    const int rc = check_set(s, true);
    if (rc) return rc;
    return for_each_rank(s->g, [&](int r) { return cid_kmerset_clean(s->part[r], t); });
}

int cid_group_kmerset_download(const cid_group_kmerset *s, uint8_t *kmers_ascii, uint32_t *counts) {
    const int rc = check_set(s, true);
    if (rc) return rc;
    std::vector<uint64_t> off(s->part.size() + 1, 0);
    for (size_t r = 0; r < s->part.size(); ++r) { uint64_t np = 0; cid_kmerset_size(s->part[r], &np); off[r + 1] = off[r] + np; }
    return for_each_rank(s->g, [&](int r) {
        return cid_kmerset_download(s->part[r], kmers_ascii ? kmers_ascii + off[r] * s->k : nullptr, counts ? counts + off[r] : nullptr);
    });
}

// a5 over the parts: every rank searches its own range against its replica (nothing moves), the 3*C counters are all-reduced;
// unique_colour in global set order
int cid_group_search_count_parts(cid_group *g, cid_index *const *replicas, const cid_group_kmerset *s, uint64_t *hits, uint64_t *n_unique,
                                 uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if ((rc = check_set(s, true))) return rc;
    if (s->g != g) return fail(CID_ERR_INVALID, "the k-mer set belongs to another group");
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    if (s->k != replicas[0]->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", s->k, replicas[0]->k);
    const int n = (int)g->ctx.size();
    const size_t C = replicas[0]->n_colors;
    std::vector<uint64_t> off(n + 1, 0);
    for (int r = 0; r < n; ++r) { uint64_t np = 0; cid_kmerset_size(s->part[r], &np); off[r + 1] = off[r] + np; }
    std::vector<uint64_t *> d_out(n, nullptr);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        View v;
        int e = view_of(s->part[r], v); if (e) return e;
        HIP_TRY(hipSetDevice(c->device));
        void *d_o, *d_uc = nullptr;
        e = cid::slot_reserve(c, S_OUT, 3 * C * 8, &d_o); if (e) return e;
        if (unique_colour) { e = cid::slot_reserve(c, S_UC, (v.n ? v.n : 1) * 4, &d_uc); if (e) return e; }
        uint64_t *o = (uint64_t *)d_o;
        e = cid::search_count_launch(c, replicas[r], nullptr, v.codes, v.counts, v.n, o, n_unique ? o + C : nullptr, sum_unique_freq ? o + 2 * C : nullptr,
                                     (uint32_t *)d_uc);
        if (e) return e;
        if (!n_unique) HIP_TRY(hipMemsetAsync(o + C, 0, C * 8, c->stream));           // the all-reduce covers all 3*C words
        if (!sum_unique_freq) HIP_TRY(hipMemsetAsync(o + 2 * C, 0, C * 8, c->stream));
        if (unique_colour && v.n) HIP_TRY(hipMemcpyAsync(unique_colour + off[r], d_uc, v.n * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        d_out[r] = o;
        return CID_OK;
    });
    if (rc) return rc;
    if ((rc = allreduce_sum(g, reinterpret_cast<void *const *>(d_out.data()), 3 * C, 8))) return rc;
    cid_ctx *c0 = g->ctx[0];
    HIP_TRY(hipSetDevice(c0->device));
    HIP_TRY(hipMemcpyAsync(hits, d_out[0], C * 8, hipMemcpyDeviceToHost, c0->stream));
    if (n_unique) HIP_TRY(hipMemcpyAsync(n_unique, d_out[0] + C, C * 8, hipMemcpyDeviceToHost, c0->stream));
    if (sum_unique_freq) HIP_TRY(hipMemcpyAsync(sum_unique_freq, d_out[0] + 2 * C, C * 8, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    return CID_OK;
}

// The same with everything reports::generate_report prints and nothing per k-mer (the group form of cid_search_count_set_report):
// every rank also reduces its part's unique hits to a (colour, multiplicity) histogram on the device; the histograms add up on the
// host and give the mode per colour (ties -> the smallest multiplicity, as on one GPU).
int cid_group_search_count_parts_report(cid_group *g, cid_index *const *replicas, const cid_group_kmerset *s, uint64_t *hits, uint64_t *n_unique,
                                        uint64_t *sum_unique_freq, uint64_t *mode_unique_freq) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if ((rc = check_set(s, true))) return rc;
    if (s->g != g) return fail(CID_ERR_INVALID, "the k-mer set belongs to another group");
    if (!hits || !n_unique || !sum_unique_freq || !mode_unique_freq) return fail(CID_ERR_INVALID, "null argument");
    if (s->k != replicas[0]->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", s->k, replicas[0]->k);
    const int n = (int)g->ctx.size();
    const size_t C = replicas[0]->n_colors;
    std::vector<uint64_t *> d_out(n, nullptr);
    std::vector<std::vector<uint64_t>> keys(n);
    std::vector<std::vector<uint32_t>> cnts(n);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        View v;
        int e = view_of(s->part[r], v); if (e) return e;
        HIP_TRY(hipSetDevice(c->device));
        void *d_o, *d_uc;
        e = cid::slot_reserve(c, S_OUT, 3 * C * 8, &d_o); if (e) return e;
        e = cid::slot_reserve(c, S_UC, (v.n ? v.n : 1) * 4, &d_uc); if (e) return e;
        uint64_t *o = (uint64_t *)d_o;
        e = cid::search_count_launch(c, replicas[r], nullptr, v.codes, v.counts, v.n, o, o + C, o + 2 * C, (uint32_t *)d_uc);
        if (e) return e;
        e = cid::unique_freq_hist(c, (const uint32_t *)d_uc, v.counts, v.n, keys[r], cnts[r]);
        d_out[r] = o;
        return e;
    });
    if (rc) return rc;
    if ((rc = allreduce_sum(g, reinterpret_cast<void *const *>(d_out.data()), 3 * C, 8))) return rc;
    cid_ctx *c0 = g->ctx[0];
    HIP_TRY(hipSetDevice(c0->device));
    HIP_TRY(hipMemcpyAsync(hits, d_out[0], C * 8, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipMemcpyAsync(n_unique, d_out[0] + C, C * 8, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipMemcpyAsync(sum_unique_freq, d_out[0] + 2 * C, C * 8, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    // merge the ranks' sorted histograms; per colour the multiplicity with the most k-mers (keys ascend, so the first maximum is the smallest)
    std::vector<std::pair<uint64_t, uint64_t>> all;
    for (int r = 0; r < n; ++r)
        for (size_t i = 0; i < keys[r].size(); ++i) all.emplace_back(keys[r][i], (uint64_t)cnts[r][i]);
    std::sort(all.begin(), all.end());
    for (size_t c = 0; c < C; ++c) mode_unique_freq[c] = 0;
    std::vector<uint64_t> best(C, 0);
    for (size_t i = 0; i < all.size();) {
        size_t j = i;
        uint64_t tot = 0;
        while (j < all.size() && all[j].first == all[i].first) tot += all[j++].second;
        const uint64_t col = all[i].first >> 32, f = all[i].first & 0xFFFFFFFFull;
        if (col < C && tot > best[col]) { best[col] = tot; mode_unique_freq[col] = f; }
        i = j;
    }
    return CID_OK;
}

// a4 over the parts: AND of the ranks' words on the host (empty parts are neutral)
int cid_group_search_perfect_parts(cid_group *g, cid_index *const *replicas, const cid_group_kmerset *s, uint32_t *and_words_le, int *any_row_missing) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if ((rc = check_set(s, true))) return rc;
    if (s->g != g) return fail(CID_ERR_INVALID, "the k-mer set belongs to another group");
    if (!and_words_le || !any_row_missing) return fail(CID_ERR_INVALID, "null argument");
    if (s->k != replicas[0]->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", s->k, replicas[0]->k);
    uint64_t nk = 0;
    cid_group_kmerset_size(s, &nk);
    if (nk == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    const int n = (int)g->ctx.size();
    const uint32_t w32 = replicas[0]->w32;
    std::vector<std::vector<uint32_t>> words(n, std::vector<uint32_t>(w32, 0xFFFFFFFFu));
    std::vector<int> missing(n, 0);
    std::vector<uint64_t> np(n, 0);
    rc = for_each_rank(g, [&](int r) -> int {
        View v;
        const int e = view_of(s->part[r], v); if (e) return e;
        np[r] = v.n;
        if (v.n == 0) return CID_OK;
        return cid::search_perfect_codes(g->ctx[r], replicas[r], v.codes, v.n, s->k, words[r].data(), &missing[r]);
    });
    if (rc) return rc;
    int miss = 0;
    for (uint32_t w = 0; w < w32; ++w) and_words_le[w] = 0xFFFFFFFFu;
    for (int r = 0; r < n; ++r) {
        if (np[r] == 0) continue;
        miss |= missing[r];
        for (uint32_t w = 0; w < w32; ++w) and_words_le[w] &= words[r][w];
    }
    if (miss) for (uint32_t w = 0; w < w32; ++w) and_words_le[w] = 0;
    *any_row_missing = miss ? 1 : 0;
    return CID_OK;
}

}  // extern "C"
