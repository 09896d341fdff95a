// Several GPUs of one node (include/colorid_hip.h "cid_group"): N contexts — one per device — with a replica
// of the index on each; query k-mers / reads are sharded over the ranks in contiguous, balanced ranges (the reference's own
// parallel boundary is the rayon map over reads, src/read_id_mt_pe.rs:300-302, `-t`, src/main.rs:718-721).  The one exchange step
// of the proportional search is the sum of the 3*C per-accession counters: RCCL ncclAllReduce over xGMI, issued for all ranks
// from this process inside one ncclGroupStart/End (librccl is loaded on first use with dlopen: the library has no link-time
// dependency on it; the library never redirects a file descriptor — RCCL's version banner goes wherever RCCL writes it, at communicator
// creation only, and a host that prints result rows to stdout wraps cid_group_create itself as host/main.cpp does).  A device id may be listed more than once — several ranks on one GPU, which RCCL refuses — then, or with
// COLORID_REDUCE=host, the 24*C bytes per rank are summed through the host.  Perfect search: AND of the ranks' W words on the
// host (RCCL has no bitwise reduction).  read_id: no exchange; rows are concatenated in input order.
// Host code only: every kernel launch goes through the single-GPU entry points.
#include "cid_group.hpp"

#include <new>

using cid::fail;
using namespace cid::slots;
using namespace cidg;

#define HIP_TRY(expr) CIDG_HIP_TRY(expr)

namespace cidg {

int allreduce_sum(cid_group *g, void *const *d_bufs, size_t count, int elem_bytes) {
    const int n = (int)g->ctx.size();
    if (n == 1 && !g->use_rccl) return CID_OK;
    if (g->use_rccl) {
        int e = g->rccl.GroupStart();
        for (int r = 0; r < n && e == 0; ++r) {
            HIP_TRY(hipSetDevice(g->dev[r]));
            e = g->rccl.AllReduce(d_bufs[r], d_bufs[r], count, elem_bytes == 8 ? kNcclUint64 : kNcclUint32, kNcclSum, g->comms[r], g->ctx[r]->stream);
        }
        const int e2 = g->rccl.GroupEnd();
        if (e || e2) return fail(CID_ERR_HIP, "ncclAllReduce: %s", g->rccl.GetErrorString(e ? e : e2));
        for (int r = 0; r < n; ++r) { HIP_TRY(hipSetDevice(g->dev[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
        return CID_OK;
    }
    const size_t bytes = count * (size_t)elem_bytes;
    std::vector<uint8_t> total(bytes, 0), part(bytes);
    for (int r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipMemcpyAsync(part.data(), d_bufs[r], bytes, hipMemcpyDeviceToHost, g->ctx[r]->stream));
        HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream));
        if (elem_bytes == 8) {
            uint64_t *t = reinterpret_cast<uint64_t *>(total.data()); const uint64_t *p = reinterpret_cast<const uint64_t *>(part.data());
            for (size_t i = 0; i < count; ++i) t[i] += p[i];
        } else {
            uint32_t *t = reinterpret_cast<uint32_t *>(total.data()); const uint32_t *p = reinterpret_cast<const uint32_t *>(part.data());
            for (size_t i = 0; i < count; ++i) t[i] += p[i];
        }
    }
    for (int r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipMemcpyAsync(d_bufs[r], total.data(), bytes, hipMemcpyHostToDevice, g->ctx[r]->stream));
        HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream));
    }
    return CID_OK;
}

}  // namespace cidg

namespace {

int allreduce_u64(cid_group *g, uint64_t *const *d_bufs, size_t count) {
    return cidg::allreduce_sum(g, reinterpret_cast<void *const *>(d_bufs), count, 8);
}

// rank 0's counters -> the caller's host arrays
int counters_to_host(cid_group *g, const uint64_t *d0, size_t C, uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq) {
    cid_ctx *c = g->ctx[0];
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(hits, d0, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_unique) HIP_TRY(hipMemcpyAsync(n_unique, d0 + C, C * 8, hipMemcpyDeviceToHost, c->stream));
    if (sum_unique_freq) HIP_TRY(hipMemcpyAsync(sum_unique_freq, d0 + 2 * C, C * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

}  // namespace

extern "C" {

int cid_group_create(const int *device_ids, int n_devices, cid_group **out) {
    if (!out) return fail(CID_ERR_INVALID, "null out");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return fail(CID_ERR_INVALID, "1..64 device ids expected");
    cid_group *g = new (std::nothrow) cid_group();
    if (!g) return fail(CID_ERR_NOMEM, "group");
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
        cid_ctx *c = nullptr;
        const int rc = cid_ctx_create(device_ids[i], &c);
        if (rc) { cid_group_destroy(g); return rc; }
        g->ctx.push_back(c);
        g->dev.push_back(device_ids[i]);
    }
    // RCCL when every rank has its own GPU (one rank too, if asked for: COLORID_REDUCE=rccl exercises the plumbing on one GPU)
    const int mode = g->ctx[0]->tune.reduce_mode;   // COLORID_REDUCE, read when the ranks' contexts were made: -1 auto, 0 host, 1 rccl
    const bool want = mode >= 0 ? mode == 1 : n_devices > 1;
    if (want && mode == 1 && !distinct) {
        cid_group_destroy(g);
        return fail(CID_ERR_INVALID, "COLORID_REDUCE=rccl needs distinct devices");
    }
    if (want && distinct) {
        if (!g->rccl.load()) {
            cid_group_destroy(g);
            return fail(CID_ERR_HIP, "cannot load librccl.so (set COLORID_REDUCE=host to sum through the host): %s", dlerror());
        }
        g->comms.assign(n_devices, nullptr);
        const int e = g->rccl.CommInitAll(g->comms.data(), n_devices, device_ids);
        if (e) { g->comms.clear(); const char *m = g->rccl.GetErrorString(e); cid_group_destroy(g); return fail(CID_ERR_HIP, "ncclCommInitAll: %s", m); }
        g->use_rccl = true;
    }
    g->sp_rows.assign(n_devices, 0);
    g->sp_entries.assign(n_devices, 0);
    *out = g;
    return CID_OK;
}

int cid_group_size(const cid_group *g, int *n_ranks) {
    if (!g || !n_ranks) return fail(CID_ERR_INVALID, "null argument");
    *n_ranks = (int)g->ctx.size();
    return CID_OK;
}

int cid_group_ctx(cid_group *g, int rank, cid_ctx **out) {
    if (!g || !out || rank < 0 || rank >= (int)g->ctx.size()) return fail(CID_ERR_INVALID, "bad rank");
    *out = g->ctx[rank];
    return CID_OK;
}

int cid_group_uses_rccl(const cid_group *g, int *yes) {
    if (!g || !yes) return fail(CID_ERR_INVALID, "null argument");
    *yes = g->use_rccl ? 1 : 0;
    return CID_OK;
}

void cid_group_destroy(cid_group *g) {
    if (!g) return;
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r]) { (void)hipSetDevice(g->dev[r]); (void)g->rccl.CommDestroy(g->comms[r]); }
    for (size_t r = 0; r < g->ev_ready.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (g->ev_ready[r]) (void)hipEventDestroy(g->ev_ready[r]);
        if (g->ev_reduced[r]) (void)hipEventDestroy(g->ev_reduced[r]);
    }
    for (cid_ctx *c : g->ctx) cid_ctx_destroy(c);
    delete g;
}

// One replica per rank of a finalized index: the source itself on the rank whose ctx owns it, device-to-device copies elsewhere
// (xGMI peer copies between GPUs; a plain copy for a second rank on the same GPU).
int cid_group_replicate_index(cid_group *g, cid_index *src, cid_index **replicas) {
    if (!g || !src || !replicas) return fail(CID_ERR_INVALID, "null argument");
    if (!src->finalized) return fail(CID_ERR_STATE, "index not finalized");
    const size_t bytes = (size_t)src->m * src->rs * 8;
    for (size_t r = 0; r < g->ctx.size(); ++r) replicas[r] = nullptr;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        cid_ctx *c = g->ctx[r];
        if (c == src->ctx) { replicas[r] = src; continue; }
        cid_index *ix = new (std::nothrow) cid_index(*src);
        if (!ix) return fail(CID_ERR_NOMEM, "index");
        ix->ctx = c;
        ix->mat = nullptr;
        replicas[r] = ix;
        HIP_TRY(hipSetDevice(c->device));
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ix->mat), bytes);
        if (e != hipSuccess) return fail(CID_ERR_NOMEM, "hipMalloc(%zu) for replica %zu: %s", bytes, r, hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(src->ctx->stream));
        if (c->device == src->ctx->device) HIP_TRY(hipMemcpyAsync(ix->mat, src->mat, bytes, hipMemcpyDeviceToDevice, c->stream));
        else HIP_TRY(hipMemcpyPeerAsync(ix->mat, c->device, src->mat, src->ctx->device, bytes, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CID_OK;
}

// a5 over the group, host k-mers: rank r takes k-mers [lo_r, hi_r)
int cid_group_search_count(cid_group *g, cid_index *const *replicas, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                           uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if (!hits || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    const int n = (int)g->ctx.size();
    const size_t k = replicas[0]->k, C = replicas[0]->n_colors;
    const bool want_unique = n_unique || sum_unique_freq || unique_colour;
    std::vector<uint64_t *> d_out(n, nullptr);
    rc = for_each_rank(g, [&](int r) {
        size_t lo, hi;
        shard_bounds(n_kmers, r, n, &lo, &hi);
        // a shard must start on a 16-byte boundary of the k-mer array for the device loads: shards are re-based copies on the device,
        // so only the host pointer moves
        return cid::search_count_host_input(g->ctx[r], replicas[r], kmers + lo * k, freq ? freq + lo : nullptr, hi - lo, want_unique,
                                            unique_colour ? unique_colour + lo : nullptr, &d_out[r]);
    });
    if (rc) return rc;
    if ((rc = allreduce_u64(g, d_out.data(), 3 * C))) return rc;
    return counters_to_host(g, d_out[0], C, hits, n_unique, sum_unique_freq);
}

// a5 over the group for a finalized k-mer set (global dedup done, SURVEY.md §8e.1 caveat): the set's [lo_r, hi_r) slice goes to rank r
// device-to-device; results in set order
int cid_group_search_count_set(cid_group *g, cid_index *const *replicas, const cid_kmerset *ks, uint64_t *hits, uint64_t *n_unique,
                               uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    cid_ctx *kc;
    const uint64_t *codes = nullptr;
    const uint8_t *ascii = nullptr;     // byte-string sets (k > 32): n x k bytes instead of 2-bit codes
    const uint32_t *counts;
    uint64_t nk;
    uint32_t kk;
    if (cid::kmerset_view_ascii(ks, &kc, &ascii, &counts, &nk, &kk) != CID_OK) {
        ascii = nullptr;
        if ((rc = cid::kmerset_view(ks, &kc, &codes, &counts, &nk, &kk))) return rc;
    }
    if (kk != replicas[0]->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", kk, replicas[0]->k);
    const int n = (int)g->ctx.size();
    const size_t C = replicas[0]->n_colors;
    const size_t unit = ascii ? kk : 8;   // bytes per k-mer in the set
    HIP_TRY(hipSetDevice(kc->device));
    HIP_TRY(hipStreamSynchronize(kc->stream));
    std::vector<uint64_t *> d_out(n, nullptr);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        size_t lo, hi;
        if (ascii) shard_bounds64(nk, r, n, &lo, &hi); else shard_bounds(nk, r, n, &lo, &hi);
        const size_t ns = hi - lo;
        HIP_TRY(hipSetDevice(c->device));
        void *d_o, *d_uc = nullptr, *d_k = nullptr, *d_f = nullptr;
        int e = cid::slot_reserve(c, S_OUT, 3 * C * 8, &d_o); if (e) return e;
        if (unique_colour) { e = cid::slot_reserve(c, S_UC, ns * 4, &d_uc); if (e) return e; }
        const uint8_t *my_keys = ascii ? ascii + lo * unit : reinterpret_cast<const uint8_t *>(codes + lo);
        const uint32_t *my_counts = counts + lo;
        if (c->device != kc->device) {   // another GPU: the slice travels over xGMI
            e = cid::slot_reserve(c, S_KMERS, ns * unit, &d_k); if (e) return e;
            e = cid::slot_reserve(c, S_FREQ, ns * 4, &d_f); if (e) return e;
            if (ns) {
                HIP_TRY(hipMemcpyPeerAsync(d_k, c->device, my_keys, kc->device, ns * unit, c->stream));
                HIP_TRY(hipMemcpyPeerAsync(d_f, c->device, counts + lo, kc->device, ns * 4, c->stream));
            }
            my_keys = (const uint8_t *)d_k; my_counts = (const uint32_t *)d_f;
        }
        uint64_t *o = (uint64_t *)d_o;
        e = cid::search_count_launch(c, replicas[r], ascii ? my_keys : nullptr, ascii ? nullptr : reinterpret_cast<const uint64_t *>(my_keys), my_counts, ns, o,
                                     n_unique ? o + C : nullptr, sum_unique_freq ? o + 2 * C : nullptr, (uint32_t *)d_uc);
        if (e) return e;
        if (!n_unique) HIP_TRY(hipMemsetAsync(o + C, 0, C * 8, c->stream));           // the all-reduce covers all 3*C words
        if (!sum_unique_freq) HIP_TRY(hipMemsetAsync(o + 2 * C, 0, C * 8, c->stream));
        if (unique_colour && ns) HIP_TRY(hipMemcpyAsync(unique_colour + lo, d_uc, ns * 4, hipMemcpyDeviceToHost, c->stream));
        d_out[r] = o;
        return CID_OK;
    });
    if (rc) return rc;
    if ((rc = allreduce_u64(g, d_out.data(), 3 * C))) return rc;
    for (int r = 1; r < n; ++r) { HIP_TRY(hipSetDevice(g->dev[r])); HIP_TRY(hipStreamSynchronize(g->ctx[r]->stream)); }
    return counters_to_host(g, d_out[0], C, hits, n_unique, sum_unique_freq);
}

// a4 over the group: every rank ANDs the rows of its shard; the W words and the absent-row flags are combined on the host
static int perfect_combine(cid_group *g, cid_index *const *replicas, const std::vector<std::vector<uint32_t>> &words, const std::vector<int> &missing,
                           const std::vector<size_t> &shard_n, uint32_t *and_words_le, int *any_row_missing) {
    const uint32_t w32 = replicas[0]->w32;
    int miss = 0;
    for (uint32_t w = 0; w < w32; ++w) and_words_le[w] = 0xFFFFFFFFu;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (shard_n[r] == 0) continue;   // an empty shard contributes the neutral element
        miss |= missing[r];
        for (uint32_t w = 0; w < w32; ++w) and_words_le[w] &= words[r][w];
    }
    if (miss) for (uint32_t w = 0; w < w32; ++w) and_words_le[w] = 0;
    *any_row_missing = miss ? 1 : 0;
    return CID_OK;
}

int cid_group_search_perfect(cid_group *g, cid_index *const *replicas, const uint8_t *kmers, size_t n_kmers, uint32_t *and_words_le,
                             int *any_row_missing) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    if (n_kmers == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    const int n = (int)g->ctx.size();
    const size_t k = replicas[0]->k;
    std::vector<std::vector<uint32_t>> words(n, std::vector<uint32_t>(replicas[0]->w32, 0xFFFFFFFFu));
    std::vector<int> missing(n, 0);
    std::vector<size_t> shard_n(n, 0);
    rc = for_each_rank(g, [&](int r) -> int {
        size_t lo, hi;
        shard_bounds(n_kmers, r, n, &lo, &hi);
        shard_n[r] = hi - lo;
        if (hi == lo) return CID_OK;
        return cid_search_perfect(g->ctx[r], replicas[r], kmers + lo * k, hi - lo, words[r].data(), &missing[r]);
    });
    if (rc) return rc;
    return perfect_combine(g, replicas, words, missing, shard_n, and_words_le, any_row_missing);
}

int cid_group_search_perfect_set(cid_group *g, cid_index *const *replicas, const cid_kmerset *ks, uint32_t *and_words_le, int *any_row_missing) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing) return fail(CID_ERR_INVALID, "null argument");
    cid_ctx *kc;
    const uint64_t *codes = nullptr;
    const uint8_t *ascii = nullptr;
    const uint32_t *counts;
    uint64_t nk;
    uint32_t kk;
    if (cid::kmerset_view_ascii(ks, &kc, &ascii, &counts, &nk, &kk) != CID_OK) {
        ascii = nullptr;
        if ((rc = cid::kmerset_view(ks, &kc, &codes, &counts, &nk, &kk))) return rc;
    }
    if (nk == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    if (kk != replicas[0]->k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", kk, replicas[0]->k);
    const int n = (int)g->ctx.size();
    HIP_TRY(hipSetDevice(kc->device));
    HIP_TRY(hipStreamSynchronize(kc->stream));
    std::vector<std::vector<uint32_t>> words(n, std::vector<uint32_t>(replicas[0]->w32, 0xFFFFFFFFu));
    std::vector<int> missing(n, 0);
    std::vector<size_t> shard_n(n, 0);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        size_t lo, hi;
        if (ascii) shard_bounds64(nk, r, n, &lo, &hi); else shard_bounds(nk, r, n, &lo, &hi);
        shard_n[r] = hi - lo;
        if (hi == lo) return CID_OK;
        HIP_TRY(hipSetDevice(c->device));
        const size_t unit = ascii ? kk : 8;
        const uint8_t *my_keys = ascii ? ascii + lo * unit : reinterpret_cast<const uint8_t *>(codes + lo);
        if (c->device != kc->device) {
            void *d_k;
            const int e = cid::slot_reserve(c, S_KMERS, (hi - lo) * unit, &d_k); if (e) return e;
            HIP_TRY(hipMemcpyPeerAsync(d_k, c->device, my_keys, kc->device, (hi - lo) * unit, c->stream));
            my_keys = (const uint8_t *)d_k;
        }
        if (ascii) return cid::search_perfect_ascii(c, replicas[r], my_keys, hi - lo, kk, words[r].data(), &missing[r]);
        return cid::search_perfect_codes(c, replicas[r], reinterpret_cast<const uint64_t *>(my_keys), hi - lo, kk, words[r].data(), &missing[r]);
    });
    if (rc) return rc;
    return perfect_combine(g, replicas, words, missing, shard_n, and_words_le, any_row_missing);
}

// a6-a10 over the group: reads [lo_r, hi_r) on rank r, each shard's sparse report kept in its ctx; fetch concatenates them in input order
int cid_group_readid_count_sparse(cid_group *g, cid_index *const *replicas, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                                  const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t *n_kmers,
                                  uint8_t *status, uint64_t *n_entries) {
    int rc = check_replicas(g, replicas);
    if (rc) return rc;
    if (!n_entries || !seq_off || !read_seq0 || (n_reads && (!n_kmers || !status))) return fail(CID_ERR_INVALID, "null argument");
    *n_entries = 0;
    if (n_reads && read_seq0[n_reads] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs");
    const int n = (int)g->ctx.size();
    g->sp_striped = false;
    rc = for_each_rank(g, [&](int r) -> int {
        size_t lo, hi;
        shard_bounds(n_reads, r, n, &lo, &hi);
        const size_t nr = hi - lo;
        g->sp_rows[r] = nr; g->sp_entries[r] = 0;
        // the shard re-based to its own offsets (every entry checked before seq_off is read through it)
        for (size_t i = lo; i < hi; ++i)
            if (read_seq0[i] > read_seq0[i + 1] || read_seq0[i + 1] > n_seqs) return fail(CID_ERR_INVALID,
                "read_seq0 not monotonic or past n_seqs at read %zu", i);
        const uint64_t s0 = read_seq0[lo], s1 = read_seq0[hi];
        std::vector<uint64_t> so(s1 - s0 + 1), r0(nr + 1);
        for (size_t i = 0; i <= s1 - s0; ++i) {
            if (seq_off[s0 + i] < seq_off[s0]) return fail(CID_ERR_INVALID, "seq_off not monotonic");
            so[i] = seq_off[s0 + i] - seq_off[s0];
        }
        for (size_t i = 0; i <= nr; ++i) r0[i] = read_seq0[lo + i] - s0;
        uint64_t ne = 0;
        const int e = cid_readid_count_sparse(g->ctx[r], replicas[r], bases ? bases + seq_off[s0] : nullptr, so.data(), s1 - s0, r0.data(), nr, stride_d,
                                              start_sample, n_kmers + lo, status + lo, &ne);
        g->sp_entries[r] = ne;
        return e;
    });
    if (rc) return rc;
    for (int r = 0; r < n; ++r) *n_entries += g->sp_entries[r];
    return CID_OK;
}

int cid_group_readid_sparse_fetch(cid_group *g, uint64_t *row_start, uint32_t *colours, uint32_t *counts) {
    if (!g || !row_start) return fail(CID_ERR_INVALID, "null argument");
    if (g->sp_striped) return cidg::stripes_sparse_fetch(g, row_start, colours, counts);
    uint64_t row = 0, ent = 0;
    row_start[0] = 0;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (g->sp_rows[r] == 0) continue;
        std::vector<uint64_t> rs(g->sp_rows[r] + 1);
        const int rc = cid_readid_sparse_fetch(g->ctx[r], rs.data(), colours ? colours + ent : nullptr, counts ? counts + ent : nullptr);
        if (rc) return rc;
        for (uint64_t i = 1; i <= g->sp_rows[r]; ++i) row_start[row + i] = ent + rs[i];
        row += g->sp_rows[r];
        ent += g->sp_entries[r];
    }
    return CID_OK;
}

}  // extern "C"
