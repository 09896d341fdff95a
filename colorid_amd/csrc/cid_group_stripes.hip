// Colour stripes over the GPUs of one node (include/colorid_hip.h "cid_group_stripes"): rank r of a cid_group holds
// the colours [base_r, base_{r+1}) of EVERY Bloom row — the placement for an index larger than one GPU's HBM (configs[4]: m = 2^30,
// 4096 colours = 512 GiB, 64 GiB per GPU).  Every rank sees every query k-mer / read; per-colour counts of a stripe are final; what
// needs all stripes is exchanged once per call:
//   search        : one u32 per k-mer (n << 26 | colour + 1, cid_search.hip) SUMMED over the ranks — RCCL ncclAllReduce over xGMI
//                   (a reduce-scatter of peer copies when device ids repeat) — then k_unique_finalize on rank 0;
//   perfect search: one u32 per k-mer (seeds whose row is all-zero in the stripe) ANDed over the ranks (RCCL has no bitwise
//                   reduction: ncclAllGather + a local AND on every rank; reduce-scatter of peer copies when device ids repeat),
//                   the stripes' AND words concatenated on the host;
//   read_id       : the zero pass's masks (one u32 per read and distinct k-mer) ANDed the same way, on every rank, then
//                   the count pass; every rank compacts its own columns, the host splices the ranks' (colour, count) lists per read.
// Host code + two elementwise kernels; every search / read_id kernel launch goes through the single-GPU stripe entry points.
#include "cid_group.hpp"

#include <new>

using cid::fail;
using namespace cid::slots;
using namespace cidg;

#define HIP_TRY(expr) CIDG_HIP_TRY(expr)

namespace {

// dst[i] (op)= src[j * stride + i] for j < n_src; ASSIGN: dst[i] = the fold of the n_src sources alone (dst's old value is one of them)
template <bool SUM, bool ASSIGN>
__global__ void k_fold_u32(uint32_t *dst, const uint32_t *src, uint32_t n_src, uint64_t stride, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = ASSIGN ? (SUM ? 0u : 0xFFFFFFFFu) : dst[i];
    for (uint32_t j = 0; j < n_src; ++j) {
        const uint32_t x = src[(uint64_t)j * stride + i];
        v = SUM ? v + x : v & x;
    }
    dst[i] = v;
}
// flag[0] |= 1 if any word has one of `mask`'s bits
__global__ void k_any_masked(const uint32_t *v, uint64_t n, uint32_t mask, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = i < n && (v[i] & mask);
    if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

struct Stripes {
    int n = 0;
    std::vector<uint32_t> base;   // n + 1
    uint32_t total = 0;
};

// the stripes' shapes must agree; their colour ranges are laid end to end in rank order (whole 64-colour words except the last)
int check_stripes(const cid_group *g, cid_index *const *stripes, Stripes &st) {
    if (!g || !stripes) return fail(CID_ERR_INVALID, "null group/stripes");
    st.n = (int)g->ctx.size();
    st.base.assign(st.n + 1, 0);
    for (int r = 0; r < st.n; ++r) {
        const int rc = cid::check_ready(g->ctx[r], stripes[r]);
        if (rc) return rc;
        if (stripes[r]->k != stripes[0]->k || stripes[r]->m != stripes[0]->m || stripes[r]->n_hash != stripes[0]->n_hash ||
            stripes[r]->mod.flags != stripes[0]->mod.flags)
            return fail(CID_ERR_INVALID, "stripe %d differs from stripe 0 in shape", r);
        if (stripes[r]->m_size) return fail(CID_ERR_UNSUPPORTED, "minimizer (.mxi) indexes are not striped");
        if (r + 1 < st.n && stripes[r]->n_colors % 64u) return fail(CID_ERR_INVALID, "stripe %d: %u colours, not whole 64-colour words", r,
                                                                    stripes[r]->n_colors);
        st.base[r + 1] = st.base[r] + stripes[r]->n_colors;
    }
    st.total = st.base[st.n];
    if (st.total > (1u << 20)) return fail(CID_ERR_UNSUPPORTED,
                                           "%u colours: the packed per-k-mer fact holds colour + 1 in 26 bits, stripes stop at 2^20", st.total);
    if (st.n > 31) return fail(CID_ERR_UNSUPPORTED, "%d ranks: the summed per-k-mer facts hold at most 31", st.n);
    return CID_OK;
}

hipError_t copy_between(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t stream) {
    return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev,
        bytes, stream);
}

int group_events(cid_group *g) {
    const size_t n = g->ctx.size();
    if (g->ev_ready.size() == n) return CID_OK;
    g->ev_ready.assign(n, nullptr);
    g->ev_reduced.assign(n, nullptr);
    for (size_t r = 0; r < n; ++r) {
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_ready[r], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g->ev_reduced[r], hipEventDisableTiming));
    }
    return CID_OK;
}

// AND over ranks with every rank on its own GPU: RCCL has no bitwise reduction, so every rank all-gathers the ranks' arrays (ring /
// direct over xGMI, RCCL's choice) 16 M words at a time and folds them locally — no rank is a funnel, every rank ends with the result
int and_by_allgather(cid_group *g, uint32_t *const *d_bufs, size_t count) {
    const int n = (int)g->ctx.size();
    const size_t chunk = (size_t)16 << 20;   // elements: n x 64 MiB of scratch per rank
    const size_t ce = count < chunk ? count : chunk;
    std::vector<void *> tmp(n, nullptr);
    int rc = CID_OK;
    for (int r = 0; r < n && rc == CID_OK; ++r) { HIP_TRY(hipSetDevice(g->dev[r])); rc = cid::ctx_alloc(g->ctx[r], ce * 4 * (size_t)n, &tmp[r]); }
    for (size_t i0 = 0; i0 < count && rc == CID_OK; i0 += chunk) {
        const size_t ne = count - i0 < chunk ? count - i0 : chunk;
        int e = g->rccl.GroupStart();
        for (int r = 0; r < n && e == 0; ++r) {
            if (hipSetDevice(g->dev[r]) != hipSuccess) { e = -1; break; }
            e = g->rccl.AllGather(d_bufs[r] + i0, tmp[r], ne, kNcclUint32, g->comms[r], g->ctx[r]->stream);
        }
        const int e2 = g->rccl.GroupEnd();
        if (e || e2) { rc = fail(CID_ERR_HIP, "ncclAllGather: %s", e > 0 || e2 ? g->rccl.GetErrorString(e > 0 ? e : e2) : "hipSetDevice failed"); break; }
        for (int r = 0; r < n; ++r) {   // rank j's words sit at tmp + j * ne
            if (hipSetDevice(g->dev[r]) != hipSuccess) { rc = fail(CID_ERR_HIP, "hipSetDevice"); break; }
            hipLaunchKernelGGL((k_fold_u32<false, true>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, g->ctx[r]->stream, d_bufs[r] + i0,
                               (const uint32_t *)tmp[r], (uint32_t)n, (uint64_t)ne, (uint64_t)ne);
            if (hipGetLastError() != hipSuccess) { rc = fail(CID_ERR_HIP, "stripe reduction: fold kernel"); break; }
        }
    }
    for (int r = 0; r < n; ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (hipStreamSynchronize(g->ctx[r]->stream) != hipSuccess && rc == CID_OK) rc = fail(CID_ERR_HIP, "stripe reduction: stream");
        cid::ctx_free(g->ctx[r], tmp[r]);
    }
    return rc;
}

// d_bufs[r]: u32[count] on rank r, produced on rank r's ctx stream.  Afterwards rank 0 — with `everywhere` every rank — holds the
// element-wise SUM / AND over the ranks.  SUM with RCCL: one ncclAllReduce.  AND with RCCL: all-gather + local fold.  Otherwise
// (device ids repeat, or COLORID_STRIPE_REDUCE=peer) a reduce-scatter / all-gather of peer copies with no funnel: the array is cut
// into one slice per rank; rank r pulls slice r of every peer on ITS OWN stream once the peer's "ready" event has fired, folds the
// n - 1 copies into its own words with one kernel and records "reduced"; then whoever needs the result (every rank, or rank 0 alone)
// pulls the reduced slices from their owners.  Every GPU moves 2 (n-1)/n of the array instead of rank 0 moving (n-1) arrays, all
// links work at once, and the host waits once per rank at the end — never inside the loops.
int reduce_u32(cid_group *g, uint32_t *const *d_bufs, size_t count, bool sum, bool everywhere) {
    const int n = (int)g->ctx.size();
    if (count == 0) return CID_OK;
    const bool force_peer = g->ctx[0]->tune.stripe_reduce_peer;   // COLORID_STRIPE_REDUCE=peer
    if (g->use_rccl && !force_peer)   // (also with one rank: COLORID_REDUCE=rccl)
        return sum ? allreduce_sum(g, reinterpret_cast<void *const *>(d_bufs), count, 4) : and_by_allgather(g, d_bufs, count);
    if (n == 1) return CID_OK;
    int rc = group_events(g);
    if (rc) return rc;
    std::vector<size_t> lo(n), hi(n);
    std::vector<void *> tmp(n, nullptr);
    for (int r = 0; r < n; ++r) {
        shard_bounds(count, r, n, &lo[r], &hi[r]);
        HIP_TRY(hipSetDevice(g->dev[r]));
        HIP_TRY(hipEventRecord(g->ev_ready[r], g->ctx[r]->stream));
    }
    auto fail_sync = [&](int code) {   // leave nothing in flight that reads a buffer the caller is about to reuse
        for (int r = 0; r < n; ++r) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); cid::ctx_free(g->ctx[r], tmp[r]); }
        return code;
    };
    for (int r = 0; r < n; ++r) {   // reduce-scatter: rank r owns [lo_r, hi_r)
        const size_t ne = hi[r] - lo[r];
        if (ne == 0) continue;
        cid_ctx *c = g->ctx[r];
        if (hipSetDevice(c->device) != hipSuccess) return fail_sync(fail(CID_ERR_HIP, "hipSetDevice"));
        if ((rc = cid::ctx_alloc(c, ne * 4 * (size_t)(n - 1), &tmp[r]))) return fail_sync(rc);
        int slot = 0;
        for (int p = 0; p < n; ++p) {
            if (p == r) continue;
            hipError_t e = hipStreamWaitEvent(c->stream, g->ev_ready[p], 0);
            if (e == hipSuccess) e = copy_between((uint32_t *)tmp[r] + (size_t)slot * ne, c->device, d_bufs[p] + lo[r], g->dev[p], ne * 4, c->stream);
            if (e != hipSuccess) return fail_sync(fail(CID_ERR_HIP, "stripe reduction (rank %d <- rank %d): %s", r, p, hipGetErrorString(e)));
            ++slot;
        }
        if (sum) hipLaunchKernelGGL((k_fold_u32<true, false>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, d_bufs[r] + lo[r],
                                    (const uint32_t *)tmp[r], (uint32_t)(n - 1), (uint64_t)ne, (uint64_t)ne);
        else hipLaunchKernelGGL((k_fold_u32<false, false>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, d_bufs[r] + lo[r],
                                (const uint32_t *)tmp[r], (uint32_t)(n - 1), (uint64_t)ne, (uint64_t)ne);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(g->ev_reduced[r], c->stream);
        if (e != hipSuccess) return fail_sync(fail(CID_ERR_HIP, "stripe reduction (rank %d): %s", r, hipGetErrorString(e)));
    }
    for (int p = 0; p < (everywhere ? n : 1); ++p) {   // all-gather (or gather on rank 0): the reduced slices travel from their owners
        cid_ctx *c = g->ctx[p];
        if (hipSetDevice(c->device) != hipSuccess) return fail_sync(fail(CID_ERR_HIP, "hipSetDevice"));
        for (int r = 0; r < n; ++r) {
            if (r == p || hi[r] == lo[r]) continue;
            hipError_t e = hipStreamWaitEvent(c->stream, g->ev_reduced[r], 0);
            if (e == hipSuccess) e = copy_between(d_bufs[p] + lo[r], c->device, d_bufs[r] + lo[r], g->dev[r], (hi[r] - lo[r]) * 4, c->stream);
            if (e != hipSuccess) return fail_sync(fail(CID_ERR_HIP, "stripe reduction (rank %d <- rank %d): %s", p, r, hipGetErrorString(e)));
        }
    }
    rc = CID_OK;
    for (int r = 0; r < n; ++r) {   // one wait per rank: afterwards no stream reads another rank's buffer any more
        (void)hipSetDevice(g->dev[r]);
        const hipError_t e = hipStreamSynchronize(g->ctx[r]->stream);
        if (e != hipSuccess && rc == CID_OK) rc = fail(CID_ERR_HIP, "stripe reduction (rank %d): %s", r, hipGetErrorString(e));
        cid::ctx_free(g->ctx[r], tmp[r]);
    }
    return rc;
}

// where a rank finds the query k-mers: a host array (uploaded by every rank) or a device-resident set (peer copies)
struct Query {
    const uint8_t *h_kmers = nullptr;        // host, n x k bytes
    const uint32_t *h_freq = nullptr;        // host multiplicities (rank 0 only needs them) or NULL
    cid_ctx *kc = nullptr;                   // the device arrays' context
    const uint64_t *d_codes = nullptr;       // device, 2-bit codes
    const uint8_t *d_ascii = nullptr;        // device, n x k bytes (k > 32 sets)
    const uint32_t *d_counts = nullptr;      // device multiplicities
    size_t n = 0;
    uint32_t k = 0;
};

// the whole query on rank r's device: *d_k (ASCII) or *d_c (codes)
int query_on_rank(cid_group *g, int r, const Query &q, const uint8_t **d_k, const uint64_t **d_c) {
    cid_ctx *c = g->ctx[r];
    *d_k = nullptr; *d_c = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    if (q.h_kmers) {
        void *d;
        const int rc = cid::slot_reserve(c, S_KMERS, q.n * q.k, &d); if (rc) return rc;
        if (q.n) HIP_TRY(hipMemcpyAsync(d, q.h_kmers, q.n * q.k, hipMemcpyHostToDevice, c->stream));
        *d_k = (const uint8_t *)d;
        return CID_OK;
    }
    const size_t unit = q.d_ascii ? q.k : 8;
    const uint8_t *src = q.d_ascii ? q.d_ascii : reinterpret_cast<const uint8_t *>(q.d_codes);
    if (c->device != q.kc->device) {   // another GPU: the set travels over xGMI
        void *d;
        const int rc = cid::slot_reserve(c, S_KMERS, q.n * unit, &d); if (rc) return rc;
        if (q.n) HIP_TRY(hipMemcpyPeerAsync(d, c->device, src, q.kc->device, q.n * unit, c->stream));
        src = (const uint8_t *)d;
    }
    if (q.d_ascii) *d_k = src; else *d_c = reinterpret_cast<const uint64_t *>(src);
    return CID_OK;
}

int query_from_set(const cid_kmerset *ks, uint32_t index_k, Query &q) {
    uint64_t nk;
    if (cid::kmerset_view_ascii(ks, &q.kc, &q.d_ascii, &q.d_counts, &nk, &q.k) != CID_OK) {
        q.d_ascii = nullptr;
        const int rc = cid::kmerset_view(ks, &q.kc, &q.d_codes, &q.d_counts, &nk, &q.k);
        if (rc) return rc;
    }
    q.n = (size_t)nk;
    if (q.k != index_k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", q.k, index_k);
    HIP_TRY(hipSetDevice(q.kc->device));
    HIP_TRY(hipStreamSynchronize(q.kc->stream));
    return CID_OK;
}

int stripes_search_count(cid_group *g, cid_index *const *stripes, const Query &q, uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq,
                         uint32_t *unique_colour, uint64_t *mode_unique_freq = nullptr) {
    Stripes st;
    int rc = check_stripes(g, stripes, st);
    if (rc) return rc;
    if (!hits) return fail(CID_ERR_INVALID, "null argument");
    const bool want_unique = n_unique || sum_unique_freq || unique_colour || mode_unique_freq;
    const size_t K = q.n;
    std::vector<uint32_t *> d_fact(st.n, nullptr);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        const uint8_t *d_k; const uint64_t *d_c;
        int e = query_on_rank(g, r, q, &d_k, &d_c); if (e) return e;
        void *d_h, *d_f;
        e = cid::slot_reserve(c, S_OUT, (size_t)stripes[r]->n_colors * 8 + 3 * (size_t)st.total * 8, &d_h); if (e) return e;
        e = cid::slot_reserve(c, S_UC, (K ? K : 1) * 4, &d_f); if (e) return e;
        HIP_TRY(hipMemsetAsync(d_f, 0, (K ? K : 1) * 4, c->stream));
        e = cid_search_count_stripe_dev(c, stripes[r], d_k, d_c, K, st.base[r], (uint64_t *)d_h, (uint32_t *)d_f);
        if (e) return e;
        HIP_TRY(hipMemcpyAsync(hits + st.base[r], d_h, (size_t)stripes[r]->n_colors * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        d_fact[r] = (uint32_t *)d_f;
        return CID_OK;
    });
    if (rc) return rc;
    if (!want_unique) return CID_OK;
    if ((rc = reduce_u32(g, d_fact.data(), K, true, false))) return rc;
    // rank 0: the facts -> n_unique / sum of multiplicities per colour and the unique colour per k-mer
    cid_ctx *c0 = g->ctx[0];
    HIP_TRY(hipSetDevice(c0->device));
    uint64_t *d_nu = (uint64_t *)c0->slot[S_OUT] + stripes[0]->n_colors, *d_sf = d_nu + st.total;
    HIP_TRY(hipMemsetAsync(d_nu, 0, 2 * (size_t)st.total * 8, c0->stream));
    const uint32_t *d_freq = nullptr;
    void *d_fr = nullptr;
    if (q.h_freq || (q.d_counts && q.kc->device != c0->device)) {
        rc = cid::slot_reserve(c0, S_FREQ, (K ? K : 1) * 4, &d_fr); if (rc) return rc;
        if (K) {
            if (q.h_freq) HIP_TRY(hipMemcpyAsync(d_fr, q.h_freq, K * 4, hipMemcpyHostToDevice, c0->stream));
            else HIP_TRY(hipMemcpyPeerAsync(d_fr, c0->device, q.d_counts, q.kc->device, K * 4, c0->stream));
        }
        d_freq = (const uint32_t *)d_fr;
    } else if (q.d_counts) d_freq = q.d_counts;
    void *d_uc;
    rc = cid::ctx_alloc(c0, (K ? K : 1) * 4, &d_uc); if (rc) return rc;
    rc = cid_search_unique_finalize_dev(c0, d_fact[0], d_freq, K, st.total, d_nu, d_sf, (uint32_t *)d_uc);
    if (rc == CID_OK && mode_unique_freq) {   // the report's mode per colour, on rank 0's device: nothing per k-mer goes to the host
        void *d_modes;
        rc = cid::ctx_alloc(c0, (size_t)st.total * 8, &d_modes);
        if (rc == CID_OK) {
            rc = cid::unique_freq_modes(c0, (const uint32_t *)d_uc, d_freq, K, st.total, (uint64_t *)d_modes);
            if (rc == CID_OK) {
                hipError_t e = hipMemcpyAsync(mode_unique_freq, d_modes, (size_t)st.total * 8, hipMemcpyDeviceToHost, c0->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
                if (e != hipSuccess) rc = fail(CID_ERR_HIP, "striped report: %s", hipGetErrorString(e));
            }
            cid::ctx_free(c0, d_modes);
        }
    }
    if (rc == CID_OK) {
        hipError_t e = hipSuccess;
        if (n_unique) e = hipMemcpyAsync(n_unique, d_nu, (size_t)st.total * 8, hipMemcpyDeviceToHost, c0->stream);
        if (e == hipSuccess && sum_unique_freq) e = hipMemcpyAsync(sum_unique_freq, d_sf, (size_t)st.total * 8, hipMemcpyDeviceToHost, c0->stream);
        if (e == hipSuccess && unique_colour && K) e = hipMemcpyAsync(unique_colour, d_uc, K * 4, hipMemcpyDeviceToHost, c0->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
        if (e != hipSuccess) rc = fail(CID_ERR_HIP, "striped search results: %s", hipGetErrorString(e));
    }
    cid::ctx_free(c0, d_uc);
    return rc;
}

int stripes_search_perfect(cid_group *g, cid_index *const *stripes, const Query &q, uint32_t *and_words_le, int *any_row_missing) {
    Stripes st;
    int rc = check_stripes(g, stripes, st);
    if (rc) return rc;
    if (!and_words_le || !any_row_missing) return fail(CID_ERR_INVALID, "null argument");
    if (q.n == 0) return fail(CID_ERR_INVALID, "perfect search needs at least one k-mer (src/perfect_search.rs:22-23)");
    const size_t K = q.n;
    const uint32_t w32_total = (st.total + 31u) / 32u;
    std::vector<uint32_t *> d_zero(st.n, nullptr);
    std::vector<std::vector<uint64_t>> words(st.n);
    rc = for_each_rank(g, [&](int r) -> int {
        cid_ctx *c = g->ctx[r];
        const uint8_t *d_k; const uint64_t *d_c;
        int e = query_on_rank(g, r, q, &d_k, &d_c); if (e) return e;
        void *d_w, *d_z;
        e = cid::slot_reserve(c, S_OUT, (size_t)stripes[r]->rs * 8, &d_w); if (e) return e;
        e = cid::slot_reserve(c, S_UC, K * 4, &d_z); if (e) return e;
        HIP_TRY(hipMemsetAsync(d_z, 0xFF, K * 4, c->stream));
        e = cid_search_perfect_stripe_dev(c, stripes[r], d_k, d_c, K, (uint64_t *)d_w, (uint32_t *)d_z);
        if (e) return e;
        words[r].resize(stripes[r]->rs);
        HIP_TRY(hipMemcpyAsync(words[r].data(), d_w, (size_t)stripes[r]->rs * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        d_zero[r] = (uint32_t *)d_z;
        return CID_OK;
    });
    if (rc) return rc;
    if ((rc = reduce_u32(g, d_zero.data(), K, false, false))) return rc;
    cid_ctx *c0 = g->ctx[0];
    HIP_TRY(hipSetDevice(c0->device));
    void *d_flag;
    rc = cid::slot_reserve(c0, S_MISC, 16, &d_flag); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_flag, 0, 4, c0->stream));
    const uint32_t seeds = stripes[0]->n_hash >= 32 ? ~0u : ((1u << stripes[0]->n_hash) - 1u);
    hipLaunchKernelGGL(k_any_masked, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, c0->stream, (const uint32_t *)d_zero[0], (uint64_t)K, seeds,
                       (int *)d_flag);
    HIP_TRY(hipGetLastError());
    int miss = 0;
    HIP_TRY(hipMemcpyAsync(&miss, d_flag, 4, hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    for (uint32_t w = 0; w < w32_total; ++w) and_words_le[w] = 0;
    if (!miss)
        for (int r = 0; r < st.n; ++r) {   // a stripe starts on a whole 64-colour word: its u32 words land at base / 32
            const uint32_t *src = reinterpret_cast<const uint32_t *>(words[r].data());
            for (uint32_t w = 0; w < stripes[r]->w32; ++w) and_words_le[st.base[r] / 32u + w] = src[w];
        }
    *any_row_missing = miss ? 1 : 0;
    return CID_OK;
}

}  // namespace

extern "C" {

// stripes of whole 64-colour words, sizes differing by at most one word; the last stripe ends at n_colors_total
int cid_group_stripes_create(cid_group *g, uint64_t bloom_size, uint32_t n_hash, uint32_t k, uint32_t n_colors_total, int hash_variant,
                             cid_index **stripes) {
    if (!g || !stripes) return fail(CID_ERR_INVALID, "null argument");
    const int n = (int)g->ctx.size();
    for (int r = 0; r < n; ++r) stripes[r] = nullptr;
    const size_t w64 = ((size_t)n_colors_total + 63) / 64;
    if (w64 < (size_t)n) return fail(CID_ERR_INVALID, "%u colours are %zu 64-colour words: fewer than the %d ranks (use fewer GPUs)", n_colors_total, w64, n);
    for (int r = 0; r < n; ++r) {
        size_t lo, hi;
        shard_bounds(w64, r, n, &lo, &hi);
        const uint32_t c0 = (uint32_t)(lo * 64), c1 = hi * 64 > n_colors_total ? n_colors_total : (uint32_t)(hi * 64);
        const int rc = cid_index_create(g->ctx[r], bloom_size, n_hash, k, c1 - c0, hash_variant, &stripes[r]);
        if (rc) {
            const std::string msg = cid_last_error();
            for (int i = 0; i < r; ++i) { cid_index_destroy(stripes[i]); stripes[i] = nullptr; }
            return fail(rc, "stripe %d: %s", r, msg.c_str());
        }
    }
    return CID_OK;
}

int cid_group_stripes_base(const cid_group *g, cid_index *const *stripes, uint32_t *colour_base) {
    if (!g || !stripes || !colour_base) return fail(CID_ERR_INVALID, "null argument");
    colour_base[0] = 0;
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        if (!stripes[r]) return fail(CID_ERR_INVALID, "null stripe %zu", r);
        colour_base[r + 1] = colour_base[r] + stripes[r]->n_colors;
    }
    return CID_OK;
}

// .bxi records of the WHOLE file (all colours): every rank uploads them and keeps its own words
int cid_group_stripes_put_records(cid_group *g, cid_index *const *stripes, const uint8_t *records, size_t n_records) {
    if (!g || !stripes || (n_records && !records)) return fail(CID_ERR_INVALID, "null argument");
    std::vector<uint32_t> base(g->ctx.size() + 1);
    int rc = cid_group_stripes_base(g, stripes, base.data());
    if (rc) return rc;
    const uint32_t total = base.back();
    return for_each_rank(g, [&](int r) { return cid::index_put_records_slice(stripes[r], records, n_records, total, base[r]); });
}

// sparse rows of the whole index (n_rows x W32_total words): sliced on the host, one cid_index_put_rows per rank
int cid_group_stripes_put_rows(cid_group *g, cid_index *const *stripes, const uint64_t *row_ids, const uint32_t *words_le, size_t n_rows) {
    if (!g || !stripes || (n_rows && (!row_ids || !words_le))) return fail(CID_ERR_INVALID, "null argument");
    std::vector<uint32_t> base(g->ctx.size() + 1);
    int rc = cid_group_stripes_base(g, stripes, base.data());
    if (rc) return rc;
    const size_t w32_total = ((size_t)base.back() + 31) / 32;
    return for_each_rank(g, [&](int r) -> int {
        const size_t w32 = stripes[r]->w32, off = base[r] / 32u;
        std::vector<uint32_t> slice(n_rows * w32);
        for (size_t i = 0; i < n_rows; ++i) memcpy(&slice[i * w32], &words_le[i * w32_total + off], w32 * 4);
        return cid_index_put_rows(stripes[r], row_ids, slice.data(), n_rows);
    });
}

int cid_group_stripes_search_count(cid_group *g, cid_index *const *stripes, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                                   uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    if (!g || !stripes || !stripes[0] || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    Query q;
    q.h_kmers = kmers ? kmers : reinterpret_cast<const uint8_t *>("");
    q.h_freq = freq; q.n = n_kmers; q.k = stripes[0]->k;
    return stripes_search_count(g, stripes, q, hits, n_unique, sum_unique_freq, unique_colour);
}

int cid_group_stripes_search_count_set(cid_group *g, cid_index *const *stripes, const cid_kmerset *ks, uint64_t *hits, uint64_t *n_unique,
                                       uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    if (!g || !stripes || !stripes[0]) return fail(CID_ERR_INVALID, "null argument");
    Query q;
    const int rc = query_from_set(ks, stripes[0]->k, q);
    if (rc) return rc;
    return stripes_search_count(g, stripes, q, hits, n_unique, sum_unique_freq, unique_colour);
}

int cid_group_stripes_search_count_set_report(cid_group *g, cid_index *const *stripes, const cid_kmerset *ks, uint64_t *hits, uint64_t *n_unique,
                                              uint64_t *sum_unique_freq, uint64_t *mode_unique_freq) {
    if (!g || !stripes || !stripes[0] || !mode_unique_freq) return fail(CID_ERR_INVALID, "null argument");
    Query q;
    const int rc = query_from_set(ks, stripes[0]->k, q);
    if (rc) return rc;
    return stripes_search_count(g, stripes, q, hits, n_unique, sum_unique_freq, nullptr, mode_unique_freq);
}

int cid_group_stripes_search_perfect(cid_group *g, cid_index *const *stripes, const uint8_t *kmers, size_t n_kmers, uint32_t *and_words_le,
                                     int *any_row_missing) {
    if (!g || !stripes || !stripes[0] || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    Query q;
    q.h_kmers = kmers; q.n = n_kmers; q.k = stripes[0]->k;
    return stripes_search_perfect(g, stripes, q, and_words_le, any_row_missing);
}

int cid_group_stripes_search_perfect_set(cid_group *g, cid_index *const *stripes, const cid_kmerset *ks, uint32_t *and_words_le, int *any_row_missing) {
    if (!g || !stripes || !stripes[0]) return fail(CID_ERR_INVALID, "null argument");
    Query q;
    const int rc = query_from_set(ks, stripes[0]->k, q);
    if (rc) return rc;
    return stripes_search_perfect(g, stripes, q, and_words_le, any_row_missing);
}

// a6-a10 over colour stripes: every rank takes the whole batch; zero pass, masks ANDed over the ranks, count pass into a report of
// the rank's own columns (+ the no-hits column on rank 0), compacted per rank; cid_group_readid_sparse_fetch splices the lists
int cid_group_stripes_readid_count_sparse(cid_group *g, cid_index *const *stripes, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                                          const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t *n_kmers,
                                          uint8_t *status, uint64_t *n_entries) {
    Stripes st;
    int rc = check_stripes(g, stripes, st);
    if (rc) return rc;
    if (!n_entries || !seq_off || !read_seq0 || (n_reads && (!n_kmers || !status))) return fail(CID_ERR_INVALID, "null argument");
    *n_entries = 0;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    g->sp_striped = true;
    g->sp_base = st.base;
    for (int r = 0; r < st.n; ++r) { g->sp_rows[r] = n_reads; g->sp_entries[r] = 0; g->ctx[r]->sp_rows = 0; g->ctx[r]->sp_entries = 0; }
    if (n_reads == 0) return CID_OK;
    for (size_t r = 0; r < n_reads; ++r)   // before seq_off is read through any entry
        if (read_seq0[r] > read_seq0[r + 1] || read_seq0[r + 1] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 not monotonic or past n_seqs at read %zu", r);
    const uint64_t total_bases = seq_off[n_seqs];
    if (total_bases && !bases) return fail(CID_ERR_INVALID, "null bases");
    uint64_t zn64 = 0;   // one mask word per k-mer window of the batch (validates the offsets too)
    if ((rc = cid_readid_stripe_mask_words(stripes[0]->k, stride_d, seq_off, read_seq0, n_reads, &zn64))) return rc;
    uint32_t widest = 0;
    for (int r = 0; r < st.n; ++r) widest = stripes[r]->n_colors > widest ? stripes[r]->n_colors : widest;
    if ((double)n_reads * ((double)widest + 1.0) * 4.0 > 64.0 * (double)(1ull << 30) || (double)zn64 * 4.0 > 64.0 * (double)(1ull << 30))
        return fail(CID_ERR_UNSUPPORTED, "%zu reads need more than 64 GiB of report rows or k-mer masks per GPU: use smaller batches", n_reads);
    const size_t zn = (size_t)zn64;
    std::vector<uint32_t *> d_z(st.n, nullptr);
    struct Dev { const uint8_t *bases; uint32_t *nk; uint8_t *status; };
    std::vector<Dev> dv(st.n);
    rc = for_each_rank(g, [&](int r) -> int {   // upload + zero pass (reads of any length: routed per stripe, cid_readid_stripe_zero)
        cid_ctx *c = g->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        void *d_b, *d_nk, *d_zz;
        int e = cid::slot_reserve(c, S_BASES, total_bases + 16, &d_b); if (e) return e;
        e = cid::slot_reserve(c, S_NK, n_reads * 4 + n_reads + 16, &d_nk); if (e) return e;
        e = cid::slot_reserve(c, S_UC, zn * 4, &d_zz); if (e) return e;
        if (total_bases) {   // through the rank's pinned arena when the batch fits (cid::pin_reserve)
            const uint8_t *src = bases;
            if (uint8_t *pin = cid::pin_reserve(c, total_bases)) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                memcpy(pin, bases, total_bases);
                src = pin;
            }
            HIP_TRY(hipMemcpyAsync(d_b, src, total_bases, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemsetAsync(d_zz, 0xFF, zn * 4, c->stream));
        dv[r] = Dev{(const uint8_t *)d_b, (uint32_t *)d_nk, (uint8_t *)d_nk + n_reads * 4};
        d_z[r] = (uint32_t *)d_zz;
        return cid_readid_stripe_zero(c, stripes[r], dv[r].bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, d_z[r], dv[r].nk, dv[r].status);
    });
    if (rc) return rc;
    if ((rc = reduce_u32(g, d_z.data(), zn, false, true))) return rc;
    rc = for_each_rank(g, [&](int r) -> int {   // count pass into the rank's own columns, then its (colour, count) lists
        cid_ctx *c = g->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        const uint32_t Cr = stripes[r]->n_colors;
        void *d_rep;
        int e = cid::slot_reserve(c, S_REPORT, n_reads * ((size_t)Cr + 1) * 4, &d_rep); if (e) return e;
        HIP_TRY(hipMemsetAsync(d_rep, 0, n_reads * ((size_t)Cr + 1) * 4, c->stream));
        e = cid_readid_stripe_count(c, stripes[r], dv[r].bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, 0, Cr,
                                    r == 0 ? 1 : 0, d_z[r], (uint32_t *)d_rep, dv[r].nk, dv[r].status);
        if (e) return e;
        cid::ctx_free(c, c->sp_start); c->sp_start = nullptr;
        cid::ctx_free(c, c->sp_col); c->sp_col = nullptr;
        cid::ctx_free(c, c->sp_cnt); c->sp_cnt = nullptr;
        e = cid::compact_report(c, (const uint32_t *)d_rep, Cr + 1, n_reads, &c->sp_start, &c->sp_col, &c->sp_cnt, &c->sp_entries);
        if (e) return e;
        c->sp_rows = n_reads;
        g->sp_entries[r] = c->sp_entries;
        if (r == 0) {
            HIP_TRY(hipMemcpyAsync(n_kmers, dv[r].nk, n_reads * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(status, dv[r].status, n_reads, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        return CID_OK;
    });
    if (rc) return rc;
    for (int r = 0; r < st.n; ++r) *n_entries += g->sp_entries[r];
    return CID_OK;
}

}  // extern "C"

// the striped half of cid_group_readid_sparse_fetch: per read, rank 0's colours, rank 1's (+ base) ..., the no-hits entry last
int cidg::stripes_sparse_fetch(cid_group *g, uint64_t *row_start, uint32_t *colours, uint32_t *counts) {
    const int n = (int)g->ctx.size();
    const uint64_t n_reads = g->sp_rows[0];
    row_start[0] = 0;
    if (n_reads == 0) return CID_OK;
    const uint32_t total = g->sp_base[n];
    std::vector<std::vector<uint64_t>> rs(n);
    std::vector<std::vector<uint32_t>> col(n), cnt(n);
    for (int r = 0; r < n; ++r) {
        rs[r].resize(n_reads + 1);
        col[r].resize(g->sp_entries[r]); cnt[r].resize(g->sp_entries[r]);
        const int rc = cid_readid_sparse_fetch(g->ctx[r], rs[r].data(), col[r].data(), cnt[r].data());
        if (rc) return rc;
    }
    const uint32_t c0_nohits = g->sp_base[1];   // rank 0's report has its no-hits column after its own colours
    uint64_t out = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        bool nohits = false;
        uint32_t nohits_count = 0;
        for (int r = 0; r < n; ++r)
            for (uint64_t e = rs[r][i]; e < rs[r][i + 1]; ++e) {
                if (r == 0 && col[0][e] == c0_nohits) { nohits = true; nohits_count = cnt[0][e]; continue; }
                if (colours) { colours[out] = g->sp_base[r] + col[r][e]; counts[out] = cnt[r][e]; }
                ++out;
            }
        if (nohits) {
            if (colours) { colours[out] = total; counts[out] = nohits_count; }
            ++out;
        }
        row_start[i + 1] = out;
    }
    return CID_OK;
}
