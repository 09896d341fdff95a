// C ABI of libcolorid_hip.so (see include/colorid_hip.h), part 1: errors, contexts and their scratch memory, per-context tunables,
// warm-up and timers.  Host-side plumbing only — there is no CPU compute path in this library.
#include <mutex>
#include <chrono>

#include "cid_api_common.hpp"

using cid::fail;
using namespace cid::slots;

namespace {
thread_local char g_err[512] = "";
}

namespace cid {
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ctx_alloc(cid_ctx *c, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    int best = -1;   // the smallest idle block that fits without wasting more than half of itself
    for (size_t i = 0; i < c->blocks.size(); ++i) {
        const cid_ctx::Block &b = c->blocks[i];
        if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + (1u << 20) && (best < 0 || b.bytes < c->blocks[best].bytes)) best = (int)i;
    }
    if (best >= 0) {
        c->blocks[best].used = true;
        c->idle_bytes -= c->blocks[best].bytes;
        *out = c->blocks[best].p;
        return CID_OK;
    }
    const size_t want = bytes + bytes / 8;   // batches vary a little in size: leave room for the next one
    void *p = nullptr;
    const auto t_alloc = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, want);
    if (c->tune.alloc_trace)
        fprintf(stderr, "cid alloc: block %zu B in %.2f ms (%zu blocks held)\n", want,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count(), c->blocks.size());
    size_t got = want;
    if (e != hipSuccess) {   // give the idle blocks back and ask for exactly what is needed
        (void)hipGetLastError();
        for (size_t i = 0; i < c->blocks.size();) {
            if (!c->blocks[i].used) { (void)hipFree(c->blocks[i].p); c->blocks.erase(c->blocks.begin() + (long)i); }
            else ++i;
        }
        c->idle_bytes = 0;
        got = bytes;
        e = hipMalloc(&p, got);
        if (e != hipSuccess) return fail(CID_ERR_NOMEM, "hipMalloc(%zu): %s", got, hipGetErrorString(e));
    }
    c->blocks.push_back(cid_ctx::Block{p, got, true});
    *out = p;
    return CID_OK;
}
void ctx_free(cid_ctx *c, void *p) {
    if (!p) return;
    constexpr size_t kMaxIdle = 64ull << 30;   // of 288 GB; beyond that blocks really go back
    for (size_t i = 0; i < c->blocks.size(); ++i) {
        if (c->blocks[i].p != p) continue;
        if (c->idle_bytes + c->blocks[i].bytes > kMaxIdle) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipFree(p);
            c->blocks.erase(c->blocks.begin() + (long)i);
        } else {
            c->blocks[i].used = false;
            c->idle_bytes += c->blocks[i].bytes;
        }
        return;
    }
    (void)hipFree(p);   // not one of ours
}
int ctx_device(const cid_ctx *c) { return c->device; }
hipStream_t ctx_stream(const cid_ctx *c) { return c->stream; }
int ctx_order_bits(const cid_ctx *c) { return c->tune.order_bits; }
int ctx_n_cu(const cid_ctx *c) { return c->n_cu; }
hipStream_t ctx_own_stream(const cid_ctx *c) { return c->own_stream; }
hipStream_t ctx_copy_stream(const cid_ctx *c) { return c->copy_stream; }

hipError_t ctx_side_streams(cid_ctx *c, hipStream_t out[4]) {
    static std::mutex mu;   // (the warm-up thread and the thread that creates the reader may both come first)
    std::lock_guard<std::mutex> lk(mu);
    if (!c->side_streams[3]) {
        hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess) return e;
        // the inflate launches run beside the classifier's kernels, which fill every CU: on a queue of the highest priority their few
        // long-lived waves are placed as soon as a classifier block retires instead of waiting their turn (CID_INFLATE_PRIORITY=0: equal)
        int prio_low = 0, prio_high = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_high = 0;
        const bool want_prio = c->tune.inflate_priority;
        hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4 && e == hipSuccess; ++i)
            e = i < 2 ? hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, want_prio ? prio_high : 0) : hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (hipStream_t x : s) if (x) (void)hipStreamDestroy(x);
            return e;
        }
        for (int i = 0; i < 4; ++i) c->side_streams[i] = s[i];
    }
    for (int i = 0; i < 4; ++i) out[i] = c->side_streams[i];
    return hipSuccess;
}
hipEvent_t ctx_event(const cid_ctx *c, int i) { return i == 0 ? c->ev_copied[0] : c->ev_done[0]; }

int slot_reserve(cid_ctx *c, int s, size_t bytes, void **out) {
    // a started cid_bgzf_inflate batch owns these five slots (and the pinned arena) until its _finish (S_ROWIDS: the wave kernel's match
    // tokens and retry list)
    if (c->inflate.open && (s == S_KMERS || s == S_MISC || s == S_BASES || s == S_FREQ || s == S_ROWIDS))
        return fail(CID_ERR_STATE, "a cid_bgzf_inflate_start on this ctx is waiting for its _finish: this call would overwrite its buffers");
    if (bytes == 0) bytes = 16;
    if (c->slot_bytes[s] < bytes) {
        if (c->slot[s]) HIP_TRY(hipFree(c->slot[s]));
        c->slot[s] = nullptr;
        c->slot_bytes[s] = 0;
        const size_t want = bytes + bytes / 4;
        const auto t_alloc = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&c->slot[s], want);
        if (c->tune.alloc_trace)
            fprintf(stderr, "cid alloc: slot %d %zu B in %.2f ms\n", s, want, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count());
        if (e != hipSuccess) return fail(CID_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        c->slot_bytes[s] = want;
    }
    *out = c->slot[s];
    return CID_OK;
}
uint8_t *pin_reserve(cid_ctx *c, size_t bytes, size_t cap) {
    if (!c->tune.pin_staging || bytes > cap) return nullptr;
    if (c->inflate.open) return nullptr;   // the arena holds a started inflate batch's text and status: callers copy without it
    if (bytes <= c->pin_bytes) return c->pin;
    if (c->pin) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_bytes = 0; }
    size_t want = bytes + bytes / 2;
    if (want < (16u << 20)) want = 16u << 20;
    if (want > cap) want = cap;
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    c->pin = (uint8_t *)p;
    c->pin_bytes = want;
    return c->pin;
}
}  // namespace cid

extern "C" {

const char *cid_last_error(void) { return g_err; }
int cid_abi_version(void) { return 4; }   // 4: + cid_readid_count_resident; bgzf launches take scratch (internal)

int cid_device_count(int *n) {
    if (!n) return fail(CID_ERR_INVALID, "null out");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(CID_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return CID_OK;
}

// The library's one reader of environment variables (cid_switches.def lists them): every switch lands in the new context's cid_tunables.
static void read_switches(cid_tunables &t) {
#define CID_SWITCH_B(name, env) if (const char *e = getenv(env)) t.name = atoi(e) != 0;
#define CID_SWITCH_I(name, env) if (const char *e = getenv(env)) t.name = atoi(e);
#define CID_SWITCH_L(name, env) if (const char *e = getenv(env)) t.name = atol(e);
#define CID_SWITCH_U(name, env) if (const char *e = getenv(env)) t.name = strtoull(e, nullptr, 10);
#define CID_SWITCH_S(name, env)
#define CID_SWITCH(name, env, kind, dflt, doc) CID_SWITCH_##kind(name, env)
#include "cid_switches.def"
#undef CID_SWITCH
#undef CID_SWITCH_B
#undef CID_SWITCH_I
#undef CID_SWITCH_L
#undef CID_SWITCH_U
#undef CID_SWITCH_S
    if (t.search_unroll != 1) t.search_unroll = 2;
    if (t.readid_blocks_per_cu < 1 || t.readid_blocks_per_cu > 4096) t.readid_blocks_per_cu = 64;
    if (t.order_bits < 0 || t.order_bits > 32) t.order_bits = 0;
#ifndef CID_TUNE_BUILD
    t.search_persist = t.search_mixed = false;   // (their kernels are only in libcolorid_hip_tune.so)
#endif
    if (const char *e = getenv("COLORID_REDUCE")) t.reduce_mode = !strcmp(e, "rccl") ? 1 : !strcmp(e, "host") ? 0 : -1;
    if (const char *e = getenv("COLORID_STRIPE_REDUCE")) t.stripe_reduce_peer = !strcmp(e, "peer");
    if (const char *sy = getenv("COLORID_SYNC")) {   // how host threads wait for the device (before the device's first use): spin | yield | block
        const unsigned f = !strcmp(sy, "spin") ? hipDeviceScheduleSpin : !strcmp(sy, "yield") ? hipDeviceScheduleYield : !strcmp(sy, "block")
            ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        (void)hipSetDeviceFlags(f);   // (refused once the device is active: the first context decides)
    }
}

int cid_ctx_create(int device_id, cid_ctx **out) {
    if (!out) return fail(CID_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(CID_ERR_HIP, "no HIP device (this library has no CPU path)");
    if (device_id < 0 || device_id >= n) return fail(CID_ERR_INVALID, "device %d of %d", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    cid_ctx *c = new (std::nothrow) cid_ctx();
    if (!c) return fail(CID_ERR_NOMEM, "ctx");
    c->device = device_id;
    read_switches(c->tune);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
        ok = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail(CID_ERR_HIP, "stream/event creation failed");
    }
    c->stream = c->own_stream;
    *out = c;
    return CID_OK;
}

// Page-locked host memory for buffers that travel to the device again and again (a reader's text or member buffers): copies from it
// run at the bus rate and truly asynchronously; the runtime pins and unpins pageable memory around every copy instead.
int cid_pinned_alloc(size_t bytes, void **out) {
    if (!out) return fail(CID_ERR_INVALID, "null out");
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; (void)hipGetLastError(); return fail(CID_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return CID_OK;
}
void cid_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int cid_ctx_set_stream(cid_ctx *c, void *hip_stream) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return CID_OK;
}

int cid_ctx_synchronize(cid_ctx *c) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

void cid_ctx_destroy(cid_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < S_COUNT; ++s)
        if (c->slot[s]) (void)hipFree(c->slot[s]);
    for (const cid_ctx::Block &b : c->blocks) (void)hipFree(b.p);   // includes the sparse read_id result
    for (int i = 0; i < 2; ++i) {
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipStream_t s : c->side_streams) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

// ------------------------------------------------------------------------------------------------ tunables + timing

// Measurement / test switches of ONE context (see cid_tunables): no process-wide state, two contexts may differ.
int cid_ctx_tune(cid_ctx *c, const char *name, long value) {
    if (!c || !name) return fail(CID_ERR_INVALID, "null argument");
    if (!strcmp(name, "search_unroll")) {
        if (value != 1 && value != 2) return fail(CID_ERR_INVALID, "search_unroll is 1 or 2");
        c->tune.search_unroll = (int)value;
        return CID_OK;
    }
    if (!strcmp(name, "readid_packed_table")) { c->tune.readid_packed_table = value != 0; return CID_OK; }
    if (!strcmp(name, "fastq_refuse_at_step")) { c->tune.fastq_refuse_at_step = value; return CID_OK; }
    if (!strcmp(name, "readid_long_from")) { c->tune.readid_long_from = value; return CID_OK; }
    if (!strcmp(name, "readid_long_deal")) { c->tune.readid_long_deal = value != 0; return CID_OK; }
    if (!strcmp(name, "readid_long_lds")) { c->tune.readid_long_lds = value != 0; return CID_OK; }
    if (!strcmp(name, "readid_long_fuse")) { c->tune.readid_long_fuse = value != 0; return CID_OK; }
    if (!strcmp(name, "readid_blocks_per_cu")) {
        if (value < 1 || value > 4096) return fail(CID_ERR_INVALID, "readid_blocks_per_cu must be 1..4096");
        c->tune.readid_blocks_per_cu = (int)value;
        return CID_OK;
    }
    if (!strcmp(name, "order_bits")) {
        if (value < 0 || value > 32) return fail(CID_ERR_INVALID, "order_bits 0..32");
        c->tune.order_bits = (int)value;
        return CID_OK;
    }
    if (!strcmp(name, "search_persist") || !strcmp(name, "search_mixed")) {
#ifdef CID_TUNE_BUILD
        (name[7] == 'p' ? c->tune.search_persist : c->tune.search_mixed) = value != 0;
        return CID_OK;
#else
        return fail(CID_ERR_UNSUPPORTED, "'%s' is a rejected scheduling of k_search_count kept for reproducibility: its kernels are only in a "
                    "`make TUNE=1` build (libcolorid_hip_tune.so)", name);
#endif
    }
    // every other switch of cid_switches.def, by its name
#define CID_SWITCH_B(nm) if (!strcmp(name, #nm)) { c->tune.nm = value != 0; return CID_OK; }
#define CID_SWITCH_I(nm) if (!strcmp(name, #nm)) { c->tune.nm = (int)value; return CID_OK; }
#define CID_SWITCH_L(nm) if (!strcmp(name, #nm)) { c->tune.nm = value; return CID_OK; }
#define CID_SWITCH_U(nm) if (!strcmp(name, #nm)) { if (value < 0) return fail(CID_ERR_INVALID, "'%s' is not negative", name); c->tune.nm = (unsigned long long)value; return CID_OK; }
#define CID_SWITCH_S(nm)
#define CID_SWITCH(nm, env, kind, dflt, doc) CID_SWITCH_##kind(nm)
#include "cid_switches.def"
#undef CID_SWITCH
#undef CID_SWITCH_B
#undef CID_SWITCH_I
#undef CID_SWITCH_L
#undef CID_SWITCH_U
#undef CID_SWITCH_S
    if (!strcmp(name, "reduce_mode")) { c->tune.reduce_mode = value < 0 ? -1 : value ? 1 : 0; return CID_OK; }
    if (!strcmp(name, "stripe_reduce_peer")) { c->tune.stripe_reduce_peer = value != 0; return CID_OK; }
    return fail(CID_ERR_INVALID, "unknown tunable '%s'", name);
}

// The runtime loads a translation unit's device code on the first launch of one of its kernels — ~60 ms for the read_id kernels,
// paid inside the first cid_readid_count* call.  This call pays it ahead of time and may run on another host thread than the one
// using the ctx (it touches no stream, no ctx state): the CLI runs it beside the index load.
// What the first call of a process pays beyond its code objects does not grow with the call: the runtime's staging buffers for copies out
// of pageable memory, its first events and its copy queues — 9-10 ms inside the first cid_kmerset_add_seqs of a process, whatever the query's
// size (profiles/r06_first_use.txt: a query of 2 000 reads first, and the million-read query after it runs at the steady state's speed).  So the
// warm-up runs a query of a few reads itself, on a context of its own (its own stream and scratch: nothing of `c` is touched, the caller may
// be loading an index into it on another thread) against an index of 1 024 rows, and throws both away.
static void warm_dry_run(int device, unsigned what) {
    cid_ctx *t = nullptr;
    if (cid_ctx_create(device, &t) != CID_OK) return;
    cid_index *ix = nullptr;
    const uint32_t k = 21;
    std::vector<uint8_t> bases(16 * 150 + 2400);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (uint8_t &b : bases) { x = x * 6364136223846793005ull + 1442695040888963407ull; b = (uint8_t)"ACGT"[x >> 62]; }
    std::vector<uint64_t> so(18), r0(18);
    for (size_t i = 0; i <= 16; ++i) so[i] = i * 150;
    so[17] = bases.size();                     // ... and one read of 2 400 bases: the long-read kernels
    for (size_t i = 0; i < 18; ++i) r0[i] = i;
    if (cid_index_create(t, 1024, 2, k, 4, 0, &ix) == CID_OK && cid_index_finalize(ix) == CID_OK) {
        if (what & CID_WARM_SEARCH) {
            cid_kmerset *ks = nullptr;
            if (cid_kmerset_create(t, k, &ks) == CID_OK) {
                uint64_t nd = 0, hits[4], nu[4], su[4], mode[4];
                if (cid_kmerset_add_seqs(ks, bases.data(), so.data(), 17, 0) == CID_OK && cid_kmerset_finalize(ks, &nd) == CID_OK)
                    (void)cid_search_count_set_report(t, ix, ks, hits, nu, su, mode);
                cid_kmerset_destroy(ks);
            }
        }
        if (what & CID_WARM_READID) {
            uint32_t nk[17];
            uint8_t st[17];
            uint64_t ne = 0;
            (void)cid_readid_count_sparse(t, ix, bases.data(), so.data(), 17, r0.data(), 17, 1, 3, nk, st, &ne);
        }
    }
    if (ix) cid_index_destroy(ix);
    cid_ctx_destroy(t);
}

int cid_warmup(cid_ctx *c, unsigned what) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (what & CID_WARM_READID) { HIP_TRY(cid::warm_readid()); HIP_TRY(cid::warm_readlong()); }
    if (what & CID_WARM_COLD) HIP_TRY(cid::warm_cold());
    if (what & CID_WARM_SEARCH) HIP_TRY(cid::warm_search());
    if (what & (CID_WARM_READID | CID_WARM_SEARCH)) HIP_TRY(cid::warm_reports());   // sparse report rows / modes: a small code object
    if (what & CID_WARM_SEARCH) HIP_TRY(cid::warm_kmerset());                         // the k-mer set's sorts: 18 MB, 0.2 s to load
    if (what & CID_WARM_INFLATE) HIP_TRY(cid::warm_inflate());
    if (what & CID_WARM_FASTQ) { hipStream_t s[4]; HIP_TRY(cid::ctx_side_streams(c, s)); }
    if (what & CID_WARM_FASTQ) HIP_TRY(cid::warm_fastq());
    if ((what & CID_WARM_PIPES) && (what & (CID_WARM_SEARCH | CID_WARM_READID)) && c->tune.warm_dry_run) {
        warm_dry_run(c->device, what);
        // ... and this context's own queues: the runtime makes a stream's hardware queue with the first command it is given (7-9 ms for the
        // copy stream, inside the first cid_kmerset_add_seqs).  A fill kernel and a 64-byte copy on each; HIP's streams may be used from any thread, and nothing
        // of the ctx's state is touched.
        {   // ... and the bus: the first large copy after the device has been idle runs at a third of the link's rate (150 MB in 10-11 ms instead of
            // 3.9: the link and the copy engines wake up; profiles/r06_first_use.txt) — 32 MB from page-locked memory, here, instead of the query's first slice
            void *hp = nullptr, *dp = nullptr;
            const size_t nb = 32u << 20;
            if (hipHostMalloc(&hp, nb, hipHostMallocDefault) == hipSuccess && hipMalloc(&dp, nb) == hipSuccess) {
                memset(hp, 0, nb);
                for (int rep = 0; rep < 3; ++rep) (void)hipMemcpy(dp, hp, nb, hipMemcpyHostToDevice);
            }
            if (dp) (void)hipFree(dp);
            if (hp) (void)hipHostFree(hp);
        }
        void *d = nullptr;
        uint8_t h[64] = {0};
        if (hipMalloc(&d, 256) == hipSuccess) {
            for (hipStream_t st : {c->copy_stream, c->own_stream})
                if (st && hipMemsetAsync(d, 0, 256, st) == hipSuccess && hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, st) == hipSuccess)   // (a dispatch and a copy)
                    (void)hipStreamSynchronize(st);
            (void)hipFree(d);
        }
        (void)hipGetLastError();
    }                            // 3.9 MB: its scans and selects are rocPRIM's
    return CID_OK;
}

int cid_timer_start(cid_ctx *c) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return CID_OK;
}

int cid_timer_stop_ms(cid_ctx *c, float *elapsed_ms) {
    if (!c || !elapsed_ms) return fail(CID_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return CID_OK;
}

}  // extern "C"
