// C ABI of libcolorid_hip.so (see include/colorid_hip.h).  Host-side plumbing only: contexts, the
// device-resident index, scratch buffers, launches.  There is no CPU compute path in this library.
#include "../../include/colorid_hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "cid_host_math.hpp"
#include "cid_internal.hpp"
#include "cid_objects.hpp"

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
}  // namespace cid

namespace {

using cid::fail;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(CID_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

using namespace cid::slots;

// A/B switch while the persistent XCD-aware scheduling of k_search_count is being measured (cid_tune "search_persist")
bool kSearchPersist = getenv("CID_SEARCH_PERSIST") && atoi(getenv("CID_SEARCH_PERSIST")) != 0;
// 32-byte rows: the last row of every k-mer through the scalar cache (cid_tune "search_mixed")
bool kSearchMixed = getenv("CID_SEARCH_MIXED") ? atoi(getenv("CID_SEARCH_MIXED")) != 0 : false;
// rows of 64 bytes and more: sub-passes of k_search_count whose row loads are issued together (cid_tune "search_unroll"; 2: -1.5 %)
int kSearchUnroll = getenv("CID_SEARCH_UNROLL") ? atoi(getenv("CID_SEARCH_UNROLL")) : 2;
// bytes per chunk of the pipelined host-pointer calls (H2D of chunk i+1 beside the kernel of chunk i)
const size_t kUploadChunkBytes = getenv("CID_UPLOAD_CHUNK_BYTES") ? strtoull(getenv("CID_UPLOAD_CHUNK_BYTES"), nullptr, 10) : (256ull << 20);

}  // namespace

namespace cid {
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
    hipError_t e = hipMalloc(&p, want);
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
uint32_t index_k(const cid_index *ix) { return ix->k; }
uint32_t index_rs(const cid_index *ix) { return ix->rs; }
ModMagic index_mod(const cid_index *ix) { return ix->mod; }
uint32_t index_n_colors(const cid_index *ix) { return ix->n_colors; }
uint32_t index_n_hash(const cid_index *ix) { return ix->n_hash; }
uint32_t index_m_size(const cid_index *ix) { return ix->m_size; }
const uint64_t *index_matrix(const cid_index *ix) { return ix->mat; }
}  // namespace cid

namespace cid {
int slot_reserve(cid_ctx *c, int s, size_t bytes, void **out) {
    // a started cid_bgzf_inflate batch owns these four slots (and the pinned arena) until its _finish
    if (c->inflate.open && (s == S_KMERS || s == S_MISC || s == S_BASES || s == S_FREQ))
        return fail(CID_ERR_STATE, "a cid_bgzf_inflate_start on this ctx is waiting for its _finish: this call would overwrite its buffers");
    if (bytes == 0) bytes = 16;
    if (c->slot_bytes[s] < bytes) {
        if (c->slot[s]) HIP_TRY(hipFree(c->slot[s]));
        c->slot[s] = nullptr;
        c->slot_bytes[s] = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&c->slot[s], want);
        if (e != hipSuccess) return fail(CID_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        c->slot_bytes[s] = want;
    }
    *out = c->slot[s];
    return CID_OK;
}
int check_ready(const cid_ctx *c, const cid_index *ix) {
    if (!c || !ix) return fail(CID_ERR_INVALID, "null ctx/index");
    if (!ix->finalized) return fail(CID_ERR_STATE, "index not finalized");
    if (ix->ctx->device != c->device) return fail(CID_ERR_INVALID, "index lives on device %d, ctx on %d", ix->ctx->device, c->device);
    return CID_OK;
}
// `search` is not defined on minimizer indices ("An index with minimizers (.mxi) is used, but not available for this
// function", src/main.rs:569-573)
int check_not_mini(const cid_index *ix) {
    return ix->m_size ? fail(CID_ERR_UNSUPPORTED, "search on a minimizer (.mxi) index is not defined by the reference") : CID_OK;
}
}  // namespace cid

namespace {
using cid::check_not_mini;
using cid::check_ready;
using cid::slot_reserve;

// Work per block: enough blocks to balance 256 CUs dynamically, few enough that the per-block flush of the
// LDS counters (<= 3*C global atomics) stays negligible.
uint32_t pick_tiles_per_block(const cid_ctx *c, uint64_t n_kmers) {
    const uint64_t n_tiles = (n_kmers + cid::kWave - 1) / cid::kWave;
    uint64_t tpb = n_tiles / ((uint64_t)c->n_cu * 32);
    if (tpb < 4) tpb = 4;
    if (tpb > 256) tpb = 256;
    return (uint32_t)tpb;
}

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
    p.unroll = (uint32_t)kSearchUnroll;
    if (cid::search_smem_bytes(p) > 160u * 1024u)
        return fail(CID_ERR_UNSUPPORTED, "LDS need %zu B exceeds 160 KiB (n_colors=%u k=%u n_hash=%u)",
                    cid::search_smem_bytes(p), ix->n_colors, ix->k, ix->n_hash);
    (void)c;
    return CID_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

namespace cid {
static const bool kUsePin = getenv("CID_PIN_STAGING") ? atoi(getenv("CID_PIN_STAGING")) != 0 : true;
uint8_t *pin_reserve(cid_ctx *c, size_t bytes, size_t cap) {
    if (!kUsePin || bytes > cap) return nullptr;
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
int cid_abi_version(void) { return 2; }

int cid_device_count(int *n) {
    if (!n) return fail(CID_ERR_INVALID, "null out");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; return fail(CID_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return CID_OK;
}

int cid_ctx_create(int device_id, cid_ctx **out) {
    if (!out) return fail(CID_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(CID_ERR_HIP, "no HIP device (this library has no CPU path)");
    if (device_id < 0 || device_id >= n) return fail(CID_ERR_INVALID, "device %d of %d", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    if (const char *sy = getenv("COLORID_SYNC")) {   // how host threads wait for the device (before the device's first use): spin | yield | block
        const unsigned f = !strcmp(sy, "spin") ? hipDeviceScheduleSpin : !strcmp(sy, "yield") ? hipDeviceScheduleYield : !strcmp(sy, "block") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        (void)hipSetDeviceFlags(f);   // (refused once the device is active: the first context decides)
    }
    cid_ctx *c = new (std::nothrow) cid_ctx();
    if (!c) return fail(CID_ERR_NOMEM, "ctx");
    c->device = device_id;
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
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

// ------------------------------------------------------------------------------------------------ index

int cid_index_create(cid_ctx *c, uint64_t bloom_size, uint32_t num_hash, uint32_t k_size, uint32_t n_colors,
                     int hash_variant, cid_index **out) {
    if (!c || !out) return fail(CID_ERR_INVALID, "null ctx/out");
    *out = nullptr;
    if (hash_variant < 0 || hash_variant >= CID_HASH_VARIANTS) return fail(CID_ERR_UNSUPPORTED, "hash variant %d", hash_variant);
    if (bloom_size == 0 || num_hash == 0 || n_colors == 0 || k_size == 0) return fail(CID_ERR_INVALID, "zero parameter");
    if (k_size > cid::kMaxK) return fail(CID_ERR_UNSUPPORTED, "k_size %u > %u", k_size, cid::kMaxK);
    if (num_hash > 32) return fail(CID_ERR_UNSUPPORTED, "num_hash %u > 32", num_hash);
    if (bloom_size > (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "bloom_size %llu > 2^32", (unsigned long long)bloom_size);
    if (n_colors > (1u << 20)) return fail(CID_ERR_UNSUPPORTED, "n_colors %u > 2^20", n_colors);
    cid_index *ix = new (std::nothrow) cid_index();
    if (!ix) return fail(CID_ERR_NOMEM, "index");
    ix->ctx = c;
    ix->m = bloom_size; ix->n_hash = num_hash; ix->k = k_size; ix->n_colors = n_colors;
    ix->w32 = (n_colors + 31) / 32;
    ix->w64 = (n_colors + 63) / 64;
    ix->rs = cid::row_stride_words(n_colors);
    const cid::ModMagicHost mh = cid::make_mod_magic(bloom_size);
    ix->mod = cid::ModMagic{mh.m, mh.magic, mh.shift, mh.flags | ((uint32_t)hash_variant << 8),
                            hash_variant == CID_HASH_XXH3_V07 ? 0x165667B19E3779F9ULL : 0x165667919E3779F9ULL};
    hipError_t e = hipSetDevice(c->device);
    const size_t bytes = (size_t)bloom_size * ix->rs * 8;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ix->mat), bytes);
    if (e != hipSuccess) { delete ix; return fail(CID_ERR_NOMEM, "hipMalloc(%zu) for the index: %s", bytes, hipGetErrorString(e)); }
    e = hipMemsetAsync(ix->mat, 0, bytes, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(ix->mat); delete ix; return fail(CID_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
    *out = ix;
    return CID_OK;
}

int cid_index_set_minimizer(cid_index *ix, uint32_t m_size) {
    if (!ix) return fail(CID_ERR_INVALID, "null index");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (m_size == 0 || m_size > ix->k) return fail(CID_ERR_INVALID, "minimizer size %u must be in 1..k_size (%u)", m_size, ix->k);
    ix->m_size = m_size;
    return CID_OK;
}

int cid_index_set_hash_variant(cid_index *ix, int hash_variant) {
    if (!ix) return fail(CID_ERR_INVALID, "null index");
    if (hash_variant < 0 || hash_variant >= CID_HASH_VARIANTS) return fail(CID_ERR_UNSUPPORTED, "hash variant %d", hash_variant);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    ix->mod.flags = (ix->mod.flags & 0xFFu) | ((uint32_t)hash_variant << 8);
    ix->mod.xmul = hash_variant == CID_HASH_XXH3_V07 ? 0x165667B19E3779F9ULL : 0x165667919E3779F9ULL;
    return CID_OK;
}

int cid_index_put_rows(cid_index *ix, const uint64_t *row_ids, const uint32_t *words_le, size_t n_rows) {
    if (!ix || (n_rows && (!row_ids || !words_le))) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t tail_bits = ix->n_colors % 32;
    const uint32_t tail_mask = tail_bits ? ((1u << tail_bits) - 1u) : 0xFFFFFFFFu;
    for (size_t i = 0; i < n_rows; ++i) {
        if (row_ids[i] >= ix->m) return fail(CID_ERR_INVALID, "row id %llu >= bloom_size", (unsigned long long)row_ids[i]);
        if (words_le[i * ix->w32 + ix->w32 - 1] & ~tail_mask) return fail(CID_ERR_INVALID, "row %llu has bits beyond n_colors", (unsigned long long)row_ids[i]);
    }
    const size_t batch = 1u << 22;
    for (size_t r0 = 0; r0 < n_rows; r0 += batch) {
        const size_t nr = n_rows - r0 < batch ? n_rows - r0 : batch;
        void *d_ids, *d_words;
        int rc = slot_reserve(c, S_ROWIDS, nr * 8, &d_ids);
        if (rc) return rc;
        rc = slot_reserve(c, S_WORDS, nr * ix->w32 * 4, &d_words);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_ids, row_ids + r0, nr * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_words, words_le + r0 * ix->w32, nr * ix->w32 * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(cid::launch_put_rows(ix->mat, ix->rs, (const uint64_t *)d_ids, (const uint32_t *)d_words, ix->w32, nr, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CID_OK;
}

}  // extern "C"

// records of a file with n_colors_total colours; the index takes the colours [colour_base, colour_base + ix->n_colors) (colour_base a
// multiple of 32: whole u32 words)
int cid::index_put_records_slice(cid_index *ix, const uint8_t *records, size_t n_records, uint32_t n_colors_total, uint32_t colour_base) {
    if (!ix || (n_records && !records)) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (colour_base % 32u || (uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) of %u colours", colour_base, ix->n_colors, n_colors_total);
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t w32_rec = (n_colors_total + 31u) / 32u;
    const size_t rec_bytes = 24 + 4ull * w32_rec;
    const size_t batch = (256u << 20) / rec_bytes;   // records per upload
    for (size_t r0 = 0; r0 < n_records; r0 += batch) {
        const size_t nr = n_records - r0 < batch ? n_records - r0 : batch;
        void *d_rec, *d_err;
        int rc = slot_reserve(c, S_WORDS, nr * rec_bytes, &d_rec);
        if (rc) return rc;
        rc = slot_reserve(c, S_MISC, 16, &d_err);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_err, 0, 4, c->stream));
        HIP_TRY(hipMemcpyAsync(d_rec, records + r0 * rec_bytes, nr * rec_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(cid::launch_put_records(ix->mat, ix->rs, (const uint32_t *)d_rec, w32_rec, colour_base / 32u, ix->w32, nr, ix->m, n_colors_total,
                                        (uint32_t *)d_err, c->stream));
        uint32_t err = 0;
        HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (err)
            return fail(CID_ERR_INVALID, "malformed row record(s):%s%s%s%s", (err & 1) ? " word count != ceil(n_colors/32)" : "",
                        (err & 2) ? " bit count != n_colors" : "", (err & 4) ? " row >= bloom_size" : "", (err & 8) ? " bits beyond n_colors" : "");
    }
    return CID_OK;
}

extern "C" {

int cid_index_put_records(cid_index *ix, const uint8_t *records, size_t n_records) {
    if (!ix) return fail(CID_ERR_INVALID, "null argument");
    return cid::index_put_records_slice(ix, records, n_records, ix->n_colors, 0);
}

int cid_index_device_matrix(cid_index *ix, void **dev_ptr, uint64_t *row_stride_words) {
    if (!ix || !dev_ptr || !row_stride_words) return fail(CID_ERR_INVALID, "null argument");
    *dev_ptr = ix->mat;
    *row_stride_words = ix->rs;
    return CID_OK;
}

int cid_index_finalize(cid_index *ix) {
    if (!ix) return fail(CID_ERR_INVALID, "null index");
    HIP_TRY(hipSetDevice(ix->ctx->device));
    HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    ix->finalized = true;
    return CID_OK;
}

int cid_index_get_rows(const cid_index *ix, const uint64_t *row_ids, uint32_t *words_le, size_t n_rows) {
    if (!ix || (n_rows && (!row_ids || !words_le))) return fail(CID_ERR_INVALID, "null argument");
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    for (size_t i = 0; i < n_rows; ++i)
        if (row_ids[i] >= ix->m) return fail(CID_ERR_INVALID, "row id %llu >= bloom_size", (unsigned long long)row_ids[i]);
    void *d_ids, *d_words;
    int rc = slot_reserve(c, S_ROWIDS, n_rows * 8, &d_ids);
    if (rc) return rc;
    rc = slot_reserve(c, S_WORDS, n_rows * ix->w32 * 4, &d_words);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_ids, row_ids, n_rows * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cid::launch_get_rows(ix->mat, ix->rs, (const uint64_t *)d_ids, (uint32_t *)d_words, ix->w32, n_rows, c->stream));
    HIP_TRY(hipMemcpyAsync(words_le, d_words, n_rows * ix->w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

int cid_index_get_records(const cid_index *ix, uint64_t row_begin, uint64_t n_rows, uint8_t *records, uint64_t *n_records) {
    if (!ix || !n_records || (n_rows && !records)) return fail(CID_ERR_INVALID, "null argument");
    if (row_begin > ix->m || n_rows > ix->m - row_begin) return fail(CID_ERR_INVALID, "rows [%llu, +%llu) outside bloom_size",
                                                                     (unsigned long long)row_begin, (unsigned long long)n_rows);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    return cid::index_get_records(ix->ctx, ix, row_begin, n_rows, records, n_records);
}

int cid_index_insert_kmers_dev(cid_index *ix, const uint8_t *d_kmers, const uint32_t *d_colour_of_kmer, size_t n_kmers) {
    if (!ix || (n_kmers && (!d_kmers || !d_colour_of_kmer))) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (!aligned16(d_kmers)) return fail(CID_ERR_INVALID, "d_kmers must be 16-byte aligned");
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    cid::InsertParams p{};
    p.mat = ix->mat; p.rs = ix->rs; p.n_hash = ix->n_hash; p.k = ix->k; p.n_colors = ix->n_colors;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
    p.m_size = ix->m_size;
    p.mod = ix->mod; p.kmers = d_kmers; p.colour_of_kmer = d_colour_of_kmer; p.n_kmers = n_kmers;
    HIP_TRY(cid::launch_insert_kmers(p, c->stream));
    return CID_OK;
}

int cid_index_insert_kmers(cid_index *ix, const uint8_t *kmers, uint32_t colour, size_t n_kmers) {
    if (!ix || (n_kmers && !kmers)) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (colour >= ix->n_colors) return fail(CID_ERR_INVALID, "colour %u >= n_colors", colour);
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    void *d_k;
    int rc = slot_reserve(c, S_KMERS, n_kmers * ix->k, &d_k);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_k, kmers, n_kmers * ix->k, hipMemcpyHostToDevice, c->stream));
    cid::InsertParams p{};
    p.mat = ix->mat; p.rs = ix->rs; p.n_hash = ix->n_hash; p.k = ix->k; p.n_colors = ix->n_colors;
    p.tiles_per_block = pick_tiles_per_block(c, n_kmers);
    p.colour = colour; p.m_size = ix->m_size;
    p.mod = ix->mod; p.kmers = (const uint8_t *)d_k; p.colour_of_kmer = nullptr; p.n_kmers = n_kmers;
    HIP_TRY(cid::launch_insert_kmers(p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}

void cid_index_destroy(cid_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    if (ix->mat) (void)hipFree(ix->mat);
    delete ix;
}

// ------------------------------------------------------------------------------------------------ a5

}  // extern "C"
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
    p.mixed = kSearchMixed ? 1u : 0u;
    if (kSearchPersist && ix->rs <= 128 && n_kmers >= (1u << 16)) {   // persistent grid, one work queue per XCD (cid_search.hip)
        void *d_q;
        rc = slot_reserve(c, S_QUEUE, 8 * 128, &d_q); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_q, 0, 8 * 128, c->stream));
        p.queues = (uint32_t *)d_q;
        p.persist_grid = c->n_cu * cid::search_count_blocks_per_cu(p);
        if (p.persist_grid <= 0) p.queues = nullptr;
    }
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

int cid_index_row_stride_words(const cid_index *ix, uint64_t *row_stride_words) {
    if (!ix || !row_stride_words) return fail(CID_ERR_INVALID, "null argument");
    *row_stride_words = ix->rs;
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
    size_t chunk = kUploadChunkBytes / (k + 8);
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
int index_insert_codes(cid_index *ix, const uint64_t *d_codes, size_t n, uint32_t k, uint32_t colour) {
    if (!ix || (n && !d_codes)) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    if (colour >= ix->n_colors) return fail(CID_ERR_INVALID, "colour %u >= n_colors", colour);
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    cid::InsertParams p{};
    p.mat = ix->mat; p.rs = ix->rs; p.n_hash = ix->n_hash; p.k = ix->k; p.n_colors = ix->n_colors;
    p.tiles_per_block = pick_tiles_per_block(c, n);
    p.colour = colour; p.m_size = ix->m_size; p.mod = ix->mod; p.codes = d_codes; p.n_kmers = n;
    HIP_TRY(cid::launch_insert_kmers(p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}
int index_insert_ascii(cid_index *ix, const uint8_t *d_ascii, size_t n, uint32_t k, uint32_t colour) {
    if (!ix || (n && !d_ascii)) return fail(CID_ERR_INVALID, "null argument");
    if (ix->finalized) return fail(CID_ERR_STATE, "index already finalized");
    if (ix->k != k) return fail(CID_ERR_INVALID, "k-mer set k=%u, index k=%u", k, ix->k);
    if (colour >= ix->n_colors) return fail(CID_ERR_INVALID, "colour %u >= n_colors", colour);
    cid_ctx *c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    cid::InsertParams p{};
    p.mat = ix->mat; p.rs = ix->rs; p.n_hash = ix->n_hash; p.k = ix->k; p.n_colors = ix->n_colors;
    p.tiles_per_block = pick_tiles_per_block(c, n);
    p.colour = colour; p.m_size = ix->m_size; p.mod = ix->mod; p.kmers = d_ascii; p.n_kmers = n;
    HIP_TRY(cid::launch_insert_kmers(p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CID_OK;
}
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

// ------------------------------------------------------------------------------------------------ a6/a7/a9/a10

// LDS layout of k_readid (bytes_kernel = false) or k_readid_bytes for reads of at most max_bytes bases / max_win windows;
// returns the bytes one wave needs (the kernels carve the same regions in the same order)
static size_t readid_layout(const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win,
                            bool bytes_kernel, cid::ReadIdParams &p, bool packed_table = false) {
    p = cid::ReadIdParams{};
    p.mat = ix->mat; p.rs = ix->rs; p.w64 = ix->w64; p.n_colors = ix->n_colors; p.n_hash = ix->n_hash; p.k = ix->k;
    p.mod = ix->mod;
    p.stride_d = stride_d; p.start_sample = start_sample;
    p.m_size = ix->m_size;
    if (max_bytes > (1ull << 30) || max_win > (1ull << 30)) return ~(size_t)0;
    const bool wide = ix->rs > 128;
    p.bases_cap = (uint32_t)((max_bytes + 16 + 15) & ~15ull);
    p.win_cap = (uint32_t)((max_win + 3) & ~3ull);
    if (p.win_cap < 4) p.win_cap = 4;
    p.hist_pad = wide ? 4u * ix->rs : ((ix->n_colors + 1 + 3) & ~3u);   // wide rows: AND word + sampled-colour set
    p.table_slots = 64;
    while (p.table_slots < p.win_cap + p.win_cap / 2) p.table_slots <<= 1;
    size_t slot_bytes = 12;
    if (packed_table) {   // one u64 per slot: code << idx_bits | window index
        uint32_t ib = 1;
        while ((1ull << ib) <= p.win_cap) ++ib;
        if (2u * ix->k + ib > 63u) return ~(size_t)0;
        p.idx_bits = ib;
        slot_bytes = 8;
    }
    const size_t chunk_rows = 4ull * cid::kWave * ix->n_hash;                     // one chunk's row numbers
    const size_t rall_bytes = wide ? 0 : 4ull * p.win_cap * ix->n_hash;           // rows of the read's distinct k-mers (wide rows search chunk by chunk)
    size_t wave_bytes = (size_t)p.bases_cap + rall_bytes;
    if (bytes_kernel)   // histogram, tags, window infos, k-mer image (+ minimizer image and the distinct minimizer strings)
        wave_bytes += 4ull * p.hist_pad + chunk_rows + 8ull * p.win_cap + cid::kmer_img_bytes(ix->k) +
                      (ix->m_size ? cid::kmer_img_bytes(ix->m_size) + (((size_t)p.win_cap * ix->m_size + 15) & ~15ull) : 0);
    else if (wide)      // chunk rows, histogram, hash table keys + indices, 2-bit bases, bad-base bits
        wave_bytes += chunk_rows + 4ull * p.hist_pad + 12ull * p.table_slots + 4ull * (p.bases_cap / 16 + 4) + 4ull * (p.bases_cap / 32 + 4);
    else                // the histogram shares the hash table's region (k_readid)
        wave_bytes += (CID_READID_ALIAS ? std::max<size_t>(slot_bytes * p.table_slots, 4ull * p.hist_pad) : slot_bytes * p.table_slots + 4ull * p.hist_pad) +
                      4ull * (p.bases_cap / 16 + 4) + 4ull * (p.bases_cap / 32 + 4);
    wave_bytes = (wave_bytes + 15) & ~15ull;
    p.wave_bytes = (uint32_t)(wave_bytes < 0xFFFFFFF0ull ? wave_bytes : 0xFFFFFFF0ull);
    return wave_bytes;
}
// what a read needs of the LDS kernels: k <= 32 reads may end up in either of them
static size_t readid_need(const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win) {
    cid::ReadIdParams p;
    const size_t b = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, true, p);
    if (ix->k > 32) return b;
    const size_t a = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, false, p);
    return a > b ? a : b;
}

constexpr size_t kLdsBytes = 160u * 1024u;
// device scratch for dense read_id report rows per launch: cid_readid_count slices larger batches, the sparse form refuses them
static size_t kDenseReportBytes = getenv("CID_DENSE_REPORT_BYTES") ? strtoull(getenv("CID_DENSE_REPORT_BYTES"), nullptr, 10) : (2ull << 30);
// k_readid keeps a read's set in one wave's LDS.  With fewer than two waves per workgroup (one per CU) the gathers are no
// longer hidden and the sort-based path is faster (tools/bench_readlen.py: 150 Mbases of 4 kb reads 50.7 vs 24.5 ms; 2 kb
// reads, two waves, 30.3 vs 36.7 ms), so such reads are routed there.
constexpr size_t kLdsReadBytesMax = kLdsBytes / 2;

static bool kReadidPackedTable = getenv("CID_READID_PACKED_TABLE") ? atoi(getenv("CID_READID_PACKED_TABLE")) != 0 : true;   // cid_tune "readid_packed_table"
static int readid_params(const cid_index *ix, uint32_t stride_d, uint32_t start_sample, uint64_t max_bytes, uint64_t max_win,
                         bool bytes_kernel, cid::ReadIdParams &p, int &waves, bool striped = false) {
    size_t wave_bytes = 0, best = 0;
    auto choose = [&](bool packed_table) {
        wave_bytes = readid_layout(ix, stride_d, start_sample, max_bytes, max_win, bytes_kernel, p, packed_table);
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
    const bool can_pack = !bytes_kernel && kReadidPackedTable && ix->rs <= 128 && !ix->m_size && ix->k <= 31 && ((ix->mod.flags >> 8) & 0xFFu) == CID_HASH_XXH3_V08 && !striped;
    choose(false);
    if (best < 24 && can_pack) {   // (where six waves per SIMD fit anyway the 12-byte slots are marginally faster: 6.02 vs 6.07 ms single-end)
        const size_t classic = best;
        choose(true);
        if (best <= classic || best <= 20) choose(false);   // worth it only with more waves than the 96-VGPR (5 per SIMD) build runs
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
    int rc = readid_params(ix, stride_d, start_sample, max_read_bytes, max_read_windows, true, pb, waves_b);
    if (rc) return rc;
    const bool packable = ix->k <= 32;
    if (packable && (rc = readid_params(ix, stride_d, start_sample, max_read_bytes, max_read_windows, false, pp, waves_p, sa.zero_acc || sa.zero_in))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    auto fill = [&](cid::ReadIdParams &p, int waves) {
        p.bases = d_bases; p.seq_off = d_seq_off; p.read_seq0 = d_read_seq0; p.n_reads = n_reads;
        p.report = d_report; p.n_kmers = d_n_kmers; p.status = d_status; p.skip = d_skip;
        p.zero_acc = sa.zero_acc; p.zero_in = sa.zero_in; p.zero_start = sa.zero_start;
        p.colour_base = sa.colour_base; p.report_width = sa.report_width; p.write_nohits = sa.write_nohits;
        uint64_t rpb = n_reads / ((uint64_t)c->n_cu * 16);
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

int cid_readid_count_dev(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off,
                         const uint64_t *d_read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                         uint64_t max_read_bytes, uint64_t max_read_windows, uint32_t *d_report, uint32_t *d_n_kmers,
                         uint8_t *d_status) {
    int rc = check_ready(c, ix);
    if (rc) return rc;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    if (n_reads == 0) return CID_OK;
    if (!d_bases || !d_seq_off || !d_read_seq0 || !d_report || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
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
        return fail(CID_ERR_UNSUPPORTED, "reads of %llu bases do not fit a wave's LDS: cid_readid_stripe_zero / _count route such reads through the sort-based path",
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
    if ((uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) outside %u colours", colour_base, ix->n_colors, n_colors_total);
    if (n_reads == 0) return CID_OK;
    if (!d_zero_acc || !d_report || !d_n_kmers || !d_status) return fail(CID_ERR_INVALID, "null argument");
    StripeArgs sa;
    sa.zero_in = d_zero_acc; sa.zero_start = d_zs; sa.colour_base = colour_base; sa.report_width = n_colors_total + 1;
    sa.write_nohits = write_nohits ? 1u : 0u;
    return readid_dev_impl(c, ix, d_bases, d_seq_off, d_read_seq0, n_reads, stride_d, start_sample, max_read_bytes, max_read_windows, nullptr, false,
                           d_report, d_n_kmers, d_status, sa);
}

// Which kernel takes which read of a batch: route empty = the LDS kernels take all of them; else route[r] = 1 sends read r through
// the sort-based path.  max_bytes / max_win: the LDS sizing, over the reads the LDS kernels will see.  Validates the offsets.
struct ReadRoute {
    std::vector<uint8_t> route;
    uint64_t max_bytes = 0, max_win = 0;
    size_t n_long = 0;
};
static int readid_route(const cid_index *ix, const uint64_t *seq_off, size_t n_seqs, const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d,
                        uint32_t start_sample, ReadRoute &rr) {
    auto read_size = [&](size_t r, uint64_t &bytes, uint64_t &win) {
        const uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
        win = 0;
        for (uint64_t s = s0; s < s1; ++s) {
            const uint64_t len = seq_off[s + 1] - seq_off[s];
            if (len >= ix->k) win += (len - ix->k) / stride_d + 1;
        }
        bytes = s1 > s0 ? seq_off[s1] - seq_off[s0] : 0;
    };
    uint64_t max_bytes = 0, max_win = 0;
    for (size_t r = 0; r < n_reads; ++r) {   // (both bounds before seq_off is read through them)
        if (read_seq0[r + 1] < read_seq0[r]) return fail(CID_ERR_INVALID, "read_seq0 not monotonic at read %zu", r);
        if (read_seq0[r + 1] > n_seqs) return fail(CID_ERR_INVALID, "read_seq0 points past n_seqs at read %zu", r);
        for (uint64_t s = read_seq0[r]; s < read_seq0[r + 1]; ++s)
            if (seq_off[s + 1] < seq_off[s]) return fail(CID_ERR_INVALID, "seq_off not monotonic at seq %llu", (unsigned long long)s);
        uint64_t bytes, win;
        read_size(r, bytes, win);
        if (bytes > max_bytes) max_bytes = bytes;
        if (win > max_win) max_win = win;
    }
    // routing: reads whose set would leave k_readid fewer than two waves per workgroup go through the sort-based path
    rr.route.clear();
    rr.n_long = 0;
    if (readid_need(ix, stride_d, start_sample, max_bytes, max_win) > kLdsReadBytesMax) {
        rr.route.assign(n_reads, 0);
        max_bytes = max_win = 0;
        for (size_t r = 0; r < n_reads; ++r) {
            uint64_t bytes, win;
            read_size(r, bytes, win);
            if (readid_need(ix, stride_d, start_sample, bytes, win) > kLdsReadBytesMax) { rr.route[r] = 1; ++rr.n_long; }
            else {
                if (bytes > max_bytes) max_bytes = bytes;
                if (win > max_win) max_win = win;
            }
        }
    }
    rr.max_bytes = max_bytes; rr.max_win = max_win;
    return CID_OK;
}

// The two stripe passes for ANY read length and stripe width: d_bases resident, offsets on the host.  Per stripe the reads are routed
// between the LDS kernels and the sort-based path exactly as cid_readid_count routes them (the mask of a read's q-th distinct k-mer
// sits at the same word whichever kernel writes it, so different stripes may route a read differently).  Masks: one word per
// window, read r's at [prefix of the windows of reads 0..r-1] (cid_readid_stripe_mask_words words in all).
static int stripe_mask_starts(uint32_t k, uint32_t stride_d, const uint64_t *seq_off, uint64_t n_seqs, const uint64_t *read_seq0, size_t n_reads, std::vector<uint64_t> &zs) {
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
    if ((rc = readid_route(ix, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, rr))) return rc;
    std::vector<uint64_t> zs;
    if ((rc = stripe_mask_starts(ix->k, stride_d, seq_off, n_seqs, read_seq0, n_reads, zs))) return rc;
    if (zs[n_reads] >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 2^32 k-mer windows in one read_id batch");
    const bool all_long = rr.n_long == n_reads, mixed = rr.n_long > 0 && !all_long;
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
    if (rr.n_long) {   // first: it writes a status for every read (2 = the other kernels')
        HIP_TRY(hipStreamSynchronize(c->stream));
        rc = cid::readid_long(c, ix, d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample, mixed ? rr.route.data() : nullptr, false,
                              d_report, d_n_kmers, d_status, sa);
        if (rc) return rc;
    }
    if (!all_long) {
        void *d_skip = nullptr;
        if (mixed) {
            rc = slot_reserve(c, S_ROUTE, n_reads, &d_skip); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(d_skip, rr.route.data(), n_reads, hipMemcpyHostToDevice, c->stream));
        }
        rc = readid_dev_impl(c, ix, d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample, rr.max_bytes,
                             rr.max_win ? rr.max_win : 1, (const uint8_t *)d_skip, false, d_report, d_n_kmers, d_status, sa);
    }
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
    if (ix && (uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) outside %u colours", colour_base, ix->n_colors, n_colors_total);
    StripeArgs sa;
    sa.zero_in = d_zero_acc; sa.colour_base = colour_base; sa.report_width = n_colors_total + 1; sa.write_nohits = write_nohits ? 1u : 0u;
    return readid_stripe_pass(c, ix, d_bases, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, sa, d_report, d_n_kmers, d_status);
}

// uploads the batch, runs the LDS or the sort-based kernel; leaves report / n_kmers / status in the ctx's device scratch
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
    ReadRoute rr;
    if ((rc = readid_route(ix, seq_off, n_seqs, read_seq0, n_reads, stride_d, start_sample, rr))) return rc;
    const std::vector<uint8_t> &route = rr.route;
    const uint64_t max_bytes = rr.max_bytes, max_win = rr.max_win;
    const size_t n_long = rr.n_long;
    const bool all_long = n_long == n_reads, mixed = n_long > 0 && !all_long;
    HIP_TRY(hipSetDevice(c->device));
    void *d_bases, *d_so, *d_r0, *d_rep, *d_nk;
    const size_t C1 = (size_t)ix->n_colors + 1;
    rc = slot_reserve(c, S_BASES, total_bases, &d_bases); if (rc) return rc;
    rc = slot_reserve(c, S_SEQOFF, (n_seqs + 1) * 8, &d_so); if (rc) return rc;
    rc = slot_reserve(c, S_READ0, (n_reads + 1) * 8, &d_r0); if (rc) return rc;
    rc = slot_reserve(c, S_REPORT, n_reads * C1 * 4, &d_rep); if (rc) return rc;
    rc = slot_reserve(c, S_NK, n_reads * 4 + n_reads + 16, &d_nk); if (rc) return rc;
    {   // the batch goes through the ctx's pinned arena when it fits (cid::pin_reserve); the arena's tail is left for the results
        const size_t b_so = (total_bases + 15) & ~(size_t)15, b_r0 = b_so + (n_seqs + 1) * 8, b_end = b_r0 + (n_reads + 1) * 8;
        uint8_t *pin = cid::pin_reserve(c, b_end + n_reads * 5 + 64);
        if (pin) {
            HIP_TRY(hipStreamSynchronize(c->stream));   // (the arena may still feed the previous call's copies)
            if (total_bases) memcpy(pin, bases, total_bases);
            memcpy(pin + b_so, seq_off, (n_seqs + 1) * 8);
            memcpy(pin + b_r0, read_seq0, (n_reads + 1) * 8);
            if (total_bases) HIP_TRY(hipMemcpyAsync(d_bases, pin, total_bases, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_so, pin + b_so, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_r0, pin + b_r0, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
        } else {
            if (total_bases) HIP_TRY(hipMemcpyAsync(d_bases, bases, total_bases, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_so, seq_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_r0, read_seq0, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
        }
    }
    uint8_t *d_status = (uint8_t *)d_nk + n_reads * 4;
    if (ix->rs > 128 && mixed) HIP_TRY(hipMemsetAsync(d_rep, 0, n_reads * C1 * 4, c->stream));   // both kernels count in place
    if (n_long) {   // first: it writes a status for every read (2 = the other kernel's)
        HIP_TRY(hipStreamSynchronize(c->stream));
        rc = cid::readid_long(c, ix, (const uint8_t *)d_bases, seq_off, read_seq0, n_reads, stride_d, start_sample,
                              mixed ? route.data() : nullptr, !mixed, (uint32_t *)d_rep, (uint32_t *)d_nk, d_status);
        if (rc) return rc;
    }
    if (!all_long) {
        void *d_skip = nullptr;
        if (mixed) {
            rc = slot_reserve(c, S_ROUTE, n_reads, &d_skip); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(d_skip, route.data(), n_reads, hipMemcpyHostToDevice, c->stream));
        }
        rc = readid_dev_impl(c, ix, (const uint8_t *)d_bases, (const uint64_t *)d_so, (const uint64_t *)d_r0, n_reads, stride_d, start_sample,
                             max_bytes, max_win, (const uint8_t *)d_skip, !mixed, (uint32_t *)d_rep, (uint32_t *)d_nk, d_status);
        if (mixed) HIP_TRY(hipStreamSynchronize(c->stream));   // `route` leaves scope
    }
    if (rc) return rc;
    *d_report_out = (uint32_t *)d_rep; *d_nk_out = (uint32_t *)d_nk; *d_status_out = (uint8_t *)d_nk + n_reads * 4;
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
    size_t per = kDenseReportBytes / (C1 * 4);
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

// ------------------------------------------------------------------------------------------------ tunables + timing

int cid_tune(const char *name, long value) {
    if (!name) return fail(CID_ERR_INVALID, "null name");
    if (!strcmp(name, "search_persist")) { kSearchPersist = value != 0; return CID_OK; }
    if (!strcmp(name, "search_mixed")) { kSearchMixed = value != 0; return CID_OK; }
    if (!strcmp(name, "search_unroll")) { kSearchUnroll = (int)value; return CID_OK; }
    if (!strcmp(name, "readid_packed_table")) { kReadidPackedTable = value != 0; return CID_OK; }
    if (!strcmp(name, "order_bits")) { if (value < 0 || value > 32) return fail(CID_ERR_INVALID, "order_bits 0..32"); cid::g_order_bits = (int)value; return CID_OK; }
    return fail(CID_ERR_INVALID, "unknown tunable '%s'", name);
}

// The runtime loads a translation unit's device code on the first launch of one of its kernels — ~60 ms for the read_id kernels,
// paid inside the first cid_readid_count* call.  This call pays it ahead of time and may run on another host thread than the one
// using the ctx (it touches no stream, no ctx state): the CLI runs it beside the index load.
int cid_warmup(cid_ctx *c, unsigned what) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (what & CID_WARM_READID) HIP_TRY(cid::warm_readid());
    if (what & CID_WARM_SEARCH) HIP_TRY(cid::warm_search());
    if (what & (CID_WARM_READID | CID_WARM_SEARCH)) HIP_TRY(cid::warm_kmerset());
    if (what & CID_WARM_INFLATE) HIP_TRY(cid::warm_inflate());
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
