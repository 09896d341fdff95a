// C ABI of libcolorid_hip.so, part 2: the device-resident index — BigsyMapNew.map (src/bigsi.rs:19-27) as a dense bit matrix:
// create / fill from .bxi rows and records / read back / Bloom inserts (src/simple_bloom.rs:19-26).  Kernels: cid_index.hip.
#include "cid_api_common.hpp"

using cid::aligned16;
using cid::fail;
using cid::pick_tiles_per_block;
using cid::slot_reserve;
using namespace cid::slots;

namespace cid {
uint32_t index_k(const cid_index *ix) { return ix->k; }
uint32_t index_rs(const cid_index *ix) { return ix->rs; }
ModMagic index_mod(const cid_index *ix) { return ix->mod; }
uint32_t index_n_colors(const cid_index *ix) { return ix->n_colors; }
uint32_t index_n_hash(const cid_index *ix) { return ix->n_hash; }
uint32_t index_m_size(const cid_index *ix) { return ix->m_size; }
const uint64_t *index_matrix(const cid_index *ix) { return ix->mat; }

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

extern "C" {

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
    if (colour_base % 32u || (uint64_t)colour_base + ix->n_colors > n_colors_total) return fail(CID_ERR_INVALID, "stripe [%u, +%u) of %u colours",
        colour_base, ix->n_colors, n_colors_total);
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

int cid_index_row_stride_words(const cid_index *ix, uint64_t *row_stride_words) {
    if (!ix || !row_stride_words) return fail(CID_ERR_INVALID, "null argument");
    *row_stride_words = ix->rs;
    return CID_OK;
}

}  // extern "C"

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
}}  // namespace cid
