/*
 * colorid_hip.h — C ABI of libcolorid_hip.so: the MI355X (gfx950) implementation of colorid's
 * BIGSI query hot path.  Plain pointers and sizes only; a Rust `extern "C"` block (see
 * INTEGRATION.md) binds these symbols in place of the loops cited next to each entry point.
 * Citations are file:line in the reference tree (hcdenbakker/colorid @ 2024_10_08).
 *
 * Conventions
 *   - every function returns CID_OK (0) or a negative CID_ERR_*; cid_last_error() returns the
 *     calling thread's last message.  Nothing throws or aborts across the ABI.  (The reference
 *     panics on every error: src/bigsi.rs:60-61, src/kmer.rs:11.)
 *   - the caller owns every host buffer; the library owns device memory.
 *   - a cid_index is immutable and shareable after cid_index_finalize(); a cid_ctx is used by
 *     one host thread at a time.  Objects made from a ctx (cid_index, cid_kmerset) borrow its device
 *     scratch: destroy them before the ctx.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     CID_ERR_HIP.
 *   - k-mers are ASCII byte strings of exactly k_size bytes, hashed byte-for-byte
 *     (upper/lower case preserved); 1 <= k_size <= 128.
 *   - `*_dev` entry points take DEVICE pointers (16-byte aligned), enqueue on the ctx stream and
 *     return without synchronising; the plain forms take HOST pointers and are synchronous.
 */
#ifndef COLORID_HIP_H
#define COLORID_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the declarations between this push and the pop at the end of the
 * file are its dynamic symbols (csrc/export.map says the same to the linker). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* CORE and EXTENDED (round 6).  The declarations marked CID_CORE are the stable core of this ABI: SURVEY.md §8b's calls — context, index,
 * the three search calls on k-mers and on k-mer sets, per-read counts — plus what the one-GPU command line runs on (INTEGRATION.md §2-§5:
 * the k-mer set, the FASTQ front end, .bxi records).  Everything else is EXTENDED: device-pointer forms for callers that keep their data in
 * HBM, colour stripes, groups of GPUs, measurement helpers.  The core changes only with cid_abi_version(); extended entry points may gain
 * arguments between rounds.  include/colorid_hip.rs carries the same split as two modules. */
#define CID_CORE
#define CID_OK 0
#define CID_ERR_INVALID (-1)     /* bad argument */
#define CID_ERR_HIP (-2)         /* HIP runtime error / no device */
#define CID_ERR_NOMEM (-3)
#define CID_ERR_UNSUPPORTED (-4) /* e.g. k_size > 128, bloom_size > 2^32 */
#define CID_ERR_STATE (-5)       /* e.g. search on an index that is not finalized */

/* Hash variants.  The reference calls xxh3::hash64_with_seed(kmer, i) % bloom_size
 * (src/simple_bloom.rs:22-23; crate xxh3 ^0.1.1, not vendored, version unpinned).
 * CID_HASH_XXH3_V08 = the published XXH3_64bits_withSeed (xxHash >= 0.8.0).  The .bxi format
 * carries no hash id, so the caller states it. */
#define CID_HASH_XXH3_V08 0
/* CID_HASH_XXH3_V07 = the XXH3 draft of xxHash v0.7.1 / v0.7.2 (2019), the most likely code base of crate xxh3 0.1.x: for the
 * k-mer lengths 17..128 it is the v0.8 construction with the avalanche multiplier PRIME64_3; restated from memory, no known
 * answer available: a CANDIDATE that `colorid hashcheck` (cid_index_set_hash_variant + cid_search_count_set per accession)
 * confirms or rejects against a reference-built index — not a parity claim. */
#define CID_HASH_XXH3_V07 1
#define CID_HASH_VARIANTS 2

#define CID_NOT_UNIQUE 0xFFFFFFFFu

typedef struct cid_ctx cid_ctx;     /* one HIP device + stream + scratch */
typedef struct cid_index cid_index; /* device-resident dense bit matrix: bloom_size rows x n_colors bits */

CID_CORE const char *cid_last_error(void);
CID_CORE int cid_abi_version(void);   /* 4 (round 5: + cid_readid_count_resident; 3: per-context cid_ctx_tune) */
int cid_device_count(int *n_devices);

/* ---- context ---- */
CID_CORE int cid_ctx_create(int device_id, cid_ctx **out);
/* Page-locked host memory for buffers a host hands to the library over and over (batches of reads, a reader's text): copies from it
 * run at the bus rate; pageable memory is pinned and unpinned by the runtime around every copy.  Needs no ctx. */
CID_CORE int cid_pinned_alloc(size_t bytes, void **out);
CID_CORE void cid_pinned_free(void *p);
/* Borrow an existing hipStream_t (e.g. the caller's framework stream); NULL restores the ctx's own stream. */
int cid_ctx_set_stream(cid_ctx *, void *hip_stream);
int cid_ctx_synchronize(cid_ctx *);
CID_CORE void cid_ctx_destroy(cid_ctx *);

/* ---- index: replaces BigsyMapNew.map (src/bigsi.rs:19-27) as handed to the search functions
 *      (src/main.rs:600-625, :810-865).  Rows follow BitVec<u32> (bit-vec_serde/src/lib.rs:218-224,
 *      :465-474): bit c of a row = colour c = words[c/32] >> (c%32) & 1. ---- */
CID_CORE int cid_index_create(cid_ctx *, uint64_t bloom_size, uint32_t num_hash, uint32_t k_size, uint32_t n_colors,
                     int hash_variant, cid_index **out);
/* Minimizer index (BigsyMapMiniNew, src/bigsi.rs:40-49, the `.mxi` file): the Bloom key of a k-mer is
 * find_minimizer(kmer, m_size) (src/kmer.rs:971-986) instead of the k-mer.  Call once after create.  Affects
 * cid_index_insert_kmers* (src/build.rs:455-459) and cid_readid_count* (src/kmer.rs:363-394); the search entry points
 * refuse such an index, as the reference does (src/main.rs:569-573). */
CID_CORE int cid_index_set_minimizer(cid_index *, uint32_t m_size);
/* Re-interpret the same rows under another hash variant (the file carries no hash id; rows are rows).  Changes which rows
 * every later insert / search on this index computes; call it only while no call on the index is in flight. */
CID_CORE int cid_index_set_hash_variant(cid_index *, int hash_variant);
/* Sparse rows as the .bxi `map` stores them (src/bigsi.rs:59-63): n_rows x W32 little-endian
 * u32 words, W32 = ceil(n_colors/32).  Rows never put stay all-zero == key absent from the map. */
CID_CORE int cid_index_put_rows(cid_index *, const uint64_t *row_ids, const uint32_t *words_le, size_t n_rows);
/* The same rows as they sit in a .bxi/.mxi file: n_records consecutive bincode records
 * { u64 row ; u64 n_words ; n_words x u32 ; u64 n_bits }, 24 + 4*W32 bytes each.  The records are parsed
 * and checked on the device (n_words == W32, n_bits == n_colors, row < bloom_size, no bit beyond n_colors), so a loader
 * only has to read the file: CID_ERR_INVALID for a malformed record (the reference: "can't deserialize" panic, bigsi.rs:61). */
CID_CORE int cid_index_put_records(cid_index *, const uint8_t *records, size_t n_records);
/* Native device layout: row r at matrix + r*row_stride_words (u64 words, little-endian pairs of the u32 words,
 * zero padded).  Exposed so a caller can generate/fill an index in HBM directly (bits >= n_colors MUST be 0). */
int cid_index_device_matrix(cid_index *, void **dev_ptr, uint64_t *row_stride_words);
CID_CORE int cid_index_finalize(cid_index *);
/* Read rows back (host): words_le receives n_rows x W32 u32 words. */
int cid_index_get_rows(const cid_index *, const uint64_t *row_ids, uint32_t *words_le, size_t n_rows);
/* save_bigsi (src/bigsi.rs:51-57): the non-zero rows of [row_begin, row_begin + n_rows) in ascending order as the file's row
 * records (see cid_index_put_records), formatted on the device.  `records` must hold n_rows * (24 + 4*W32) bytes;
 * *n_records = how many were written (all-zero rows are not keys of the map, src/build.rs:123-127).  n_rows < 2^32. */
CID_CORE int cid_index_get_records(const cid_index *, uint64_t row_begin, uint64_t n_rows, uint8_t *records, uint64_t *n_records);
/* Bloom insert on device: simple_bloom.rs:19-26 for colour `colour` of every k-mer (used to build/plant
 * indices without leaving HBM; src/build.rs:116-128 transposed on the fly).  Before finalize only. */
int cid_index_insert_kmers_dev(cid_index *, const uint8_t *d_kmers, const uint32_t *d_colour_of_kmer,
                               size_t n_kmers);
/* Host-buffer form: every k-mer goes into one colour — one accession's BloomFilter::insert loop
 * (src/build.rs:62-66, :93-97). */
CID_CORE int cid_index_insert_kmers(cid_index *, const uint8_t *kmers, uint32_t colour, size_t n_kmers);
CID_CORE void cid_index_destroy(cid_index *);

/* ---- a5: proportional search, the hot loop of batch_search_pe::batch_search
 *      (src/batch_search_pe.rs:45-84 and :125-164).  For each distinct k-mer: n hashes -> n rows -> AND;
 *      hits[c] += 1 for every set colour c; if exactly one colour is set: n_unique[c] += 1,
 *      sum_unique_freq[c] += freq[kmer] (1 if freq == NULL), unique_colour[kmer] = c (else CID_NOT_UNIQUE).
 *      n_unique / sum_unique_freq / unique_colour may be NULL. ---- */
CID_CORE int cid_search_count(cid_ctx *, const cid_index *, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                     uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
int cid_search_count_dev(cid_ctx *, const cid_index *, const uint8_t *d_kmers, const uint32_t *d_freq,
                         size_t n_kmers, uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq,
                         uint32_t *d_unique_colour);

/* Same, for k-mers given as 2-bit codes (k_size <= 32): one u64 per canonical UPPER-CASE k-mer, base 0 in the most
 * significant of the 2*k_size low bits, A,C,G,T = 0..3 (8 bytes of input per k-mer instead of k_size). */
int cid_search_count_codes_dev(cid_ctx *, const cid_index *, const uint64_t *d_codes, const uint32_t *d_freq,
                               size_t n_kmers, uint64_t *d_hits, uint64_t *d_n_unique, uint64_t *d_sum_unique_freq,
                               uint32_t *d_unique_colour);

/* ---- colour-striped indices: one cid_index per stripe of colours [colour_base, colour_base+n_colors)
 *      of a wider index, several on one GPU and/or one (set of) stripe(s) per GPU (an index larger than one HBM).  Stripes of
 *      any width work, including more than 8192 colours.  Per-colour hits of a stripe are final.  "Exactly one colour hit"
 *      (batch_search_pe.rs:75-82) and "row absent" (perfect_search.rs:31-39) need every stripe: each stripe call merges
 *      per-k-mer facts into caller arrays —
 *        d_fact[n_kmers] (u32, zeroed before the first stripe):  n << 26 | (colour + 1),  n = min(set bits of the AND word, 2)
 *            over the stripes merged so far (saturating), the colour field kept only while n == 1.  Across GPUs the caller
 *            SUMS the arrays (one RCCL all-reduce of 4 bytes per k-mer; at most 31 ranks, at most 2^20 colours in total) and
 *            hands the result to cid_search_unique_finalize_dev: a k-mer is unique iff the summed n is 1, and then the low
 *            26 bits are its colour + 1.
 *        d_zero_acc[n_kmers] (u32, preset to 0xFFFFFFFF): bit s stays set while row s was all-zero in every stripe; AND
 *            across GPUs, "row absent" iff != 0 afterwards.
 *      Exactly one of d_kmers / d_codes is given. ---- */
int cid_search_count_stripe_dev(cid_ctx *, const cid_index *, const uint8_t *d_kmers, const uint64_t *d_codes, size_t n_kmers,
                                uint32_t colour_base, uint64_t *d_hits, uint32_t *d_fact);
int cid_search_unique_finalize_dev(cid_ctx *, const uint32_t *d_fact, const uint32_t *d_freq,
                                   size_t n_kmers, uint32_t n_colors_total, uint64_t *d_n_unique /* n_colors_total, zeroed */,
                                   uint64_t *d_sum_unique_freq, uint32_t *d_unique_colour);
int cid_search_perfect_stripe_dev(cid_ctx *, const cid_index *, const uint8_t *d_kmers, const uint64_t *d_codes, size_t n_kmers,
                                  uint64_t *d_and_words /* row_stride_words u64 */, uint32_t *d_zero_acc);
int cid_index_row_stride_words(const cid_index *, uint64_t *row_stride_words);
/* read_id over colour stripes (src/read_id_mt_pe.rs:66-165): per-colour counts of a stripe are independent of the other stripes
 * except for the rule "the first k-mer with an absent row ends the read" (:81-89, :126-128), where absent means all-zero in EVERY
 * stripe.  Two passes over the same resident read batch (arguments as cid_readid_count_dev; short reads, stripes of at most 8192
 * colours):
 *   zero pass, once per stripe: d_zero_acc[read * max_read_windows + q] (u32, preset to 0xFFFFFFFF) &= the seeds whose row is
 *     all-zero in this stripe, q = the k-mer's rank in the read's first-occurrence order.  Stripes on other GPUs: AND the arrays
 *     (all-gather + AND, or sum per-seed "non-zero" indicators) before the second pass.
 *   count pass, once per stripe: the ordered search with "absent" read from the masks; d_report has n_colors_total + 1 columns
 *     (zeroed by the caller before the first stripe), a stripe fills [colour_base, colour_base + n_colors) and — exactly one
 *     stripe, write_nohits != 0 — the last column (the reference's no_hits_num entry). */
int cid_readid_stripe_zero_dev(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0,
                               size_t n_reads, uint32_t stride_d, uint64_t max_read_bytes, uint64_t max_read_windows, uint32_t *d_zero_acc,
                               uint32_t *d_n_kmers, uint8_t *d_status);
int cid_readid_stripe_count_dev(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0,
                                size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint64_t max_read_bytes, uint64_t max_read_windows,
                                uint32_t colour_base, uint32_t n_colors_total, int write_nohits, const uint32_t *d_zero_acc, uint32_t *d_report,
                                uint32_t *d_n_kmers, uint8_t *d_status);
/* The same two passes for reads of ANY length (long reads, contigs, batches that mix them with short reads) over stripes of any
 * width: the bases are resident, the offsets are host arrays, and every call routes the batch between the LDS kernels and the
 * sort-based long-read path exactly as cid_readid_count does (src/read_id_mt_pe.rs:431-465 stream_fasta's records included).
 * The masks are one u32 per k-mer window, read r's at word [windows of reads 0..r-1] + q — the same word whichever kernel writes
 * it, so a read may be routed differently in stripes of different width; cid_readid_stripe_mask_words gives the array's length
 * (preset it to 0xFFFFFFFF; AND the arrays of other GPUs between the passes, as above). */
int cid_readid_stripe_mask_words(uint32_t k_size, uint32_t stride_d, const uint64_t *seq_off, const uint64_t *read_seq0, size_t n_reads,
                                 uint64_t *n_words);
int cid_readid_stripe_zero(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs,
                           const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t *d_zero_acc, uint32_t *d_n_kmers,
                           uint8_t *d_status);
int cid_readid_stripe_count(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs,
                            const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t colour_base,
                            uint32_t n_colors_total, int write_nohits, const uint32_t *d_zero_acc, uint32_t *d_report,
                            uint32_t *d_n_kmers, uint8_t *d_status);

/* ---- a10 (next row): canonical k-mer counting on the GPU — replaces the
 *      FnvHashMap<String,usize> producers of `search`: kmerize_vector (src/kmer.rs:87-125, mode 0: has_no_n filter,
 *      orientation chosen on the raw bytes, then upper-cased) and the fastq bodies (src/kmer.rs:481-503 / :619-647,
 *      mode 1: has_no_n filter, case preserved — a lower-case base cannot be packed and makes add_seqs return
 *      CID_ERR_UNSUPPORTED: count that file on the host).  Sequences: `bases` + seq_off[n_seqs+1] (for mode 1
 *      already quality-masked, src/seq.rs:36-56).  The set lives in HBM: windows -> 2-bit codes -> radix sort ->
 *      run-length.  clean == clean_map (src/kmer.rs:826-837: keep multiplicity > t); count_histogram gives the
 *      (multiplicity, number of k-mers) pairs auto_cutoff needs (src/kmer.rs:866-942).
 *      k_size 33..128: a k-mer no longer packs into 64 bits, the keys are byte strings (case kept in mode 1, so lower-case bases
 *      are fine there): the sequences stay resident until finalize, every window's key is sorted on a 4-bit-per-base image
 *      (one stable radix pass per 16 bases) and run-length counted; the finished set is n x k_size ASCII bytes.  Such a set
 *      serves every call below (the cid_group_*_set calls included) except cid_kmerset_device_arrays / _order_for_index. ---- */
typedef struct cid_kmerset cid_kmerset;
CID_CORE int cid_kmerset_create(cid_ctx *, uint32_t k_size, cid_kmerset **out);
CID_CORE int cid_kmerset_add_seqs(cid_kmerset *, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs, int mode);
/* the same for reads already in HBM (k_size <= 32): d_seq_off[n_seqs + 1] are offsets into d_bases, no sequence longer than max_len
 * (<= one segment of windows: reads, not genomes — CID_ERR_UNSUPPORTED otherwise) */
int cid_kmerset_add_seqs_dev(cid_kmerset *, const uint8_t *d_bases, const uint64_t *d_seq_off, size_t n_seqs, uint64_t max_len, int mode);
/* Optional, before the first cid_kmerset_add_seqs*: build the set FOR this index.  Every window then carries, next to its code, a 32-bit
 * key derived from the row its first hash selects in that index (seed 0 of src/batch_search_pe.rs:48-49), the set's own sort takes
 * the key's leading bits as its partition digits and finishes on (key, code): the set comes out ordered by (first row, code) instead
 * of by code, equal k-mers still adjacent.  The search's first-row fetches of neighbouring k-mers then share 128-byte lines of the
 * matrix (a fifth fewer line fetches at n = 4, 32-byte rows) and the ordering costs no pass of its own — only 4 more bytes per window
 * through the sort.  Contents, multiplicities and every result are unchanged (the reference iterates a hash map: order is
 * unspecified there).  No-op for byte-string sets (k_size > 32) and for bloom_size >= 2^32 - 1. */
CID_CORE int cid_kmerset_set_target_index(cid_kmerset *, const cid_index *);
CID_CORE int cid_kmerset_finalize(cid_kmerset *, uint64_t *n_distinct);
CID_CORE int cid_kmerset_size(const cid_kmerset *, uint64_t *n_distinct);
CID_CORE int cid_kmerset_count_histogram(const cid_kmerset *, uint32_t *multiplicity, uint64_t *n_kmers, size_t cap, size_t *n_bins);
CID_CORE int cid_kmerset_clean(cid_kmerset *, uint64_t t);
/* Optional: reorder the set by the 128-byte index line of each k-mer's first row (seed 0), so that consecutive
 * k-mers share that line (order is unspecified in the reference: it iterates a hash map). */
int cid_kmerset_order_for_index(cid_kmerset *, const cid_index *);
/* The same on plain device arrays (asynchronous on the ctx stream; d_counts / d_counts_out may be NULL): out = in grouped by the
 * index line of each k-mer's first row.  The search then finds a k-mer's first row in a line its neighbours have just fetched
 * (one quarter fewer HBM line fetches at n = 4); the grouping itself costs about half a search of the same k-mers, so it
 * pays when a set is searched more than once. */
int cid_order_codes_for_index_dev(cid_ctx *, const cid_index *, const uint64_t *d_codes, const uint32_t *d_counts, size_t n_kmers,
                                  uint64_t *d_codes_out, uint32_t *d_counts_out);
/* Host copies in set order: n_distinct x k_size ASCII bytes and/or multiplicities (either may be NULL).
 * cid_kmerset_order_for_index works on 2-bit-code sets and on byte-string sets (k_size > 32: the rows are permuted) alike. */
CID_CORE int cid_kmerset_download(const cid_kmerset *, uint8_t *kmers_ascii, uint32_t *counts);
int cid_kmerset_device_arrays(const cid_kmerset *, void **d_codes, void **d_counts, uint64_t *n_distinct);
/* the same for a byte-string set (k_size 33..128): n_distinct x k_size ASCII bytes, in set order */
int cid_kmerset_device_ascii(const cid_kmerset *, void **d_kmers_ascii, void **d_counts, uint64_t *n_distinct);
CID_CORE void cid_kmerset_destroy(cid_kmerset *);
/* Bloom insert of a whole finalized set into one colour: one accession of `build` without the k-mers leaving HBM
 * (src/build.rs:54-99 with the map on the GPU). */
CID_CORE int cid_index_insert_kmerset(cid_index *, const cid_kmerset *, uint32_t colour);
/* a5 / a4 over a finalized set (results in set order; unique_colour has n_distinct entries). */
CID_CORE int cid_search_count_set(cid_ctx *, const cid_index *, const cid_kmerset *, uint64_t *hits, uint64_t *n_unique,
                         uint64_t *sum_unique_freq, uint32_t *unique_colour);
CID_CORE int cid_search_perfect_set(cid_ctx *, const cid_index *, const cid_kmerset *, uint32_t *and_words_le, int *any_row_missing);
/* The same search with everything reports::generate_report prints (src/reports.rs:8-48) and nothing per k-mer: per colour the
 * hits, the number of k-mers that hit only that colour, the sum of their multiplicities (-> mean) and their MODE
 * (src/reports.rs:65-77; ties -> the smallest value, the reference's tie follows HashMap order) — 4 x n_colors u64 instead of
 * 8 bytes per k-mer crossing PCIe.  cid_unique_freq_modes_dev is the mode step alone on device arrays (d_freq NULL = all 1). */
CID_CORE int cid_search_count_set_report(cid_ctx *, const cid_index *, const cid_kmerset *, uint64_t *hits, uint64_t *n_unique,
                                uint64_t *sum_unique_freq, uint64_t *mode_unique_freq);
int cid_unique_freq_modes_dev(cid_ctx *, const uint32_t *d_unique_colour, const uint32_t *d_freq, size_t n_kmers, uint32_t n_colors,
                              uint64_t *d_modes);

/* ---- a4: perfect search, perfect_search::batch_search / batch_search_mf
 *      (src/perfect_search.rs:25-52, :83-110): AND of all n*K rows.  and_words_le: W32 u32 words;
 *      *any_row_missing = 1 iff some row is absent (the reference's "No perfect hits!"), in which case
 *      and_words_le is all zero. ---- */
CID_CORE int cid_search_perfect(cid_ctx *, const cid_index *, const uint8_t *kmers, size_t n_kmers,
                       uint32_t *and_words_le, int *any_row_missing);

/* ---- a6/a7/a9/a10: per-read classification counts, the body of read_id_mt_pe::parallel_vec before
 *      kmer_poll_plus (src/read_id_mt_pe.rs:300-331): too_short test on the first mate (:305), distinct
 *      canonical k-mers of the read(-pair) with stride d (src/kmer.rs:221-243, src/seq.rs:59-70), then
 *      search_index_classic (start_sample == 0, :66-102) or search_index (:104-165).
 *      Reads: `bases` concatenated (already quality-masked, src/seq.rs:36-56); seq s = bases[seq_off[s]..seq_off[s+1]);
 *      read r = seqs read_seq0[r]..read_seq0[r+1]-1 (1 = SE, 2 = PE).
 *      report: n_reads x (n_colors+1) counts, column n_colors = the reference's no_hits_num entry;
 *      n_kmers[r] = |k-mer set|; status[r] = 1 for too_short, else 0.
 *      k-mer iteration order is first occurrence (mate 1 then mate 2) — SURVEY.md App. B Q7. ---- */
CID_CORE int cid_readid_count(cid_ctx *, const cid_index *, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                     const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                     uint32_t *report, uint32_t *n_kmers, uint8_t *status);

/* Sparse form of the same call: report rows are compacted on the device to their non-zero (colour, count) entries in
 * ascending colour order, so only those cross PCIe (a row has n_colors+1 counters, a read hits a handful).  The result
 * stays in the ctx until the next call; fetch it with cid_readid_sparse_fetch into row_start[n_reads+1] and
 * colours/counts[*n_entries] (column n_colors, the no_hits_num entry, appears like any colour). */
CID_CORE int cid_readid_count_sparse(cid_ctx *, const cid_index *, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                            const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                            uint32_t *n_kmers, uint8_t *status, uint64_t *n_entries);
CID_CORE int cid_readid_sparse_fetch(cid_ctx *, uint64_t *row_start, uint32_t *colours, uint32_t *counts);

/* Device-pointer form: bases AND offsets in HBM (what cid_fastq_* hands over).  The caller states the longest read(-pair) of the
 * batch in bytes and in k-mer windows (sum over its mates of (len-k)/d+1 for len >= k).  A read that exceeds either is not
 * processed: its status is 3, its row and n_kmers are zero.  Reads of any length (round 6): when the stated maximum is beyond what
 * a wave's LDS holds well (about 900 bases at d = 1), the device itself routes every read — the long ones through the long-read
 * kernels (their work lists are made on the device), the others through the LDS kernels — and the call then waits for the stream
 * (the work lists' sizes come back, and reads with a lower-case base are redone on byte strings); with short reads only it is
 * asynchronous on the ctx stream, as before. */
int cid_readid_count_dev(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *d_seq_off,
                         const uint64_t *d_read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                         uint64_t max_read_bytes, uint64_t max_read_windows, uint32_t *d_report, uint32_t *d_n_kmers,
                         uint8_t *d_status);

/* The bases already in HBM (the caller uploaded or produced them), the offsets on the host: reads of ANY length — each read is
 * routed, inside the batch, to the per-wave LDS kernels (up to ~900 bases) or to the long-read path (cid_readlong.hip: per-read k-mer
 * sets by workgroup-wide LDS hash tables, the ordered search by slices of a read; src/read_id_mt_pe.rs:282-363 takes any read
 * length; src/kmer.rs:221-243).  A batch of short reads only is asynchronous on the ctx stream; one that holds long reads is waited
 * for ONCE, at the end of the call (the host-made work lists leave scope there, and a lower-case base among the long reads — their
 * case is kept — sends the batch through the byte-string path).  Results go to the caller's DEVICE arrays
 * report[n_reads x (n_colors+1)], n_kmers[n_reads], status[n_reads] (status 1 = too_short). */
int cid_readid_count_resident(cid_ctx *, const cid_index *, const uint8_t *d_bases, const uint64_t *seq_off, size_t n_seqs,
                              const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                              uint32_t *d_report, uint32_t *d_n_kmers, uint8_t *d_status);

/* ---- several GPUs of one node: the reference's parallel boundary is the rayon map over reads
 *      (src/read_id_mt_pe.rs:300-302, `-t`, src/main.rs:718-721); here the reads / distinct k-mers of a query are sharded in
 *      contiguous balanced ranges over N contexts, one per GPU, each with its own replica of the index (1.6-6.4 GB of 288 GB).
 *      The single exchange step is the sum of the 3*n_colors per-accession counters (hits | n_unique | sum_unique_freq):
 *      RCCL ncclAllReduce over xGMI on the ranks' streams (librccl is dlopen'ed on first use); if a device id is listed twice
 *      (several ranks on one GPU — RCCL refuses that) or COLORID_REDUCE=host is set, the 24*n_colors bytes per rank are summed
 *      through the host instead.  Perfect search ANDs the ranks' words on the host (RCCL has no bitwise reduction); read_id
 *      needs no exchange, its rows are concatenated in input order.  Results are identical to the single-GPU calls.
 *      A group and its ranks' contexts are used from one host thread at a time (the calls run one thread per rank inside).
 *      The library never redirects a file descriptor: RCCL may print its version banner (and NCCL_DEBUG output) to stdout while
 *      cid_group_create / cid_group_destroy make and free the communicators — never during a search; a host that prints result
 *      rows to stdout points fd 1 elsewhere around those two calls (colorid's CLI does, host/main.cpp). ---- */
typedef struct cid_group cid_group;
int cid_group_create(const int *device_ids, int n_devices, cid_group **out);
int cid_group_size(const cid_group *, int *n_ranks);
int cid_group_ctx(cid_group *, int rank, cid_ctx **out);   /* borrowed: load / build the index of rank 0 with it */
int cid_group_uses_rccl(const cid_group *, int *yes);
void cid_group_destroy(cid_group *);                         /* after the replicas and k-mer sets made from its contexts */
/* replicas[r] = an index on rank r's device holding src's rows: src itself where src lives on that rank's ctx, otherwise a
 * device-to-device copy (xGMI peer copy) that the caller destroys with cid_index_destroy. */
int cid_group_replicate_index(cid_group *, cid_index *src, cid_index **replicas /* n_ranks */);
/* a5 / a4 / a6-a10 with the work sharded over the ranks; arguments as in the single-GPU calls of the same name */
int cid_group_search_count(cid_group *, cid_index *const *replicas, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                           uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
int cid_group_search_count_set(cid_group *, cid_index *const *replicas, const cid_kmerset *, uint64_t *hits, uint64_t *n_unique,
                               uint64_t *sum_unique_freq, uint32_t *unique_colour);
int cid_group_search_perfect(cid_group *, cid_index *const *replicas, const uint8_t *kmers, size_t n_kmers, uint32_t *and_words_le,
                             int *any_row_missing);
int cid_group_search_perfect_set(cid_group *, cid_index *const *replicas, const cid_kmerset *, uint32_t *and_words_le, int *any_row_missing);
int cid_group_readid_count_sparse(cid_group *, cid_index *const *replicas, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                                  const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample, uint32_t *n_kmers,
                                  uint8_t *status, uint64_t *n_entries);
int cid_group_readid_sparse_fetch(cid_group *, uint64_t *row_start, uint32_t *colours, uint32_t *counts);

/* ---- canonical k-mer counting over the group (SURVEY.md §8e.1 caveat + §8f.1): the reference counts DISTINCT k-mers over the whole
 *      query, so a read set is not simply cut into shards — every rank counts the windows of its share of the sequences, the code
 *      space is cut into n_ranks ranges (splitters from the ranks' own quantiles), every range travels to its owner (peer copies over
 *      xGMI, 12 bytes per locally-distinct k-mer: the one exchange of this path) and is merged there.  Rank r then holds the r-th
 *      range, ascending: the parts laid end to end are the set in the order a cid_kmerset has.  k_size <= 32; the calls mirror
 *      cid_kmerset_* (add_seqs: CID_ERR_UNSUPPORTED on a lower-case base under mode 1, as there).  _search_*_parts: every rank
 *      searches its own part against its replica, only the 3*C counters are reduced; unique_colour in set order. ---- */
typedef struct cid_group_kmerset cid_group_kmerset;
int cid_group_kmerset_create(cid_group *, uint32_t k_size, cid_group_kmerset **out);
int cid_group_kmerset_add_seqs(cid_group_kmerset *, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs, int mode);
int cid_group_kmerset_finalize(cid_group_kmerset *, uint64_t *n_distinct);
int cid_group_kmerset_size(const cid_group_kmerset *, uint64_t *n_distinct);
int cid_group_kmerset_part_sizes(const cid_group_kmerset *, uint64_t *sizes /* n_ranks */);
int cid_group_kmerset_count_histogram(const cid_group_kmerset *, uint32_t *multiplicity, uint64_t *n_kmers, size_t cap, size_t *n_bins);
int cid_group_kmerset_clean(cid_group_kmerset *, uint64_t t);
int cid_group_kmerset_download(const cid_group_kmerset *, uint8_t *kmers_ascii, uint32_t *counts);
void cid_group_kmerset_destroy(cid_group_kmerset *);
int cid_group_search_count_parts(cid_group *, cid_index *const *replicas, const cid_group_kmerset *, uint64_t *hits, uint64_t *n_unique,
                                 uint64_t *sum_unique_freq, uint32_t *unique_colour);
/* the group form of cid_search_count_set_report: per colour hits, unique k-mers, their multiplicities' sum and MODE; nothing per k-mer
 * leaves the GPUs (every rank reduces its part to a (colour, multiplicity) histogram, the histograms add up on the host) */
int cid_group_search_count_parts_report(cid_group *, cid_index *const *replicas, const cid_group_kmerset *, uint64_t *hits, uint64_t *n_unique,
                                        uint64_t *sum_unique_freq, uint64_t *mode_unique_freq);
int cid_group_search_perfect_parts(cid_group *, cid_index *const *replicas, const cid_group_kmerset *, uint32_t *and_words_le, int *any_row_missing);

/* ---- colour stripes over a group (BASELINE configs[4]): rank r holds the colours [base_r, base_{r+1}) of EVERY row —
 *      an index larger than one GPU's HBM.  Every rank sees every query k-mer / read; per call ONE exchange: the packed per-k-mer
 *      facts summed (RCCL all-reduce, 4 bytes per k-mer), the perfect search's / read_id's zero-row masks ANDed on every rank
 *      (RCCL all-gather + a local AND; when device ids repeat both reductions run as a reduce-scatter / all-gather of peer copies
 *      on the ranks' own streams — no rank is a funnel).
 *      `stripes` = n_ranks handles, stripes[r] made from rank r's ctx; all but the last hold whole 64-colour words.
 *   create: balanced runs of 64-colour words (CID_ERR_INVALID when there are fewer words than ranks); fill with _put_records (the
 *      records of the whole .bxi: every rank keeps its own words) or _put_rows, then cid_index_finalize each stripe; destroy each
 *      with cid_index_destroy.  The call families mirror cid_group_search_* / cid_group_readid_*; outputs have n_colors_total
 *      entries (W32_total words), the sparse report's colours are global (no-hits entry = n_colors_total, last in its read).
 *      read_id takes reads of any length and stripes of any width (routed per stripe as cid_readid_stripe_zero / _count do). ---- */
int cid_group_stripes_create(cid_group *, uint64_t bloom_size, uint32_t num_hash, uint32_t k_size, uint32_t n_colors_total, int hash_variant,
                             cid_index **stripes /* n_ranks */);
int cid_group_stripes_base(const cid_group *, cid_index *const *stripes, uint32_t *colour_base /* n_ranks + 1 */);
int cid_group_stripes_put_records(cid_group *, cid_index *const *stripes, const uint8_t *records, size_t n_records);
int cid_group_stripes_put_rows(cid_group *, cid_index *const *stripes, const uint64_t *row_ids, const uint32_t *words_le /* n_rows x W32_total */,
                               size_t n_rows);
int cid_group_stripes_search_count(cid_group *, cid_index *const *stripes, const uint8_t *kmers, const uint32_t *freq, size_t n_kmers,
                                   uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
int cid_group_stripes_search_count_set(cid_group *, cid_index *const *stripes, const cid_kmerset *, uint64_t *hits, uint64_t *n_unique,
                                       uint64_t *sum_unique_freq, uint32_t *unique_colour);
/* the striped form of cid_search_count_set_report (per colour hits, unique k-mers, sum and MODE of their multiplicities) */
int cid_group_stripes_search_count_set_report(cid_group *, cid_index *const *stripes, const cid_kmerset *, uint64_t *hits, uint64_t *n_unique,
                                              uint64_t *sum_unique_freq, uint64_t *mode_unique_freq);
int cid_group_stripes_search_perfect(cid_group *, cid_index *const *stripes, const uint8_t *kmers, size_t n_kmers, uint32_t *and_words_le,
                                     int *any_row_missing);
int cid_group_stripes_search_perfect_set(cid_group *, cid_index *const *stripes, const cid_kmerset *, uint32_t *and_words_le, int *any_row_missing);
/* fetch the lists with cid_group_readid_sparse_fetch */
int cid_group_stripes_readid_count_sparse(cid_group *, cid_index *const *stripes, const uint8_t *bases, const uint64_t *seq_off, size_t n_seqs,
                                          const uint64_t *read_seq0, size_t n_reads, uint32_t stride_d, uint32_t start_sample,
                                          uint32_t *n_kmers, uint8_t *status, uint64_t *n_entries);

/* ---- measurement helpers (bench only): HIP-event timing on the ctx stream ---- */
/* Measurement / test switches of ONE context (no process-wide state; read once from the environment variable named in brackets when
 * the ctx is made).  None of them changes a result.  Unknown names give CID_ERR_INVALID:
 *   "search_unroll"        1/2    k_search_count on 64- and 128-byte rows: sub-passes whose row loads are issued together (default 2)
 *                                 [CID_SEARCH_UNROLL]
 *   "readid_packed_table"  0/1    k_readid: 8-byte (k <= 27 or so) or 4-byte (position only) k-mer-set slots where they buy a sixth wave per SIMD;
 *                                 0 keeps the 12-byte slots (default 1) [CID_READID_PACKED_TABLE]
 *   "readid_blocks_per_cu" 1..4096  k_readid: a batch is cut into about this many workgroups per CU (default 64)
 *   "order_bits"           0..32  cid_kmerset_order_for_index / cid_order_codes_for_index_dev group by this many leading bits of the
 *                                 first row's position (0 = by its exact 128-byte line, the default)
 *   "search_persist", "search_mixed"  0/1  the two measured-and-REJECTED schedulings of k_search_count (persistent grid with one work
 *                                 queue per XCD; each k-mer's last 32-byte row through the scalar cache).  Their kernels are only in
 *                                 libcolorid_hip_tune.so (`make -C colorid_amd/csrc tune`); the shipped library answers
 *                                 CID_ERR_UNSUPPORTED. */
int cid_ctx_tune(cid_ctx *, const char *name, long value);
/* Loads the device code of the read_id and/or search kernels now instead of inside the first call that launches one of them (the
 * runtime loads a code object on first use: ~60 ms for the read_id kernels).  Touches no stream and no ctx state: it may run on
 * another host thread while the ctx loads an index (what the CLI does). */
/* ---- input side: block-gzip (BGZF) members inflated on the GPU.  A BGZF file (bgzip, htslib, Illumina's
 *      converters) is a series of independent gzip members of at most 64 KiB of text, each with its compressed size in a "BC"
 *      extra field and its text size (ISIZE) in its trailer; the reference inflates such a file like any gzip stream, on one
 *      thread (flate2 MultiGzDecoder, src/read_id_mt_pe.rs:848-856, src/kmer.rs:469-476).  `members` holds n_members whole members
 *      (header, DEFLATE data, CRC-32, ISIZE) at member_off / member_len; member i's text (text_len[i] = its ISIZE) is written to
 *      text + text_off[i].  Every member is checked as zlib checks it — block structure, Huffman codes, ISIZE and CRC-32; the
 *      first bad one is reported (CID_ERR_INVALID, *bad_member = its index).  Host buffers in and out; one call is one batch. ---- */
CID_CORE int cid_bgzf_inflate(cid_ctx *, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                     const uint32_t *text_off, const uint32_t *text_len, size_t n_members, uint8_t *text, size_t text_bytes,
                     size_t *bad_member);
/* The same call in two halves, for a reader that keeps two batches in flight on two contexts: _start queues the upload, the kernel
 * and the copies back (the caller's `members` buffer is free again when it returns); _finish waits, checks and hands out the text.
 * Between the two the batch lives in the ctx's scratch: every other host-pointer entry point on the SAME ctx (cid_search_count,
 * cid_search_perfect, cid_readid_count*, cid_readid_stripe_*, cid_index_insert_kmers, a second _start ...) fails with CID_ERR_STATE
 * instead of overwriting it; the *_dev calls on caller-owned device memory and cid_readid_sparse_fetch stay usable. */
int cid_bgzf_inflate_start(cid_ctx *, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                           const uint32_t *text_off, const uint32_t *text_len, size_t n_members, size_t text_bytes);
int cid_bgzf_inflate_finish(cid_ctx *, uint8_t *text, size_t text_bytes, size_t *bad_member);
/* ---- the FASTQ front end of read_id on the device: what the reference does per read before its search — inflate
 *      (src/read_id_mt_pe.rs:848-856), take the lines four at a time (:862-879 / pairs :927-975: header, sequence, '+', quality; lines()
 *      strips "\n" and "\r\n"; an unterminated last line counts), seq::qual_mask (src/seq.rs:36-56) and the (id, [seq(, mate)]) batch
 *      — done for a whole stretch of the input at once, in HBM: the compressed bytes go up, and per read the id line, n_kmers,
 *      status and the non-zero (colour, count) entries come back; the reads themselves never exist in host memory.
 *        create        n_files = 1 (single-end) or 2 (record r of either file = read pair r); quality = -Q (0: no masking)
 *        push_bgzf     the next whole block-gzip members of one file (as cid_bgzf_inflate takes them; text_len[i] = member i's ISIZE).
 *                      Returns once the bytes are on the device: the members are inflated on a stream of the reader's own, beside
 *                      whatever the ctx stream is doing — push stretch i + 1, THEN classify stretch i (max_pushes = 1), and the
 *                      serial latency of DEFLATE (a launch takes ~14 ms however small) hides behind the classification before it.
 *                      push_text: already-decoded text instead (plain FASTQ, a gzip stream or block-gzip members inflated by the
 *                      host), cut anywhere.  Pushes of either kind wait their turn in order (up to 64 per file): a host with
 *                      spare cores inflates part of a stretch's members itself and lets the device take the rest.
 *                      flags: CID_FASTQ_LAST = the file ends with this push.
 *        classify      per file the oldest max_pushes waiting pushes (<= 0: all of them) join the text; every complete
 *                      record then held (for pairs: as many as both files hold) goes through cid_readid_count_dev's kernels and
 *                      the sparse-report compaction; what is left of the text waits on the device for the next call.
 *                      Corrupt members -> CID_ERR_INVALID naming the first one; a quality line longer than its sequence ->
 *                      CID_ERR_INVALID (the reference's "could not get the next nt" panic); records of ANY length are taken (round 6: reads too
 *                      long for a wave's LDS go through the long-read kernels inside the same step, cid_readid_count_dev's routing;
 *                      until round 5 such a step was refused with CID_ERR_UNSUPPORTED).
 *        classify_begin / classify_end   the same step in two halves, for a caller that has something to do while the classifier runs:
 *                      _begin takes the pushes, cuts and packs the records and LAUNCHES the classifier; _end waits for it, compacts
 *                      the report and publishes the step's results (sizes as classify returns them).  Between the two the caller
 *                      may push the next stretch and fetch the step BEFORE: a step's results stay fetchable until the next _end,
 *                      and fetch travels on a stream of its own (it does not wait for the classifier in flight).  One step in flight
 *                      at a time: _begin, then _end; classify = _begin + _end.
 *        fetch         n_kmers / status [n_reads], row_start [n_reads + 1] + colours / counts [n_entries] as cid_readid_sparse_fetch
 *                      gives them, id_off [n_reads + 1] and ids [id_bytes]: read r's header line (with its '@'), NUL-terminated,
 *                      at ids + id_off[r].
 *        count_kmers   the same step with a k-mer set as the sink instead of an index: the whole records pushed so far are masked and
 *                      packed as for classify, and every read of either file adds its k-mers to `set` (cid_kmerset_add_seqs_dev, mode 1:
 *                      the fastq producers of `search`, src/kmer.rs:461-510 single-end / :581-655 pairs — the walk over a pair of files
 *                      ends with the shorter one).  Nothing to fetch.  CID_ERR_UNSUPPORTED: a read holds a lower-case base (its case
 *                      would have to be kept: count that input through cid_kmerset_add_seqs / on the host) or is longer than a segment.
 *      One cid_fastq per input (pair); it borrows the ctx's stream and scratch: destroy it before the ctx. ---- */
typedef struct cid_fastq cid_fastq;
#define CID_FASTQ_LAST 1   /* push flags: the file ends with this push */
#define CID_FASTQ_KEEP 2   /* push_text: the buffer (page-locked, cid_pinned_alloc) stays untouched until the next push on this file or the
                            * classify call that takes this one — the copy then runs beside the caller instead of being waited for.
                            * push_bgzf: the same for `members` (the three arrays are read before the call returns), until the next
                            * push_bgzf on this file or the destruction of the reader */
CID_CORE int cid_fastq_create(cid_ctx *, int n_files, uint32_t quality, cid_fastq **out);
CID_CORE int cid_fastq_push_bgzf(cid_fastq *, int file, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                        const uint32_t *text_len, size_t n_members, int flags);
CID_CORE int cid_fastq_push_text(cid_fastq *, int file, const uint8_t *text, size_t n_bytes, int flags);
CID_CORE int cid_fastq_classify(cid_fastq *, const cid_index *, uint32_t stride_d, uint32_t start_sample, int max_pushes, uint64_t *n_reads,
                       uint64_t *n_entries, uint64_t *id_bytes);
CID_CORE int cid_fastq_classify_begin(cid_fastq *, const cid_index *, uint32_t stride_d, uint32_t start_sample, int max_pushes);
CID_CORE int cid_fastq_classify_end(cid_fastq *, uint64_t *n_reads, uint64_t *n_entries, uint64_t *id_bytes);
CID_CORE int cid_fastq_count_kmers(cid_fastq *, cid_kmerset *set, int max_pushes, uint64_t *n_reads);
CID_CORE int cid_fastq_fetch(cid_fastq *, uint32_t *n_kmers, uint8_t *status, uint64_t *row_start, uint32_t *colours, uint32_t *counts, uint64_t *id_off,
                    char *ids);
CID_CORE void cid_fastq_destroy(cid_fastq *);
#define CID_WARM_READID 1u
#define CID_WARM_SEARCH 2u
#define CID_WARM_INFLATE 4u
#define CID_WARM_FASTQ 8u /* the FASTQ front end (cid_fastq_*): its record / packing kernels and scans */
#define CID_WARM_PIPES 32u /* with CID_WARM_SEARCH and/or CID_WARM_READID: a query of a few reads on a context of its own, a command on each of
                            * this context's queues and 3 x 32 MB over the bus — what the first call after an idle start pays beyond its code
                            * objects (9-10 ms of a 16 ms search, profiles/r06_first_use.txt).  For callers whose context sits idle before its
                            * first query; a command line that has just uploaded an index gains nothing (its bus and queues are awake) */
#define CID_WARM_COLD 16u /* the rocPRIM-built cold paths (10 MB of device code, ~40 ms): read_id's sorting path — long reads with k > 32, soft-masked
                           * reads of more than 16 384 windows — byte-string k-mer sets, reordering */
CID_CORE int cid_warmup(cid_ctx *, unsigned what);
int cid_timer_start(cid_ctx *);
int cid_timer_stop_ms(cid_ctx *, float *elapsed_ms); /* synchronises on the stop event */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
