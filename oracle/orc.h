/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Plain-C restatement of colorid's BIGSI query path (reference @ 2024_10_08); every function
 * cites the reference file:line it follows.  See orc_xxh3.c for the parity status of the hash.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hash (3rd-party crate xxh3 ^0.1.1; restated from the published XXH3 spec) ---- */
uint64_t orc_xxh3_64_with_seed(const uint8_t *in, size_t len, uint64_t seed);            /* the variant selected below */
uint64_t orc_xxh3_published_64_with_seed(const uint8_t *in, size_t len, uint64_t seed);  /* variant 0: xxHash >= 0.8 */
uint64_t orc_xxh3_v07_64_with_seed(const uint8_t *in, size_t len, uint64_t seed);        /* variant 1: v0.7.1/0.7.2 draft, <= 128 bytes, unverified */
void orc_set_hash_variant(int v);                                                        /* process-wide; every later hash uses it */
int orc_get_hash_variant(void);

/* ---- src/seq.rs ---- */
int orc_is_good_base(uint8_t c);                                  /* seq.rs:59-64 */
int orc_has_no_n(const uint8_t *s, size_t len);                   /* seq.rs:66-70 */
void orc_qual_mask(const uint8_t *seq, const uint8_t *qual, size_t len, uint8_t q, uint8_t *out); /* seq.rs:36-56 */
void orc_revcomp(const uint8_t *in, size_t len, uint8_t *out);    /* kmer.rs:839-863 */

/* ---- k-mer maps (FnvHashMap<String,usize> / FnvHashSet<String>), iteration = first-occurrence order ---- */
typedef struct orc_kmers orc_kmers;
orc_kmers *orc_kmers_new(uint32_t k);
void orc_kmers_free(orc_kmers *);
uint64_t orc_kmers_len(const orc_kmers *);
const uint8_t *orc_kmers_keys(const orc_kmers *);   /* len*k bytes */
const uint64_t *orc_kmers_counts(const orc_kmers *);
void orc_kmers_insert(orc_kmers *, const uint8_t *key, uint64_t add);
int orc_kmerize_vector(orc_kmers *, const uint8_t *l, size_t len, size_t d);           /* kmer.rs:87-125 (one string of v) */
int orc_kmerize_string(orc_kmers *, const uint8_t *l, size_t len);                     /* kmer.rs:271-299; -1 = None */
int orc_kmerize_skip_n_set(orc_kmers *, const uint8_t *l, size_t len, size_t d);       /* kmer.rs:221-243 (one string of v) */
int orc_kmerize_fq_read(orc_kmers *, const uint8_t *seq, const uint8_t *qual, size_t len, uint8_t q); /* kmer.rs:479-504 / 619-647 body */
orc_kmers *orc_clean_map(const orc_kmers *, uint64_t t);                               /* kmer.rs:826-837 */
int64_t orc_auto_cutoff(const orc_kmers *);                                            /* kmer.rs:866-942; -1 = reference would panic */

/* ---- FASTA / FASTQ(.gz) readers ---- */
typedef struct orc_strvec { size_t n; char **s; size_t *len; } orc_strvec;
void orc_strvec_free(orc_strvec *);
int orc_read_fasta(const char *path, orc_strvec *seqs);                                /* kmer.rs:10-45 */
int orc_read_fasta_mf(const char *path, orc_strvec *labels, orc_strvec *seqs);         /* kmer.rs:47-84 */
orc_kmers *orc_kmers_from_fq_qual(const char *path, uint32_t k, uint8_t q);            /* kmer.rs:461-510 */
orc_kmers *orc_kmers_fq_pe_qual(const char *p1, const char *p2, uint32_t k, uint8_t q);/* kmer.rs:581-655 */

/* ---- index model: BigsyMapNew (bigsi.rs:19-27) with BitVec<u32> rows (bit-vec lib.rs:218-224) ---- */
typedef struct orc_index {
    uint64_t bloom_size, num_hash, k_size, n_colors;
    uint64_t m_size;         /* 0 = BigsyMapNew (.bxi); > 0 = BigsyMapMiniNew (.mxi, bigsi.rs:40-49): keys are minimizers */
    uint32_t w32;            /* blocks_for_bits(C) */
    uint32_t *rows;          /* dense bloom_size x w32; an all-zero row == key absent from `map` (build.rs:123-127) */
    char **colors;           /* colour id -> accession */
    uint64_t *n_ref_kmers;   /* by colour id */
} orc_index;
orc_index *orc_index_new(uint64_t m, uint64_t n_hash, uint64_t k, uint64_t n_colors);
void orc_index_free(orc_index *);
void orc_index_set_color(orc_index *, uint64_t c, const char *name, uint64_t n_ref_kmers);
const uint32_t *orc_index_rows(const orc_index *);
uint32_t *orc_index_rows_mut(orc_index *);
/* simple_bloom.rs:19-26 applied to colour c's column (== BloomFilter::insert then transpose, build.rs:116-128) */
void orc_index_insert(orc_index *, uint64_t c, const uint8_t *kmer);
int orc_index_contains(const orc_index *, uint64_t c, const uint8_t *kmer);           /* simple_bloom.rs:28-38 */
/* build.rs:33-130 (FASTA and fastq.gz accessions, colours = rank in sorted names) */
orc_index *orc_build_single(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint8_t quality, int64_t cutoff);
/* build.rs:396-492 (build_single_mini): the same k-mer maps, each distinct k-mer contributes find_minimizer(kmer, m) */
orc_index *orc_build_single_mini(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint64_t m_size, uint8_t quality, int64_t cutoff);
void orc_find_minimizer(const uint8_t *kmer, size_t k, size_t m, uint8_t *out);        /* kmer.rs:971-986 */
int orc_minimerize_skip_n_set(orc_kmers *set_of_minimizers, const uint8_t *l, size_t len, size_t k, size_t d);  /* kmer.rs:363-394 (one string) */
int orc_save_bigsi(const char *path, const orc_index *);                               /* bigsi.rs:51-57 / 71-77 + bincode 1.x layout */
orc_index *orc_read_bigsi(const char *path);                                           /* bigsi.rs:59-63 */

/* ---- search loops ---- */
/* batch_search_pe.rs:45-84 / 125-164.  unique_colour[i] = colour if hits.len()==1 else 0xFFFFFFFF (may be NULL) */
void orc_search_count(const orc_index *, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers,
                      uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
/* the same loop over n_threads POSIX threads (the bench's "best-effort CPU" figure; the reference is single-threaded here) */
void orc_search_count_mt(const orc_index *, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers, int n_threads,
                         uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
/* the same loop over the reference's data structure: FNV-hashed map row -> bit vector, one heap clone per k-mer (the
 * bench's "faithful structure" CPU figure) */
typedef struct orc_sparse orc_sparse;
orc_sparse *orc_sparse_build(const orc_index *);
void orc_sparse_free(orc_sparse *);
void orc_search_count_sparse(const orc_sparse *, const orc_index *, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers,
                             uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq);
/* perfect_search.rs:25-52: AND of all n*K rows; *missing = 1 iff some row is absent ("No perfect hits!") */
void orc_search_perfect(const orc_index *, const uint8_t *kmers, uint64_t n_kmers, uint32_t *and_words, int *missing);
/* read_id_mt_pe.rs:66-102 ; report has C+1 entries, [C] = no_hits_num */
void orc_search_index_classic(const orc_index *, const uint8_t *kmers, uint64_t n_kmers, uint64_t *report);
/* read_id_mt_pe.rs:104-165 */
void orc_search_index(const orc_index *, const uint8_t *kmers, uint64_t n_kmers, uint64_t start_sample, uint64_t *report);
/* read_id_mt_pe.rs:282-363 without the poll: per read k-mer set + search; status 0 ok, 1 too_short.
 * seq_off[n_seqs+1] into bases; read r = seqs read_seq0[r] .. read_seq0[r+1]-1.  report: n_reads x (C+1) */
void orc_readid_counts(const orc_index *, const uint8_t *bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                       uint64_t n_reads, uint64_t d, uint64_t start_sample,
                       uint32_t *report, uint32_t *n_kmers, uint8_t *status);
/* the same on n_threads threads (the reference's rayon pool over the batch, read_id_mt_pe.rs:300) */
void orc_readid_counts_mt(const orc_index *, const uint8_t *bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                          uint64_t n_reads, uint64_t d, uint64_t start_sample, int n_threads,
                          uint32_t *report, uint32_t *n_kmers, uint8_t *status);
double orc_false_prob(double m, double k, double n);                                   /* read_id_mt_pe.rs:695-698 */
/* read_id_mt_pe.rs:168-181 (binomial pmf via lgamma; crate `probability` is unpinned) */
int orc_not_fp_significant(uint64_t observations, double p_false, double fp_correct, uint64_t taxon_hits);
/* read_id_mt_pe.rs:187-251 on a dense report (ties keep ascending colour id).  label buffer >= label_cap.
 * returns 0; outputs: label, count, kmer_length passthrough, accept(1)/reject(0), n_top */
int orc_kmer_poll_plus(const orc_index *, const uint64_t *report, uint64_t kmer_length, double fp_correct,
                       char *label, size_t label_cap, uint64_t *count, int *accept, uint64_t *n_top);

/* reports.rs:8-48 / 50-62 / 65-77: rows appended to buf in ascending colour id; returns bytes written (or needed) */
size_t orc_generate_report(const orc_index *, const char *query, const uint64_t *hits, const uint64_t *n_unique,
                           const uint64_t *sum_unique_freq, const uint64_t *modes, uint64_t num_kmers, double cov,
                           char *buf, size_t cap);
size_t orc_generate_report_gene(const orc_index *, const char *query, const uint64_t *hits, uint64_t num_kmers,
                                double cov, char *buf, size_t cap);
/* reports.rs:65-77 per colour from (unique_colour, freq); ties -> smallest value (reference: arbitrary) */
void orc_unique_modes(const uint32_t *unique_colour, const uint64_t *freq, uint64_t n_kmers, uint64_t n_colors, uint64_t *modes);

#ifdef __cplusplus
}
#endif
#endif
