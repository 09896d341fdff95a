/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Plain-C, single-threaded restatement of colorid's BIGSI query path (reference @ 2024_10_08).
 * Written from the reference's behaviour, one function per reference function, each citing the
 * file:line it follows.  Deliberately simple (byte strings, per-bit scans) — it is the checker.
 *
 * Normalisations of reference non-determinism, stated once:
 *   - k-mer maps/sets iterate in FIRST-OCCURRENCE order (the reference iterates Rust hash order);
 *   - report rows / ties are emitted in ascending colour id (the reference: RandomState order);
 *   - mode() ties resolve to the smallest value (the reference: arbitrary).
 */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ src/seq.rs */

int orc_is_good_base(uint8_t c) { /* seq.rs:59-64 */
    switch (c) {
    case 'a': case 'c': case 'g': case 't': case 'A': case 'C': case 'G': case 'T': return 1;
    default: return 0;
    }
}

int orc_has_no_n(const uint8_t *s, size_t len) { /* seq.rs:66-70 */
    for (size_t i = 0; i < len; ++i)
        if (!orc_is_good_base(s[i])) return 0;
    return 1;
}

void orc_qual_mask(const uint8_t *seq, const uint8_t *qual, size_t len, uint8_t q, uint8_t *out) { /* seq.rs:36-56 */
    if (q == 0) { memcpy(out, seq, len); return; }
    uint8_t max_quality = (uint8_t)(q + 33);
    for (size_t i = 0; i < len; ++i) out[i] = (qual[i] < max_quality) ? 'N' : seq[i];
}

static uint8_t switch_base(uint8_t c) { /* kmer.rs:847-863 */
    switch (c) {
    case 'a': return 't'; case 'c': return 'g'; case 't': return 'a'; case 'g': return 'c';
    case 'u': return 'a'; case 'n': return 'n';
    case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C';
    case 'U': return 'A'; case 'N': return 'N';
    default: return 'N';
    }
}

void orc_revcomp(const uint8_t *in, size_t len, uint8_t *out) { /* kmer.rs:839-845 */
    for (size_t i = 0; i < len; ++i) out[i] = switch_base(in[len - 1 - i]);
}

/* ------------------------------------------------------------------ k-mer map */

struct orc_kmers {
    uint32_t k;
    uint64_t n, cap;       /* entries */
    uint8_t *keys;         /* n*k */
    uint64_t *counts;
    uint64_t tcap;         /* table size, power of two */
    uint64_t *table;       /* entry index + 1, 0 = empty */
};

static uint64_t fnv1a(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

orc_kmers *orc_kmers_new(uint32_t k) {
    orc_kmers *m = (orc_kmers *)calloc(1, sizeof(*m));
    m->k = k; m->cap = 1024; m->tcap = 4096;
    m->keys = (uint8_t *)malloc(m->cap * (size_t)k);
    m->counts = (uint64_t *)malloc(m->cap * sizeof(uint64_t));
    m->table = (uint64_t *)calloc(m->tcap, sizeof(uint64_t));
    return m;
}
void orc_kmers_free(orc_kmers *m) {
    if (!m) return;
    free(m->keys); free(m->counts); free(m->table); free(m);
}
uint64_t orc_kmers_len(const orc_kmers *m) { return m->n; }
const uint8_t *orc_kmers_keys(const orc_kmers *m) { return m->keys; }
const uint64_t *orc_kmers_counts(const orc_kmers *m) { return m->counts; }

static void kmers_rehash(orc_kmers *m) {
    free(m->table);
    m->tcap *= 2;
    m->table = (uint64_t *)calloc(m->tcap, sizeof(uint64_t));
    for (uint64_t e = 0; e < m->n; ++e) {
        uint64_t h = fnv1a(m->keys + e * m->k, m->k) & (m->tcap - 1);
        while (m->table[h]) h = (h + 1) & (m->tcap - 1);
        m->table[h] = e + 1;
    }
}

void orc_kmers_insert(orc_kmers *m, const uint8_t *key, uint64_t add) {
    uint64_t h = fnv1a(key, m->k) & (m->tcap - 1);
    while (m->table[h]) {
        uint64_t e = m->table[h] - 1;
        if (memcmp(m->keys + e * m->k, key, m->k) == 0) { m->counts[e] += add; return; }
        h = (h + 1) & (m->tcap - 1);
    }
    if (m->n == m->cap) {
        m->cap *= 2;
        m->keys = (uint8_t *)realloc(m->keys, m->cap * (size_t)m->k);
        m->counts = (uint64_t *)realloc(m->counts, m->cap * sizeof(uint64_t));
    }
    memcpy(m->keys + m->n * m->k, key, m->k);
    m->counts[m->n] = add;
    m->table[h] = ++m->n;
    if (m->n * 2 > m->tcap) kmers_rehash(m);
}

static void upper_copy(uint8_t *dst, const uint8_t *src, size_t k) { /* String::to_uppercase, ASCII */
    for (size_t i = 0; i < k; ++i) dst[i] = (src[i] >= 'a' && src[i] <= 'z') ? (uint8_t)(src[i] - 32) : src[i];
}

/* shared window walk: filter_n = apply seq::has_no_n; upper = .to_uppercase() on the chosen string */
static void walk_windows(orc_kmers *m, const uint8_t *l, size_t len, size_t d, int filter_n, int upper) {
    size_t k = m->k;
    uint8_t *l_r = (uint8_t *)malloc(len ? len : 1);
    uint8_t *tmp = (uint8_t *)malloc(k);
    orc_revcomp(l, len, l_r);
    for (size_t i = 0; i + k <= len; i += d) {
        const uint8_t *fwd = l + i;
        const uint8_t *rc = l_r + (len - (i + k));
        if (filter_n && !orc_has_no_n(fwd, k)) continue;
        const uint8_t *pick = (memcmp(fwd, rc, k) < 0) ? fwd : rc; /* `l[i..i+k] < l_r[..]` on raw bytes */
        if (upper) { upper_copy(tmp, pick, k); orc_kmers_insert(m, tmp, 1); }
        else orc_kmers_insert(m, pick, 1);
    }
    free(l_r); free(tmp);
}

int orc_kmerize_vector(orc_kmers *m, const uint8_t *l, size_t len, size_t d) { /* kmer.rs:87-125 */
    if (len < m->k) return 0;
    walk_windows(m, l, len, d, 1, 1);
    return 0;
}

int orc_kmerize_string(orc_kmers *m, const uint8_t *l, size_t len) { /* kmer.rs:271-299: no has_no_n filter */
    if (len < m->k) return -1; /* None */
    walk_windows(m, l, len, 1, 0, 1);
    return 0;
}

int orc_kmerize_skip_n_set(orc_kmers *m, const uint8_t *l, size_t len, size_t d) { /* kmer.rs:221-243: set, no upper-casing */
    if (len < m->k) return -1; /* the reference underflows `l.len() - k + 1` and panics: skip */
    walk_windows(m, l, len, d, 1, 0);
    return 0;
}

int orc_kmerize_fq_read(orc_kmers *m, const uint8_t *seq, const uint8_t *qual, size_t len, uint8_t q) {
    /* kmer.rs:481-503 (SE) / 619-647 (per mate): qual_mask, skip if shorter than k, d = 1, raw-case k-mers */
    uint8_t *masked = (uint8_t *)malloc(len ? len : 1);
    orc_qual_mask(seq, qual, len, q, masked);
    if (len >= m->k) walk_windows(m, masked, len, 1, 1, 0);
    free(masked);
    return 0;
}

void orc_find_minimizer(const uint8_t *seq, size_t k, size_t m, uint8_t *out) { /* kmer.rs:971-986 */
    uint8_t *r_seq = (uint8_t *)malloc(k ? k : 1);
    orc_revcomp(seq, k, r_seq);
    const uint8_t *minmer = seq;                                    /* &seq[..m]: the rc of position 0 is never tried */
    for (size_t i = 1; i + m <= k; ++i) {
        const uint8_t *min_f = seq + i, *min_rc = r_seq + (k - (i + m));
        if (memcmp(min_f, minmer, m) < 0) minmer = min_f;
        if (memcmp(min_rc, minmer, m) < 0) minmer = min_rc;
    }
    memcpy(out, minmer, m);
    free(r_seq);
}

int orc_minimerize_skip_n_set(orc_kmers *set, const uint8_t *l, size_t len, size_t k, size_t d) { /* kmer.rs:363-394; set->k == m */
    if (len < k) return 0;                                          /* `continue` */
    const size_t m = set->k;
    uint8_t *l_r = (uint8_t *)malloc(len), *mn = (uint8_t *)malloc(m), *up = (uint8_t *)malloc(m);
    orc_revcomp(l, len, l_r);
    for (size_t i = 0; i + k <= len; i += d) {
        const uint8_t *fwd = l + i, *rc = l_r + (len - (i + k));
        if (!orc_has_no_n(fwd, k)) continue;
        orc_find_minimizer(memcmp(fwd, rc, k) < 0 ? fwd : rc, k, m, mn);
        upper_copy(up, mn, m);                                       /* min.to_uppercase() */
        orc_kmers_insert(set, up, 1);
    }
    free(l_r); free(mn); free(up);
    return 0;
}

orc_kmers *orc_clean_map(const orc_kmers *m, uint64_t t) { /* kmer.rs:826-837: keep value > t */
    orc_kmers *o = orc_kmers_new(m->k);
    for (uint64_t e = 0; e < m->n; ++e)
        if (m->counts[e] > t) orc_kmers_insert(o, m->keys + e * m->k, m->counts[e]);
    return o;
}

int64_t orc_auto_cutoff(const orc_kmers *m) { /* kmer.rs:866-942 */
    uint64_t max_cov = 0;
    for (uint64_t e = 0; e < m->n; ++e) if (m->counts[e] > max_cov) max_cov = m->counts[e];
    uint64_t *histo = (uint64_t *)calloc(max_cov + 2, sizeof(uint64_t));
    for (uint64_t e = 0; e < m->n; ++e) histo[m->counts[e]]++;
    uint64_t sum = 0;
    for (uint64_t i = 0; i <= max_cov; ++i) sum += i * histo[i];
    double total_mean = (double)sum / (double)m->n;
    int64_t result;
    if (total_mean < 1.5) { free(histo); return 0; }
    /* coverages[j] = #k-mers with multiplicity j+1, j = 0..max_cov-2  (`for c in 1..max_cov`) */
    size_t ncov = max_cov >= 1 ? (size_t)(max_cov - 1) : 0;
    if (ncov == 0) { free(histo); return -1; }          /* `coverages.len() - 1` underflows: panic */
    const uint64_t *cov = histo + 1;
    size_t nd1 = ncov >= 2 ? ncov - 2 : 0;              /* for i in 1..len-1 */
    double *d1 = (double *)malloc((nd1 + 1) * sizeof(double));
    for (size_t i = 1; i + 1 < ncov; ++i) d1[i - 1] = (double)cov[i] / (double)cov[i + 1];
    if (nd1 == 0) { free(d1); free(histo); return -1; } /* `d1.len() - 1` underflows: panic */
    size_t nd2 = nd1 - 1;
    size_t first_pos_d1 = 0, first_pos_d2 = 0;
    for (size_t i = 0; i < nd1; ++i) if (d1[i] < 1.0) { first_pos_d1 = i + 1; break; }
    for (size_t i = 0; i < nd2; ++i) { double v = d1[i] / d1[i + 1]; if (v < 1.0) { first_pos_d2 = i + 1; break; } }
    uint64_t bigsum = 0, num = 0;
    for (size_t i = 0; i + 1 < ncov; ++i) { bigsum += (uint64_t)i * cov[1 + i]; num += cov[1 + i]; }
    double mean = (double)bigsum / (double)num;
    if (first_pos_d1 > 0 && (double)first_pos_d1 < mean * 0.75) result = (int64_t)first_pos_d1;
    else if (first_pos_d2 > 0) result = (int64_t)first_pos_d2;
    else {
        double c = ceil(mean / 2.0);
        uint64_t cu = (c != c || c <= 0.0) ? 0 : (uint64_t)c; /* Rust `as usize`: NaN -> 0, saturating */
        result = (int64_t)(cu > 1 ? cu : 1);
    }
    free(d1); free(histo);
    return result;
}

/* ------------------------------------------------------------------ readers */

void orc_strvec_free(orc_strvec *v) {
    for (size_t i = 0; i < v->n; ++i) free(v->s[i]);
    free(v->s); free(v->len); v->s = NULL; v->len = NULL; v->n = 0;
}
static void strvec_push(orc_strvec *v, const char *s, size_t len) {
    v->s = (char **)realloc(v->s, (v->n + 1) * sizeof(char *));
    v->len = (size_t *)realloc(v->len, (v->n + 1) * sizeof(size_t));
    v->s[v->n] = (char *)malloc(len + 1);
    memcpy(v->s[v->n], s, len); v->s[v->n][len] = 0;
    v->len[v->n] = len; v->n++;
}

typedef struct { char *p; size_t len, cap; } sbuf;
static void sbuf_add(sbuf *b, const char *s, size_t n) {
    if (b->len + n + 1 > b->cap) { b->cap = (b->len + n + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
    memcpy(b->p + b->len, s, n); b->len += n; b->p[b->len] = 0;
}

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *p = (char *)malloc((size_t)n + 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(p); return NULL; }
    fclose(f); p[n] = 0; *len = (size_t)n;
    return p;
}

/* str::lines(): split at \n, strip one trailing \r, no empty last line after a final \n */
static size_t next_line(const char *p, size_t n, size_t pos, size_t *b, size_t *e) {
    size_t i = pos;
    while (i < n && p[i] != '\n') ++i;
    *b = pos; *e = i;
    if (*e > *b && p[*e - 1] == '\r') --*e;
    return i < n ? i + 1 : n;
}

static int read_fasta_impl(const char *path, orc_strvec *labels, orc_strvec *seqs) { /* kmer.rs:10-45 / 47-84 */
    size_t n; char *c = slurp(path, &n);
    if (!c) return -1;
    size_t nlines = 0, pos = 0, b, e;
    while (pos < n) { pos = next_line(c, n, pos, &b, &e); ++nlines; }
    sbuf sub = {0};
    size_t count_line = 0; pos = 0;
    while (pos < n) {
        pos = next_line(c, n, pos, &b, &e);
        ++count_line;
        if (memchr(c + b, '>', e - b)) {               /* line.contains('>') */
            if (labels) strvec_push(labels, c + b + 1, e > b ? e - b - 1 : 0); /* line[1..] */
            if (sub.len > 0) strvec_push(seqs, sub.p, sub.len);
            sub.len = 0;
        } else if (count_line == nlines) {
            sbuf_add(&sub, c + b, e - b);
            if (sub.len > 0) strvec_push(seqs, sub.p, sub.len);
        } else {
            sbuf_add(&sub, c + b, e - b);
        }
    }
    free(sub.p); free(c);
    return 0;
}
int orc_read_fasta(const char *path, orc_strvec *seqs) { return read_fasta_impl(path, NULL, seqs); }
int orc_read_fasta_mf(const char *path, orc_strvec *labels, orc_strvec *seqs) { return read_fasta_impl(path, labels, seqs); }

/* BufRead::lines() over MultiGzDecoder: returns 1 and the line without \n / \r\n, 0 at EOF */
static int gz_line(gzFile f, sbuf *b) {
    char tmp[65536];
    b->len = 0; int got = 0;
    while (gzgets(f, tmp, (int)sizeof(tmp))) {
        size_t l = strlen(tmp); got = 1;
        sbuf_add(b, tmp, l);
        if (l && tmp[l - 1] == '\n') break;
    }
    if (!got) return 0;
    if (b->len && b->p[b->len - 1] == '\n') b->p[--b->len] = 0;
    if (b->len && b->p[b->len - 1] == '\r') b->p[--b->len] = 0;
    return 1;
}

orc_kmers *orc_kmers_from_fq_qual(const char *path, uint32_t k, uint8_t q) { /* kmer.rs:461-510 */
    gzFile f = gzopen(path, "rb");
    if (!f) return NULL;
    orc_kmers *m = orc_kmers_new(k);
    sbuf line = {0}, seq = {0};
    uint64_t line_count = 1;
    while (gz_line(f, &line)) {
        if (line_count % 4 == 2) { seq.len = 0; sbuf_add(&seq, line.p ? line.p : "", line.len); }
        else if (line_count % 4 == 0) {
            /* qual_mask walks qual.chars() and consumes seq chars: output length = qual length */
            size_t l = line.len <= seq.len ? line.len : seq.len;
            if (q == 0) {  /* qual_mask returns seq unchanged, whatever the quality string's length */
                if (seq.len >= m->k) walk_windows(m, (const uint8_t *)seq.p, seq.len, 1, 1, 0);
            } else
            orc_kmerize_fq_read(m, (const uint8_t *)seq.p, (const uint8_t *)line.p, l, q);
        }
        ++line_count;
    }
    gzclose(f); free(line.p); free(seq.p);
    return m;
}

orc_kmers *orc_kmers_fq_pe_qual(const char *p1, const char *p2, uint32_t k, uint8_t q) { /* kmer.rs:581-655 */
    gzFile f1 = gzopen(p1, "rb"), f2 = gzopen(p2, "rb");
    if (!f1 || !f2) { if (f1) gzclose(f1); if (f2) gzclose(f2); return NULL; }
    orc_kmers *m = orc_kmers_new(k);
    sbuf l1 = {0}, l2 = {0}, s1 = {0}, s2 = {0};
    uint64_t line_count = 1;
    while (gz_line(f1, &l1)) {
        if (!gz_line(f2, &l2)) break;                   /* `None => break` */
        if (line_count % 4 == 2) {
            s1.len = 0; sbuf_add(&s1, l1.p ? l1.p : "", l1.len);
            s2.len = 0; sbuf_add(&s2, l2.p ? l2.p : "", l2.len);
        } else if (line_count % 4 == 0) {
            size_t a = l1.len <= s1.len ? l1.len : s1.len, b = l2.len <= s2.len ? l2.len : s2.len;
            if (q == 0) {
                if (s1.len >= m->k) walk_windows(m, (const uint8_t *)s1.p, s1.len, 1, 1, 0);
                if (s2.len >= m->k) walk_windows(m, (const uint8_t *)s2.p, s2.len, 1, 1, 0);
            } else {
            orc_kmerize_fq_read(m, (const uint8_t *)s1.p, (const uint8_t *)l1.p, a, q);
            orc_kmerize_fq_read(m, (const uint8_t *)s2.p, (const uint8_t *)l2.p, b, q);
            }
        }
        ++line_count;
    }
    gzclose(f1); gzclose(f2); free(l1.p); free(l2.p); free(s1.p); free(s2.p);
    return m;
}

/* ------------------------------------------------------------------ index */

orc_index *orc_index_new(uint64_t m, uint64_t n_hash, uint64_t k, uint64_t n_colors) {
    orc_index *ix = (orc_index *)calloc(1, sizeof(*ix));
    ix->bloom_size = m; ix->num_hash = n_hash; ix->k_size = k; ix->n_colors = n_colors;
    ix->w32 = (uint32_t)((n_colors + 31) / 32);          /* bit-vec blocks_for_bits */
    ix->rows = (uint32_t *)calloc((size_t)m * (ix->w32 ? ix->w32 : 1), sizeof(uint32_t));
    ix->colors = (char **)calloc(n_colors ? n_colors : 1, sizeof(char *));
    ix->n_ref_kmers = (uint64_t *)calloc(n_colors ? n_colors : 1, sizeof(uint64_t));
    return ix;
}
void orc_index_free(orc_index *ix) {
    if (!ix) return;
    for (uint64_t c = 0; c < ix->n_colors; ++c) free(ix->colors[c]);
    free(ix->colors); free(ix->n_ref_kmers); free(ix->rows); free(ix);
}
void orc_index_set_color(orc_index *ix, uint64_t c, const char *name, uint64_t n_ref) {
    free(ix->colors[c]);
    ix->colors[c] = strdup(name);
    ix->n_ref_kmers[c] = n_ref;
}
const uint32_t *orc_index_rows(const orc_index *ix) { return ix->rows; }
uint32_t *orc_index_rows_mut(orc_index *ix) { return ix->rows; }

static inline uint64_t bit_index(const orc_index *ix, const uint8_t *key, uint64_t i) {
    /* xxh3::hash64_with_seed(&k.as_bytes(), i as u64) % bloom_size as u64 — the key is a k-mer, or a minimizer in a .mxi */
    return orc_xxh3_64_with_seed(key, (size_t)(ix->m_size ? ix->m_size : ix->k_size), i) % ix->bloom_size;
}
static inline const uint32_t *row_ptr(const orc_index *ix, uint64_t r) { return ix->rows + r * ix->w32; }
static inline int row_absent(const orc_index *ix, uint64_t r) {  /* !bigsi_map.contains_key(&bi) */
    const uint32_t *p = row_ptr(ix, r);
    for (uint32_t w = 0; w < ix->w32; ++w) if (p[w]) return 0;
    return 1;
}
static inline int bit_get(const uint32_t *row, uint64_t c) { return (row[c / 32] >> (c % 32)) & 1; } /* lib.rs:465-474 */

void orc_index_insert(orc_index *ix, uint64_t c, const uint8_t *kmer) { /* simple_bloom.rs:19-26 + build.rs:116-128 */
    uint8_t mini[256];
    if (ix->m_size) {                                               /* build.rs:455-459: filter.insert(&find_minimizer(&kmer, m)) */
        orc_find_minimizer(kmer, (size_t)ix->k_size, (size_t)ix->m_size, mini);
        kmer = mini;
    }
    for (uint64_t i = 0; i < ix->num_hash; ++i) {
        uint64_t r = bit_index(ix, kmer, i);
        ix->rows[r * ix->w32 + c / 32] |= 1u << (c % 32);     /* lib.rs:492-500 */
    }
}
int orc_index_contains(const orc_index *ix, uint64_t c, const uint8_t *kmer) { /* simple_bloom.rs:28-38 */
    for (uint64_t i = 0; i < ix->num_hash; ++i)
        if (!bit_get(row_ptr(ix, bit_index(ix, kmer, i)), c)) return 0;
    return 1;
}

typedef struct { char *name; char *f1; char *f2; } acc_t;
static int acc_cmp(const void *a, const void *b) { return strcmp(((const acc_t *)a)->name, ((const acc_t *)b)->name); }

static orc_index *build_single_impl(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint64_t m_size, uint8_t quality, int64_t cutoff);
orc_index *orc_build_single(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint8_t quality, int64_t cutoff) {
    return build_single_impl(ref_tsv, m, n_hash, k, 0, quality, cutoff);
}
orc_index *orc_build_single_mini(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint64_t m_size, uint8_t quality, int64_t cutoff) {
    return build_single_impl(ref_tsv, m, n_hash, k, m_size, quality, cutoff);
}
static orc_index *build_single_impl(const char *ref_tsv, uint64_t m, uint64_t n_hash, uint64_t k, uint64_t m_size, uint8_t quality, int64_t cutoff) {
    /* build.rs:15-31 (tab_to_map) + build.rs:33-130; with m_size > 0: build.rs:396-492 (same maps, minimizers inserted;
     * n_ref_kmers is only recorded for FASTA accessions there — SURVEY App. B Q13) */
    size_t n; char *c = slurp(ref_tsv, &n);
    if (!c) return NULL;
    acc_t *acc = NULL; size_t na = 0;
    size_t pos = 0, b, e;
    while (pos < n) {
        pos = next_line(c, n, pos, &b, &e);
        char *line = strndup(c + b, e - b);
        char *t1 = strchr(line, '\t');
        if (!t1) { free(line); continue; }                 /* reference would panic on v[1] */
        *t1 = 0;
        char *t2 = strchr(t1 + 1, '\t');
        if (t2) { *t2 = 0; char *t3 = strchr(t2 + 1, '\t'); if (t3) *t3 = 0; }
        size_t j;
        for (j = 0; j < na; ++j) if (strcmp(acc[j].name, line) == 0) break; /* map.insert overwrites */
        if (j == na) { acc = (acc_t *)realloc(acc, (na + 1) * sizeof(acc_t)); memset(&acc[na], 0, sizeof(acc_t)); ++na; }
        else { free(acc[j].name); free(acc[j].f1); free(acc[j].f2); }
        acc[j].name = strdup(line); acc[j].f1 = strdup(t1 + 1); acc[j].f2 = t2 ? strdup(t2 + 1) : NULL;
        free(line);
    }
    free(c);
    qsort(acc, na, sizeof(acc_t), acc_cmp);                /* accessions.sort(): colour = rank, build.rs:102-113 */
    orc_index *ix = orc_index_new(m, n_hash, k, na);
    ix->m_size = m_size;
    for (size_t a = 0; a < na; ++a) {
        orc_kmers *km;
        int is_fasta = 0;
        if (acc[a].f2) {                                    /* build.rs:54-67 */
            orc_kmers *u = orc_kmers_fq_pe_qual(acc[a].f1, acc[a].f2, (uint32_t)k, quality);
            if (!u) { orc_index_free(ix); return NULL; }
            int64_t t = cutoff == -1 ? orc_auto_cutoff(u) : cutoff;
            km = orc_clean_map(u, (uint64_t)(t < 0 ? 0 : t)); orc_kmers_free(u);
        } else {
            size_t l = strlen(acc[a].f1);
            if (l >= 2 && strcmp(acc[a].f1 + l - 2, "gz") == 0) {  /* build.rs:69-84 */
                orc_kmers *u = orc_kmers_from_fq_qual(acc[a].f1, (uint32_t)k, quality);
                if (!u) { orc_index_free(ix); return NULL; }
                int64_t t = cutoff == -1 ? orc_auto_cutoff(u) : cutoff;
                km = orc_clean_map(u, (uint64_t)(t < 0 ? 0 : t)); orc_kmers_free(u);
            } else {                                        /* build.rs:85-99 */
                orc_strvec v = {0};
                if (orc_read_fasta(acc[a].f1, &v) != 0) { orc_index_free(ix); return NULL; }
                orc_kmers *u = orc_kmers_new((uint32_t)k);
                for (size_t s = 0; s < v.n; ++s) orc_kmerize_vector(u, (const uint8_t *)v.s[s], v.len[s], 1);
                orc_strvec_free(&v);
                if (cutoff == -1) km = u;
                else { km = orc_clean_map(u, (uint64_t)cutoff); orc_kmers_free(u); }
                is_fasta = 1;
            }
        }
        /* ref_kmer.insert(accession, kmers.len()); build_single_mini forgets it for fastq accessions */
        orc_index_set_color(ix, a, acc[a].name, (m_size && !is_fasta) ? 0 : orc_kmers_len(km));
        for (uint64_t e2 = 0; e2 < orc_kmers_len(km); ++e2) orc_index_insert(ix, a, orc_kmers_keys(km) + e2 * k);
        orc_kmers_free(km);
    }
    for (size_t a = 0; a < na; ++a) { free(acc[a].name); free(acc[a].f1); free(acc[a].f2); }
    free(acc);
    return ix;
}

/* bincode 1.x default config: little-endian fixed-width ints, usize -> u64, len-prefixed seqs/strings/maps */
static void w64(FILE *f, uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * i)); fwrite(b, 1, 8, f); }
static void w32(FILE *f, uint32_t v) { uint8_t b[4]; for (int i = 0; i < 4; ++i) b[i] = (uint8_t)(v >> (8 * i)); fwrite(b, 1, 4, f); }

int orc_save_bigsi(const char *path, const orc_index *ix) { /* bigsi.rs:51-57; struct field order bigsi.rs:19-27 */
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    w64(f, ix->bloom_size); w64(f, ix->num_hash); w64(f, ix->k_size);
    if (ix->m_size) w64(f, ix->m_size);                     /* BigsyMapMiniNew.m_size (bigsi.rs:40-49) */
    w64(f, ix->n_colors);                                   /* colors: FnvHashMap<usize,String> */
    for (uint64_t c = 0; c < ix->n_colors; ++c) {
        size_t l = strlen(ix->colors[c]);
        w64(f, c); w64(f, l); fwrite(ix->colors[c], 1, l, f);
    }
    uint64_t nrows = 0;                                     /* map: FnvHashMap<usize,BitVec>; only non-zero rows exist */
    for (uint64_t r = 0; r < ix->bloom_size; ++r) nrows += !row_absent(ix, r);
    w64(f, nrows);
    for (uint64_t r = 0; r < ix->bloom_size; ++r) {
        if (row_absent(ix, r)) continue;
        w64(f, r);
        w64(f, ix->w32);                                    /* BitVec.storage: Vec<u32> */
        for (uint32_t w = 0; w < ix->w32; ++w) w32(f, row_ptr(ix, r)[w]);
        w64(f, ix->n_colors);                               /* BitVec.nbits */
    }
    w64(f, ix->n_colors);                                   /* n_ref_kmers: FnvHashMap<String,usize> */
    for (uint64_t c = 0; c < ix->n_colors; ++c) {
        size_t l = strlen(ix->colors[c]);
        w64(f, l); fwrite(ix->colors[c], 1, l, f); w64(f, ix->n_ref_kmers[c]);
    }
    fclose(f);
    return 0;
}

static int r64(FILE *f, uint64_t *v) {
    uint8_t b[8];
    if (fread(b, 1, 8, f) != 8) return -1;
    *v = 0; for (int i = 7; i >= 0; --i) *v = (*v << 8) | b[i];
    return 0;
}

orc_index *orc_read_bigsi(const char *path) { /* bigsi.rs:59-63; a path ending in ".mxi" is a BigsyMapMiniNew (bigsi.rs:79-83) */
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    uint64_t m, nh, k, nc, msz = 0;
    const size_t pl = strlen(path);
    const int mini = pl >= 4 && strcmp(path + pl - 4, ".mxi") == 0;
    if (r64(f, &m) || r64(f, &nh) || r64(f, &k) || (mini && r64(f, &msz)) || r64(f, &nc)) { fclose(f); return NULL; }
    orc_index *ix = orc_index_new(m, nh, k, nc);
    ix->m_size = msz;
    for (uint64_t i = 0; i < nc; ++i) {
        uint64_t id, l;
        if (r64(f, &id) || r64(f, &l) || id >= nc) goto fail;
        char *s = (char *)malloc(l + 1);
        if (fread(s, 1, l, f) != l) { free(s); goto fail; }
        s[l] = 0; free(ix->colors[id]); ix->colors[id] = s;
    }
    uint64_t nrows;
    if (r64(f, &nrows)) goto fail;
    for (uint64_t i = 0; i < nrows; ++i) {
        uint64_t r, nw, nbits;
        if (r64(f, &r) || r64(f, &nw) || r >= m || nw != ix->w32) goto fail;
        for (uint64_t w = 0; w < nw; ++w) {
            uint8_t b[4];
            if (fread(b, 1, 4, f) != 4) goto fail;
            ix->rows[r * ix->w32 + w] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        }
        if (r64(f, &nbits) || nbits != nc) goto fail;
    }
    uint64_t nref;
    if (r64(f, &nref)) goto fail;
    for (uint64_t i = 0; i < nref; ++i) {
        uint64_t l, v;
        if (r64(f, &l)) goto fail;
        char *s = (char *)malloc(l + 1);
        if (fread(s, 1, l, f) != l) { free(s); goto fail; }
        s[l] = 0;
        if (r64(f, &v)) { free(s); goto fail; }
        for (uint64_t c = 0; c < nc; ++c) if (ix->colors[c] && strcmp(ix->colors[c], s) == 0) ix->n_ref_kmers[c] = v;
        free(s);
    }
    fclose(f);
    return ix;
fail:
    fclose(f); orc_index_free(ix);
    return NULL;
}

/* ------------------------------------------------------------------ search loops */

void orc_search_count(const orc_index *ix, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers,
                      uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    /* batch_search_pe.rs:45-84 (fastq branch) == :125-164 (FASTA branch) */
    const uint64_t C = ix->n_colors, n = ix->num_hash, k = ix->k_size;
    uint32_t *first = (uint32_t *)malloc((ix->w32 ? ix->w32 : 1) * sizeof(uint32_t));
    memset(hits, 0, C * sizeof(uint64_t));
    if (n_unique) memset(n_unique, 0, C * sizeof(uint64_t));
    if (sum_unique_freq) memset(sum_unique_freq, 0, C * sizeof(uint64_t));
    for (uint64_t j = 0; j < n_kmers; ++j) {
        const uint8_t *km = kmers + j * k;
        if (unique_colour) unique_colour[j] = 0xFFFFFFFFu;
        uint64_t got = 0;
        for (uint64_t i = 0; i < n; ++i) {                 /* :47-56 */
            uint64_t bi = bit_index(ix, km, i);
            if (row_absent(ix, bi)) break;
            const uint32_t *row = row_ptr(ix, bi);
            if (got == 0) memcpy(first, row, ix->w32 * sizeof(uint32_t));       /* kmer_slices[0].to_owned() */
            else for (uint32_t w = 0; w < ix->w32; ++w) first[w] &= row[w];     /* intersect, lib.rs:598-600 */
            ++got;
        }
        if (got < n) continue;                              /* :57-58 */
        uint64_t nh = 0, last = 0;
        for (uint64_t c = 0; c < C; ++c)                    /* :65-70 per-bit scan */
            if (bit_get(first, c)) { hits[c] += 1; ++nh; last = c; }  /* :71-74 */
        if (nh == 1) {                                      /* :75-82 */
            if (n_unique) n_unique[last] += 1;
            if (sum_unique_freq) sum_unique_freq[last] += freq ? freq[j] : 1;
            if (unique_colour) unique_colour[j] = (uint32_t)last;
        }
    }
    free(first);
}

/* Faithful-structure variant for the bench's third baseline figure (BASELINE.md §2 mode i): the index as the reference holds
 * it — a hash map row -> bit vector with the FNV-1a hasher (bigsi.rs:19-27, FnvHashMap<usize, BitVec>) — and one heap
 * allocation per k-mer for `kmer_slices[0].to_owned()` (batch_search_pe.rs:60).  Same results as orc_search_count. */
struct orc_sparse { uint64_t cap; uint64_t *keys; uint32_t *vals; };   /* open addressing, linear probing; vals = row number + 1 */
static inline uint64_t fnv1a_u64(uint64_t x) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < 8; ++i) { h ^= (x >> (8 * i)) & 0xff; h *= 0x100000001b3ull; }
    return h;
}
orc_sparse *orc_sparse_build(const orc_index *ix) {
    uint64_t n = 0;
    for (uint64_t r = 0; r < ix->bloom_size; ++r) n += !row_absent(ix, r);
    orc_sparse *m = (orc_sparse *)calloc(1, sizeof *m);
    m->cap = 16;
    while (m->cap < n * 2) m->cap <<= 1;
    m->keys = (uint64_t *)malloc(m->cap * sizeof(uint64_t));
    m->vals = (uint32_t *)calloc(m->cap, sizeof(uint32_t));
    if (!m->keys || !m->vals) { orc_sparse_free(m); return NULL; }
    for (uint64_t r = 0; r < ix->bloom_size; ++r) {
        if (row_absent(ix, r)) continue;
        uint64_t i = fnv1a_u64(r) & (m->cap - 1);
        while (m->vals[i]) i = (i + 1) & (m->cap - 1);
        m->keys[i] = r; m->vals[i] = (uint32_t)(r + 1);
    }
    return m;
}
void orc_sparse_free(orc_sparse *m) { if (m) { free(m->keys); free(m->vals); free(m); } }
static inline const uint32_t *sparse_get(const orc_sparse *m, const orc_index *ix, uint64_t r) {   /* bigsi_map.get(&bit_index) */
    uint64_t i = fnv1a_u64(r) & (m->cap - 1);
    while (m->vals[i]) {
        if (m->keys[i] == r) return row_ptr(ix, r);
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
void orc_search_count_sparse(const orc_sparse *map, const orc_index *ix, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers,
                             uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq) {
    const uint64_t C = ix->n_colors, n = ix->num_hash, k = ix->k_size;
    const size_t wbytes = (ix->w32 ? ix->w32 : 1) * sizeof(uint32_t);
    memset(hits, 0, C * sizeof(uint64_t));
    if (n_unique) memset(n_unique, 0, C * sizeof(uint64_t));
    if (sum_unique_freq) memset(sum_unique_freq, 0, C * sizeof(uint64_t));
    for (uint64_t j = 0; j < n_kmers; ++j) {
        const uint8_t *km = kmers + j * k;
        uint32_t *first = NULL;
        uint64_t got = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t *row = sparse_get(map, ix, bit_index(ix, km, i));
            if (!row) break;
            if (got == 0) { first = (uint32_t *)malloc(wbytes); memcpy(first, row, wbytes); }
            else for (uint32_t w = 0; w < ix->w32; ++w) first[w] &= row[w];
            ++got;
        }
        if (got == n) {
            uint64_t nh = 0, last = 0;
            for (uint64_t c = 0; c < C; ++c)
                if (bit_get(first, c)) { hits[c] += 1; ++nh; last = c; }
            if (nh == 1) {
                if (n_unique) n_unique[last] += 1;
                if (sum_unique_freq) sum_unique_freq[last] += freq ? freq[j] : 1;
            }
        }
        free(first);
    }
}

/* Best-effort CPU variant for the bench's second baseline figure (BASELINE.md §2 mode ii): the same loop, k-mers split
 * over n_threads POSIX threads, per-thread counters summed at the end.  (The reference's `search` is single-threaded.) */
#include <pthread.h>
typedef struct {
    const orc_index *ix; const uint8_t *kmers; const uint64_t *freq; uint64_t n;
    uint64_t *hits, *nu, *sf; uint32_t *uc;
} mt_job;
static void *mt_run(void *arg) {
    mt_job *j = (mt_job *)arg;
    orc_search_count(j->ix, j->kmers, j->freq, j->n, j->hits, j->nu, j->sf, j->uc);
    return NULL;
}
void orc_search_count_mt(const orc_index *ix, const uint8_t *kmers, const uint64_t *freq, uint64_t n_kmers, int n_threads,
                         uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour) {
    const uint64_t C = ix->n_colors;
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job *jobs = (mt_job *)calloc((size_t)n_threads, sizeof(mt_job));
    uint64_t *buf = (uint64_t *)calloc((size_t)n_threads * 3 * C, sizeof(uint64_t));
    for (int t = 0; t < n_threads; ++t) {
        const uint64_t lo = n_kmers * (uint64_t)t / (uint64_t)n_threads, hi = n_kmers * (uint64_t)(t + 1) / (uint64_t)n_threads;
        jobs[t] = (mt_job){ix, kmers + lo * ix->k_size, freq ? freq + lo : NULL, hi - lo, buf + (size_t)t * 3 * C,
                           buf + (size_t)t * 3 * C + C, buf + (size_t)t * 3 * C + 2 * C, unique_colour ? unique_colour + lo : NULL};
        pthread_create(&th[t], NULL, mt_run, &jobs[t]);
    }
    memset(hits, 0, C * sizeof(uint64_t));
    if (n_unique) memset(n_unique, 0, C * sizeof(uint64_t));
    if (sum_unique_freq) memset(sum_unique_freq, 0, C * sizeof(uint64_t));
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        for (uint64_t c = 0; c < C; ++c) {
            hits[c] += jobs[t].hits[c];
            if (n_unique) n_unique[c] += jobs[t].nu[c];
            if (sum_unique_freq) sum_unique_freq[c] += jobs[t].sf[c];
        }
    }
    free(buf); free(jobs); free(th);
}

void orc_search_perfect(const orc_index *ix, const uint8_t *kmers, uint64_t n_kmers, uint32_t *and_words, int *missing) {
    /* perfect_search.rs:25-52 (and :83-110) */
    const uint64_t n = ix->num_hash, k = ix->k_size;
    int have = 0; *missing = 0;
    memset(and_words, 0, ix->w32 * sizeof(uint32_t));
    for (uint64_t j = 0; j < n_kmers; ++j) {
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t bi = bit_index(ix, kmers + j * k, i);
            if (row_absent(ix, bi)) { *missing = 1; break; }   /* :31-33: break; later slices.len() < n*K */
            const uint32_t *row = row_ptr(ix, bi);
            if (!have) { memcpy(and_words, row, ix->w32 * sizeof(uint32_t)); have = 1; }
            else for (uint32_t w = 0; w < ix->w32; ++w) and_words[w] &= row[w];
        }
    }
    if (*missing) memset(and_words, 0, ix->w32 * sizeof(uint32_t)); /* "No perfect hits!": nothing is printed */
}

static void and_rows(const orc_index *ix, const uint8_t *km, uint32_t *first, int *absent) { /* read_id_mt_pe.rs:41-52 */
    *absent = 0;
    for (uint64_t i = 0; i < ix->num_hash; ++i) {
        uint64_t bi = bit_index(ix, km, i);
        if (row_absent(ix, bi)) { *absent = 1; return; }
        const uint32_t *row = row_ptr(ix, bi);
        if (i == 0) memcpy(first, row, ix->w32 * sizeof(uint32_t));
        else for (uint32_t w = 0; w < ix->w32; ++w) first[w] &= row[w];
    }
}

void orc_search_index_classic(const orc_index *ix, const uint8_t *kmers, uint64_t n_kmers, uint64_t *report) {
    /* read_id_mt_pe.rs:66-102 */
    const uint64_t C = ix->n_colors;
    uint32_t *first = (uint32_t *)malloc((ix->w32 ? ix->w32 : 1) * sizeof(uint32_t));
    memset(report, 0, (C + 1) * sizeof(uint64_t));
    for (uint64_t j = 0; j < n_kmers; ++j) {
        int absent;
        and_rows(ix, kmers + j * (ix->m_size ? ix->m_size : ix->k_size), first, &absent);
        if (absent) { report[C] += 1; break; }             /* :86-89 */
        for (uint64_t c = 0; c < C; ++c) if (bit_get(first, c)) report[c] += 1;   /* :91-98 */
    }
    free(first);
}

void orc_search_index(const orc_index *ix, const uint8_t *kmers, uint64_t n_kmers, uint64_t start_sample, uint64_t *report) {
    /* read_id_mt_pe.rs:104-165 */
    const uint64_t C = ix->n_colors;
    uint32_t *first = (uint32_t *)malloc((ix->w32 ? ix->w32 : 1) * sizeof(uint32_t));
    uint8_t *in_set = (uint8_t *)calloc(C ? C : 1, 1);      /* `report` FnvHashSet<usize> */
    memset(report, 0, (C + 1) * sizeof(uint64_t));
    uint64_t counter = 0;
    for (uint64_t j = 0; j < n_kmers; ++j) {
        int absent;
        and_rows(ix, kmers + j * (ix->m_size ? ix->m_size : ix->k_size), first, &absent);
        if (absent) { report[C] += 1; break; }             /* :126-128 / :150-152 */
        if (counter < start_sample) {
            for (uint64_t c = 0; c < C; ++c) if (bit_get(first, c)) { in_set[c] = 1; report[c] += 1; }  /* :130-138 */
        } else {
            for (uint64_t c = 0; c < C; ++c) if (in_set[c] && bit_get(first, c)) report[c] += 1;        /* :154-159 */
        }
        ++counter;
    }
    free(first); free(in_set);
}

void orc_readid_counts(const orc_index *ix, const uint8_t *bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                       uint64_t n_reads, uint64_t d, uint64_t start_sample,
                       uint32_t *report, uint32_t *n_kmers, uint8_t *status) {
    /* read_id_mt_pe.rs:300-331 (the part of parallel_vec before kmer_poll_plus) */
    const uint64_t C = ix->n_colors, k = ix->k_size;
    uint64_t *rep = (uint64_t *)malloc((C + 1) * sizeof(uint64_t));
    for (uint64_t r = 0; r < n_reads; ++r) {
        uint32_t *out = report + r * (C + 1);
        memset(out, 0, (C + 1) * sizeof(uint32_t));
        n_kmers[r] = 0; status[r] = 0;
        uint64_t s0 = read_seq0[r], s1 = read_seq0[r + 1];
        if (s0 == s1 || seq_off[s0 + 1] - seq_off[s0] < k) { status[r] = 1; continue; }  /* :305 too_short (first mate only) */
        orc_kmers *set = orc_kmers_new((uint32_t)(ix->m_size ? ix->m_size : k));
        for (uint64_t s = s0; s < s1; ++s) {
            const uint8_t *l = bases + seq_off[s];
            const size_t len = (size_t)(seq_off[s + 1] - seq_off[s]);
            if (ix->m_size) orc_minimerize_skip_n_set(set, l, len, (size_t)k, (size_t)d);   /* :317-318 (m > 0) */
            else orc_kmerize_skip_n_set(set, l, len, (size_t)d);                            /* :315-316 */
        }
        n_kmers[r] = (uint32_t)orc_kmers_len(set);
        if (start_sample == 0) orc_search_index_classic(ix, orc_kmers_keys(set), orc_kmers_len(set), rep);
        else orc_search_index(ix, orc_kmers_keys(set), orc_kmers_len(set), start_sample, rep);
        for (uint64_t c = 0; c <= C; ++c) out[c] = (uint32_t)rep[c];
        orc_kmers_free(set);
    }
    free(rep);
}

/* The reference runs parallel_vec's per-read closure on a rayon pool (read_id_mt_pe.rs:300, main.rs -t): contiguous slices of the
 * batch on n_threads pthreads; rows of the outputs are per read, so the slices write disjoint memory. */
typedef struct { const orc_index *ix; const uint8_t *bases; const uint64_t *seq_off, *read_seq0; uint64_t n_reads, d, start_sample;
                 uint32_t *report, *n_kmers; uint8_t *status; } readid_job;
static void *readid_run(void *arg) {
    readid_job *j = (readid_job *)arg;
    orc_readid_counts(j->ix, j->bases, j->seq_off, j->read_seq0, j->n_reads, j->d, j->start_sample, j->report, j->n_kmers, j->status);
    return NULL;
}
void orc_readid_counts_mt(const orc_index *ix, const uint8_t *bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                          uint64_t n_reads, uint64_t d, uint64_t start_sample, int n_threads,
                          uint32_t *report, uint32_t *n_kmers, uint8_t *status) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    readid_job *jobs = (readid_job *)calloc((size_t)n_threads, sizeof(readid_job));
    for (int t = 0; t < n_threads; ++t) {
        const uint64_t lo = n_reads * (uint64_t)t / (uint64_t)n_threads, hi = n_reads * (uint64_t)(t + 1) / (uint64_t)n_threads;
        jobs[t] = (readid_job){ix, bases, seq_off, read_seq0 + lo, hi - lo, d, start_sample,
                               report + lo * (ix->n_colors + 1), n_kmers + lo, status + lo};
        pthread_create(&th[t], NULL, readid_run, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(jobs); free(th);
}

double orc_false_prob(double m, double k, double n) { /* read_id_mt_pe.rs:695-698 */
    return pow(1.0 - pow(M_E, -((k * (n + 0.5)) / (m - 1.0))), k);
}

static double binom_mass(uint64_t n, double p, uint64_t x) { /* probability::Binomial::mass — crate unpinned; log-space pmf */
    if (x > n) return 0.0;
    if (p <= 0.0) return x == 0 ? 1.0 : 0.0;
    if (p >= 1.0) return x == n ? 1.0 : 0.0;
    double lc = lgamma((double)n + 1.0) - lgamma((double)x + 1.0) - lgamma((double)(n - x) + 1.0);
    return exp(lc + (double)x * log(p) + (double)(n - x) * log1p(-p));
}

int orc_not_fp_significant(uint64_t observations, double p_false, double fp_correct, uint64_t taxon_hits) {
    /* read_id_mt_pe.rs:168-181 */
    double critical = (double)observations * p_false;
    double mpf = binom_mass(observations, p_false, taxon_hits);
    return ((double)taxon_hits < critical) || (((double)taxon_hits > critical) && (mpf >= fp_correct));
}

int orc_kmer_poll_plus(const orc_index *ix, const uint64_t *report, uint64_t kmer_length, double fp_correct,
                       char *label, size_t label_cap, uint64_t *count, int *accept, uint64_t *n_top) {
    /* read_id_mt_pe.rs:187-251; caller semantics of :332-340 folded in (empty report -> no_hits) */
    const uint64_t C = ix->n_colors;
    uint64_t nent = 0, best = 0;
    for (uint64_t c = 0; c <= C; ++c) nent += report[c] != 0;
    *count = 0; *n_top = 0; *accept = 1;
    if (nent == 0 || (nent == 1 && report[C] != 0)) { snprintf(label, label_cap, "no_hits"); return 0; } /* :197-205 */
    /* significant hits; stable sort by count desc == take max count among survivors, ties ascending id */
    uint8_t *sig = (uint8_t *)calloc(C ? C : 1, 1);
    uint64_t nsig = 0;
    for (uint64_t c = 0; c < C; ++c) {
        if (!report[c]) continue;
        double p = orc_false_prob((double)ix->bloom_size, (double)ix->num_hash, (double)ix->n_ref_kmers[c]);
        if (orc_not_fp_significant(kmer_length, p, fp_correct, report[c])) continue;
        sig[c] = 1; ++nsig;
        if (report[c] > best) best = report[c];
    }
    if (nsig == 0) { snprintf(label, label_cap, "no_significant_hits"); *accept = 0; free(sig); return 0; } /* :216-223 */
    size_t pos = 0; uint64_t ntop = 0;
    label[0] = 0;
    for (uint64_t c = 0; c < C; ++c) {
        if (!sig[c] || report[c] != best) continue;
        int w = snprintf(label + pos, pos < label_cap ? label_cap - pos : 0, "%s%s", ntop ? "," : "", ix->colors[c]);
        if (w > 0) pos += (size_t)w;
        ++ntop;
    }
    *count = best; *n_top = ntop; *accept = ntop == 1;
    free(sig);
    return 0;
}

/* ------------------------------------------------------------------ reports */

void orc_unique_modes(const uint32_t *unique_colour, const uint64_t *freq, uint64_t n_kmers, uint64_t n_colors, uint64_t *modes) {
    /* reports.rs:65-77 per colour; O(C * K) but the oracle is only used on small cases */
    for (uint64_t c = 0; c < n_colors; ++c) {
        uint64_t maxf = 0, any = 0;
        for (uint64_t j = 0; j < n_kmers; ++j) if (unique_colour[j] == c) { any = 1; uint64_t f = freq ? freq[j] : 1; if (f > maxf) maxf = f; }
        modes[c] = 0;
        if (!any) continue;
        uint64_t *h = (uint64_t *)calloc(maxf + 1, sizeof(uint64_t));
        for (uint64_t j = 0; j < n_kmers; ++j) if (unique_colour[j] == c) h[freq ? freq[j] : 1]++;
        uint64_t bestc = 0, bestv = 0;
        for (uint64_t v = 0; v <= maxf; ++v) if (h[v] > bestc) { bestc = h[v]; bestv = v; }
        modes[c] = bestv;
        free(h);
    }
}

size_t orc_generate_report(const orc_index *ix, const char *query, const uint64_t *hits, const uint64_t *n_unique,
                           const uint64_t *sum_unique_freq, const uint64_t *modes, uint64_t num_kmers, double cov,
                           char *buf, size_t cap) { /* reports.rs:8-48 */
    size_t pos = 0;
    for (uint64_t c = 0; c < ix->n_colors; ++c) {
        if (!hits[c]) continue;                             /* only colours present in `report` */
        double mean = 0.0; uint64_t modus = 0, specific = 0;
        if (n_unique[c]) {
            mean = (double)sum_unique_freq[c] / (double)n_unique[c];
            modus = modes[c]; specific = n_unique[c];
        }
        double genome_cov = (double)hits[c] / (double)ix->n_ref_kmers[c];
        if (genome_cov > cov) {
            int w = snprintf(buf ? buf + pos : NULL, (buf && pos < cap) ? cap - pos : 0,
                             "%s\t%llu\t%s\t%.2f\t%.2f\t%llu\t%llu\n", query, (unsigned long long)num_kmers,
                             ix->colors[c], genome_cov, mean, (unsigned long long)modus, (unsigned long long)specific);
            if (w > 0) pos += (size_t)w;
        }
    }
    return pos;
}

size_t orc_generate_report_gene(const orc_index *ix, const char *query, const uint64_t *hits, uint64_t num_kmers,
                                double cov, char *buf, size_t cap) { /* reports.rs:50-62 */
    size_t pos = 0;
    for (uint64_t c = 0; c < ix->n_colors; ++c) {
        if (!hits[c]) continue;
        double gene_match = (double)hits[c] / (double)num_kmers;
        if (gene_match >= cov) {
            int w = snprintf(buf ? buf + pos : NULL, (buf && pos < cap) ? cap - pos : 0,
                             "%s\t%s\t%llu\t%.3f\n", query, ix->colors[c], (unsigned long long)num_kmers, gene_match);
            if (w > 0) pos += (size_t)w;
        }
    }
    return pos;
}
