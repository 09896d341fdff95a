// The reference's workload drivers with the hot loops replaced by C-ABI calls:
//   perfect_search::batch_search / batch_search_mf   (src/perfect_search.rs)  -> cid_search_perfect
//   batch_search_pe::batch_search                    (src/batch_search_pe.rs) -> cid_search_count
//   read_id_mt_pe::per_read_stream_se/_pe, stream_fasta (src/read_id_mt_pe.rs) -> cid_readid_count
// plus the CPU-side tails (src/reports.rs, kmer_poll_plus).  stdout/stderr/file formats follow the reference;
// where the reference iterates a RandomState HashMap, rows come out in ascending colour id.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

#include "colorid_host.hpp"

namespace colorid {

#define CID_TRY(expr)                                                  \
    do {                                                               \
        if ((expr) != CID_OK) die("%s: %s", #expr, cid_last_error()); \
    } while (0)

using Clock = std::chrono::steady_clock;
static long secs_since(Clock::time_point t0) { return (long)std::chrono::duration_cast<std::chrono::seconds>(Clock::now() - t0).count(); }
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
// COLORID_TIMING=1: sub-second phase times on stderr (the reference's own timers print whole seconds)
static bool g_timing = getenv("COLORID_TIMING") != nullptr;
static double g_ms_gpu = 0, g_ms_poll = 0, g_ms_gpu_count = 0, g_ms_write = 0;
static uint64_t g_entries = 0;
static double g_ms_wait[4] = {0, 0, 0, 0};   // parser blocked by a full queue | GPU stage idle | GPU stage blocked by the poll | poll idle

// ---------------------------------------------------------------------------------------------- several GPUs
static cid_group *g_group = nullptr;
static std::vector<cid_index *> g_replicas;   // one handle per rank: replicas of the index, or (g_striped) its colour stripes
static bool g_striped = false;
void set_group(cid_group *group, const std::vector<cid_index *> &replicas) { g_group = group; g_replicas = replicas; g_striped = false; }
void set_stripes(cid_group *group, const std::vector<cid_index *> &stripes) { g_group = group; g_replicas = stripes; g_striped = true; }

// the hot-path calls: one GPU, a group with replicas (query sharded), or a group with colour stripes (index sharded)
static int hot_search_count(cid_ctx *ctx, const Bigsi &b, const uint8_t *kmers, const uint32_t *freq, size_t n, uint64_t *hits, uint64_t *nu,
                            uint64_t *sf, uint32_t *uc) {
    if (g_striped) return cid_group_stripes_search_count(g_group, g_replicas.data(), kmers, freq, n, hits, nu, sf, uc);
    return g_group ? cid_group_search_count(g_group, g_replicas.data(), kmers, freq, n, hits, nu, sf, uc)
                   : cid_search_count(ctx, b.index, kmers, freq, n, hits, nu, sf, uc);
}
static int hot_search_count_set(cid_ctx *ctx, const Bigsi &b, const cid_kmerset *ks, uint64_t *hits, uint64_t *nu, uint64_t *sf, uint32_t *uc) {
    if (g_striped) return cid_group_stripes_search_count_set(g_group, g_replicas.data(), ks, hits, nu, sf, uc);
    return g_group ? cid_group_search_count_set(g_group, g_replicas.data(), ks, hits, nu, sf, uc) : cid_search_count_set(ctx, b.index, ks, hits, nu, sf, uc);
}
static int hot_search_perfect(cid_ctx *ctx, const Bigsi &b, const uint8_t *kmers, size_t n, uint32_t *words, int *missing) {
    if (g_striped) return cid_group_stripes_search_perfect(g_group, g_replicas.data(), kmers, n, words, missing);
    return g_group ? cid_group_search_perfect(g_group, g_replicas.data(), kmers, n, words, missing) : cid_search_perfect(ctx, b.index, kmers, n, words, missing);
}
static int hot_search_perfect_set(cid_ctx *ctx, const Bigsi &b, const cid_kmerset *ks, uint32_t *words, int *missing) {
    if (g_striped) return cid_group_stripes_search_perfect_set(g_group, g_replicas.data(), ks, words, missing);
    return g_group ? cid_group_search_perfect_set(g_group, g_replicas.data(), ks, words, missing) : cid_search_perfect_set(ctx, b.index, ks, words, missing);
}

// ---------------------------------------------------------------------------------------------- reports.rs

double false_prob(double m, double k, double n) { return std::pow(1.0 - std::pow(M_E, -((k * (n + 0.5)) / (m - 1.0))), k); }

// probability::Binomial::mass in log space.  The poll evaluates it ~10 times per read (once per candidate colour), so the pieces that
// repeat are tabulated — ln Gamma(i + 1) for small i, ln p and ln(1 - p) per distinct p — by the very calls the direct formula makes:
// the tabulated form returns the same doubles.
static const std::vector<double> &lgamma_table() {
    static const std::vector<double> t = [] {
        std::vector<double> v(1u << 16);
        int sg;   // lgamma_r: plain lgamma writes the global signgam, and several poll threads run at once
        for (size_t i = 0; i < v.size(); ++i) v[i] = lgamma_r((double)i + 1.0, &sg);
        return v;
    }();
    return t;
}
static inline double lgamma1p_int(uint64_t i) {   // ln Gamma(i + 1)
    const std::vector<double> &t = lgamma_table();
    if (i < t.size()) return t[i];
    int sg;
    return lgamma_r((double)i + 1.0, &sg);
}
static double binomial_mass(uint64_t n, double p, uint64_t x) {
    if (x > n) return 0.0;
    if (p <= 0.0) return x == 0 ? 1.0 : 0.0;
    if (p >= 1.0) return x == n ? 1.0 : 0.0;
    struct Logs { double p = -1.0, lp = 0.0, l1mp = 0.0; };
    static thread_local Logs memo[1024];   // one p per colour (its false-positive rate); direct-mapped on the bits of p
    uint64_t bits;
    memcpy(&bits, &p, 8);
    Logs &m = memo[(bits * 0x9E3779B97F4A7C15ull) >> 54];
    if (m.p != p) { m.p = p; m.lp = std::log(p); m.l1mp = std::log1p(-p); }
    const double lc = lgamma1p_int(n) - lgamma1p_int(x) - lgamma1p_int(n - x);
    return std::exp(lc + (double)x * m.lp + (double)(n - x) * m.l1mp);
}

static bool not_fp_significant(uint64_t observations, double p_false, double fp_correct, uint64_t hits) {  // read_id_mt_pe.rs:168-181
    const double critical = (double)observations * p_false;
    // ((hits < critical) || ((hits > critical) && (mpf >= fp_correct))): the mass only matters above the critical value
    if ((double)hits < critical) return true;
    if (!((double)hits > critical)) return false;
    return binomial_mass(observations, p_false, hits) >= fp_correct;
}

// kmer_poll_plus (read_id_mt_pe.rs:187-251) on a read's sparse report — its non-zero entries in ascending colour id (colour C =
// no_hits_num): which entries are significant (sig[e]) and, among those, the highest count and how many entries hold it
struct Poll { int kind; uint64_t best, n_top; uint32_t first_top; };   // kind: 0 no_hits, 1 no_significant_hits, 2 the top colours
static Poll poll_core(const uint32_t *colours, const uint32_t *counts, size_t n_entries, uint64_t kmer_length, size_t C, const std::vector<double> &fp,
                      double fp_correct, uint8_t *sig) {
    if (n_entries == 0 || (n_entries == 1 && colours[0] == C)) return {0, 0, 0, 0};  // :197-205, :332-340
    uint64_t best = 0, n_sig = 0;
    for (size_t e = 0; e < n_entries; ++e) {
        sig[e] = 0;
        if (colours[e] == C) continue;
        if (not_fp_significant(kmer_length, fp[colours[e]], fp_correct, counts[e])) continue;
        sig[e] = 1;
        ++n_sig;
        best = std::max<uint64_t>(best, counts[e]);
    }
    if (n_sig == 0) return {1, 0, 0, 0};  // :216-223
    uint64_t n_top = 0;
    uint32_t first = 0;
    for (size_t e = 0; e < n_entries; ++e) {
        sig[e] = sig[e] && counts[e] == best;
        if (sig[e] && n_top++ == 0) first = colours[e];
    }
    return {2, best, n_top, first};
}

Classification kmer_poll_plus(const uint32_t *colours, const uint32_t *counts, size_t n_entries, uint64_t kmer_length, const Bigsi &b,
                              const std::vector<double> &fp, double fp_correct) {
    std::vector<uint8_t> sig(n_entries + 1, 0);
    const Poll p = poll_core(colours, counts, n_entries, kmer_length, b.colors.size(), fp, fp_correct, sig.data());
    if (p.kind == 0) return {"no_hits", 0, kmer_length, "accept", 0};
    if (p.kind == 1) return {"no_significant_hits", 0, kmer_length, "reject", 0};
    std::string label;
    for (size_t e = 0; e < n_entries; ++e)
        if (sig[e]) {
            if (!label.empty()) label += ",";
            label += b.colors[colours[e]];
        }
    return {label, p.best, kmer_length, p.n_top == 1 ? "accept" : "reject", p.n_top};
}

// The tally of <prefix>_counts.txt, kept while the rows of <prefix>_reads.txt are written: the reference re-reads the file it
// has just written (reports.rs:98-120); the result is the same unless a read id or an accession name holds a tab (the re-read
// would then split the row differently), in which case the tally is dropped and the file is parsed as the reference does.
static std::map<std::string, uint64_t> g_read_counts;
static bool g_read_counts_valid = false;

void read_counts_five_fields(const std::string &reads_file, const std::string &prefix) {  // reports.rs:98-120
    std::map<std::string, uint64_t> counts;
    std::string line;
    if (g_read_counts_valid) {
        counts.swap(g_read_counts);
        g_read_counts_valid = false;
    } else {
    LineReader r(reads_file);
    while (r.next(line)) {
        std::vector<std::string> v;
        size_t p = 0;
        while (true) {
            size_t e = line.find('\t', p);
            v.push_back(line.substr(p, e == std::string::npos ? std::string::npos : e - p));
            if (e == std::string::npos) break;
            p = e + 1;
        }
        if (v.size() < 5) die("malformed line in %s", reads_file.c_str());
        counts[v[4] == "accept" ? v[1] : std::string("reject")] += 1;
    }
    }
    FILE *f = fopen((prefix + "_counts.txt").c_str(), "w");
    if (!f) die("could not create outfile!");
    for (auto &kv : counts) fprintf(f, "%s\t%llu\n", kv.first.c_str(), (unsigned long long)kv.second);
    fclose(f);
}

// reports.rs:8-48: hits / n_ref_kmers > cov -> query, K, accession, cov, mean, mode, n_unique
// `modes`: the per-colour mode of the unique-hit k-mer frequencies when the device computed it (cid_search_count_set_report);
// NULL: derived here from the per-k-mer unique colours and multiplicities (reports.rs:65-77; ties -> smallest value)
static void generate_report(const std::string &query, const Bigsi &b, const std::vector<uint64_t> &hits,
                            const std::vector<uint64_t> &n_unique, const std::vector<uint64_t> &sum_freq,
                            const std::vector<uint32_t> &unique_colour, const uint32_t *counts, size_t n_kmers, double cov,
                            const std::vector<uint64_t> *modes = nullptr) {
    const size_t C = b.colors.size();
    std::vector<std::map<uint32_t, uint64_t>> occ(modes ? 0 : C);
    if (!modes)
        for (size_t j = 0; j < n_kmers; ++j)
            if (unique_colour[j] != CID_NOT_UNIQUE) occ[unique_colour[j]][counts[j]] += 1;
    for (size_t c = 0; c < C; ++c) {
        if (!hits[c]) continue;
        double mean = 0.0;
        uint64_t modus = 0, specific = 0;
        if (n_unique[c]) {
            mean = (double)sum_freq[c] / (double)n_unique[c];
            if (modes) modus = (*modes)[c];
            else {
                uint64_t bestc = 0;
                for (auto &kv : occ[c])
                    if (kv.second > bestc) { bestc = kv.second; modus = kv.first; }
            }
            specific = n_unique[c];
        }
        const double genome_cov = (double)hits[c] / (double)b.n_ref_kmers[c];
        if (genome_cov > cov)
            printf("%s\t%zu\t%s\t%.2f\t%.2f\t%llu\t%llu\n", query.c_str(), n_kmers, b.colors[c].c_str(), genome_cov, mean,
                   (unsigned long long)modus, (unsigned long long)specific);
    }
}

static void generate_report_gene(const std::string &query, const Bigsi &b, const std::vector<uint64_t> &hits, size_t num_kmers,
                                 double cov) {  // reports.rs:50-62
    for (size_t c = 0; c < b.colors.size(); ++c) {
        if (!hits[c]) continue;
        const double gene_match = (double)hits[c] / (double)num_kmers;
        if (gene_match >= cov) printf("%s\t%s\t%zu\t%.3f\n", query.c_str(), b.colors[c].c_str(), num_kmers, gene_match);
    }
}

// ---------------------------------------------------------------------------------------------- perfect_search.rs

// ---------------------------------------------------------------------------------------------- GPU k-mer counting
//  the k-mer map is built and kept on the device (2-bit codes for k <= 32, byte strings beyond);
// COLORID_HOST_KMERS=1 forces the host map.

bool gpu_counting_enabled(uint64_t k) { return k <= 128 && !getenv("COLORID_HOST_KMERS"); }
static bool gpu_counting(const Bigsi &b) { return gpu_counting_enabled(b.k_size); }

// qual_mask (seq.rs:36-56) applied while the read is appended to a batch: q == 0 keeps the sequence as it is; otherwise the output
// has one base per quality character, 'N' where the quality is below q + 33
static void append_masked(std::vector<uint8_t> &bases, const char *seq, size_t slen, const char *qual, size_t qlen, uint8_t q) {
    const size_t at = bases.size();
    if (q == 0) { bases.insert(bases.end(), seq, seq + slen); return; }
    if (slen < qlen) die("ERROR: could not get the next nt in the sequence");
    bases.resize(at + qlen);
    uint8_t *o = bases.data() + at;
    const uint8_t max_quality = (uint8_t)(q + 33);
    for (size_t i = 0; i < qlen; ++i) o[i] = (uint8_t)qual[i] < max_quality ? (uint8_t)'N' : (uint8_t)seq[i];
}

struct SeqBatch {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off{0};
    void push(const std::string &s) { bases.insert(bases.end(), s.begin(), s.end()); off.push_back(bases.size()); }
    void push_masked(const std::string &s, const char *qual, size_t qlen, uint8_t q) { append_masked(bases, s.data(), s.size(), qual, qlen, q); off.push_back(bases.size()); }
    size_t n() const { return off.size() - 1; }
    void clear() { bases.clear(); off.assign(1, 0); }
};

namespace {

struct ReadBatch {  // Vec<(String, Vec<String>)> packed for cid_readid_count
    std::string id_chars;                 // the ids, NUL-terminated, back to back (no allocation per read)
    std::vector<uint64_t> id_off;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> seq_off{0};
    std::vector<uint64_t> read_seq0{0};
    void push(const std::string &id, const std::string *seqs, size_t n) {
        begin(id);
        for (size_t s = 0; s < n; ++s) {
            bases.insert(bases.end(), seqs[s].begin(), seqs[s].end());
            seq_off.push_back(bases.size());
        }
        end();
    }
    // the same in pieces: begin(id), one mate(...) per sequence (quality-masked while it is copied), end()
    void begin(const char *id, size_t n) { id_off.push_back(id_chars.size()); id_chars.append(id, n); id_chars.push_back('\0'); }
    void begin(const std::string &id) { begin(id.data(), id.size()); }
    void mate(const char *seq, size_t slen, const char *qual, size_t qlen, uint8_t q) { append_masked(bases, seq, slen, qual, qlen, q); seq_off.push_back(bases.size()); }
    // the reads of `o` after this batch's own (pieces parsed on other threads, in input order)
    void append(const ReadBatch &o) {
        const uint64_t id0 = id_chars.size(), b0 = bases.size(), s0 = seq_off.size() - 1;
        id_chars.append(o.id_chars);
        for (uint64_t v : o.id_off) id_off.push_back(id0 + v);
        bases.insert(bases.end(), o.bases.begin(), o.bases.end());
        for (size_t i = 1; i < o.seq_off.size(); ++i) seq_off.push_back(b0 + o.seq_off[i]);
        for (size_t i = 1; i < o.read_seq0.size(); ++i) read_seq0.push_back(s0 + o.read_seq0[i]);
    }
    void end() { read_seq0.push_back(seq_off.size() - 1); }
    const char *id(size_t r) const { return id_chars.data() + id_off[r]; }
    size_t size() const { return id_off.size(); }
    // long reads: a batch also closes once it holds this many bases (the library numbers a batch's k-mer windows in 32 bits;
    // batch boundaries never change a read's result)
    bool heavy() const { return bases.size() >= (256u << 20); }
    void clear() { id_chars.clear(); id_off.clear(); bases.clear(); seq_off.assign(1, 0); read_seq0.assign(1, 0); }
};

// ---- FASTQ text -> packed batches on several threads.  RecordChunker cuts each input's decoded blocks at record boundaries (a
// newline scan on the calling thread); the records of a chunk — for pairs: as many records of either file's current chunk as
// both have — are split into lines, quality-masked (seq.rs:36-56) and packed by COLORID_PARSE_THREADS threads; the
// pieces reach `sink` in input order.  The same reads in the same order as the line loops of read_id_mt_pe.rs:862-895 / :927-975
// and kmer.rs:481-503 / :619-647: a record is pushed at its fourth line; for pairs the walk ends with the shorter file.
// (default: a fifth of cpu_budget(), at most 4 — 3 on a 16-CPU share of a GPU box)
const int g_parse_threads = [] { const char *e = getenv("COLORID_PARSE_THREADS"); const int v = e ? atoi(e) : std::min(4, cpu_budget() / 5); return v < 1 ? 1 : v; }();

struct Line { const char *p; size_t n; };
inline void record_lines(const RecChunk &c, size_t r, Line out[4]) {
    const char *p = c.buf.data() + c.rec_begin(r), *end = c.buf.data() + c.rec_end[r];
    for (int i = 0; i < 4; ++i) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));   // there: the chunker counted four of them
        size_t n = (size_t)(nl - p);
        if (n && p[n - 1] == '\r') --n;
        out[i] = Line{p, n};
        p = nl + 1;
    }
}
ReadBatch pack_records(ReadBatch rb, const RecChunk *c1, size_t a0, const RecChunk *c2, size_t b0, size_t n, uint8_t q, bool want_ids) {
    rb.clear();   // (a recycled batch keeps its buffers: no fresh pages to fault in)
    const size_t text = c1->rec_end[a0 + n - 1] - c1->rec_begin(a0);
    rb.bases.reserve((c2 ? 2 : 1) * (text / 2 + 64));
    rb.seq_off.reserve((c2 ? 2 : 1) * n + 1);
    rb.read_seq0.reserve(n + 1);
    if (want_ids) { rb.id_off.reserve(n); rb.id_chars.reserve(text / 4 + 64); }
    Line l[4], m[4];
    for (size_t i = 0; i < n; ++i) {
        record_lines(*c1, a0 + i, l);
        if (want_ids) rb.begin(l[0].p, l[0].n); else rb.id_off.push_back(0);
        rb.mate(l[1].p, l[1].n, l[3].p, l[3].n, q);
        if (c2) {
            record_lines(*c2, b0 + i, m);
            rb.mate(m[1].p, m[1].n, m[3].p, m[3].n, q);
        }
        rb.end();
    }
    return rb;
}

inline ReadBatch no_spare() { return ReadBatch(); }
template <typename Sink, typename Spare = ReadBatch (*)()>
void stream_fastq_records(const std::string &f1, const std::string *f2, uint8_t q, bool want_ids, Sink &&sink, Spare &&spare = no_spare) {
    LineReader r1(f1);
    std::unique_ptr<LineReader> r2(f2 ? new LineReader(*f2) : nullptr);
    RecordChunker k1(r1);
    std::unique_ptr<RecordChunker> k2(r2 ? new RecordChunker(*r2) : nullptr);
    auto fresh = [](RecordChunker &k) {   // a chunk whose buffer returns to its reader when the last piece cut from it is packed
        RecordChunker *kp = &k;
        return std::shared_ptr<RecChunk>(new RecChunk, [kp](RecChunk *c) { kp->recycle(*c); delete c; });
    };
    TaskPool pool(g_parse_threads);   // the packers
    std::deque<std::future<ReadBatch>> inflight;
    auto drain_one = [&] { ReadBatch piece = inflight.front().get(); inflight.pop_front(); sink(std::move(piece)); };
    std::shared_ptr<RecChunk> c1, c2;
    size_t p1 = 0, p2 = 0;
    for (;;) {
        if (!c1 || p1 == c1->records()) { c1 = fresh(k1); p1 = 0; if (!k1.next(*c1)) break; }
        size_t n = c1->records() - p1;
        if (k2) {
            if (!c2 || p2 == c2->records()) { c2 = fresh(*k2); p2 = 0; if (!k2->next(*c2)) break; }
            n = std::min(n, c2->records() - p2);
        }
        while (inflight.size() >= (size_t)g_parse_threads) drain_one();
        std::shared_ptr<ReadBatch> buf(new ReadBatch(spare()));   // a batch whose buffers an earlier round already grew (or an empty one)
        auto task = std::make_shared<std::packaged_task<ReadBatch()>>([buf, c1, p1, c2, p2, n, q, want_ids] { return pack_records(std::move(*buf), c1.get(), p1, c2.get(), p2, n, q, want_ids); });
        inflight.push_back(task->get_future());
        pool.submit([task] { (*task)(); });
        p1 += n; p2 += n;
    }
    while (!inflight.empty()) drain_one();
}

}  // namespace

// host-only helper of the CPU tests: what the record pipeline hands to the GPU calls, one line per read — id, then its masked mates
void debug_records(const std::string &f1, const std::string *f2, uint8_t q) {
    stream_fastq_records(f1, f2, q, true, [&](ReadBatch &&piece) {
        for (size_t r = 0; r < piece.size(); ++r) {
            fputs(piece.id(r), stdout);
            for (uint64_t sq = piece.read_seq0[r]; sq < piece.read_seq0[r + 1]; ++sq) {
                fputc('\t', stdout);
                fwrite(piece.bases.data() + piece.seq_off[sq], 1, (size_t)(piece.seq_off[sq + 1] - piece.seq_off[sq]), stdout);
            }
            fputc('\n', stdout);
        }
    });
}

cid_kmerset *count_fasta_gpu(cid_ctx *ctx, uint64_t k, const std::vector<std::string> &seqs) {
    cid_kmerset *ks = nullptr;
    CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
    SeqBatch sb;
    for (const std::string &s : seqs) sb.push(s);
    CID_TRY(cid_kmerset_add_seqs(ks, sb.bases.data(), sb.off.data(), sb.n(), 0));
    CID_TRY(cid_kmerset_finalize(ks, nullptr));
    return ks;
}

// fastq(.gz) SE or PE (kmer.rs:461-510 / :581-655) in batches of 256 MB of bases: add(batch) returns the add_seqs code;
// false = a batch held lower-case bases (CID_ERR_UNSUPPORTED): count the file on the host
template <typename Add>
static bool stream_fastq_batches(const std::string &f1, const std::string *f2, uint8_t q, Add &&add) {
    SeqBatch sb;
    bool ok = true;
    auto flush = [&]() {
        if (sb.n() == 0 || !ok) return;
        const int rc = add(sb);
        if (rc == CID_ERR_UNSUPPORTED) ok = false;
        else if (rc != CID_OK) die("cid_kmerset_add_seqs: %s", cid_last_error());
        sb.clear();
    };
    stream_fastq_records(f1, f2, q, false, [&](ReadBatch &&piece) {
        if (!ok) return;   // (the rest of the file is still read: the caller falls back to the host map, which reads it again)
        const uint64_t b0 = sb.bases.size();
        sb.bases.insert(sb.bases.end(), piece.bases.begin(), piece.bases.end());
        for (size_t i = 1; i < piece.seq_off.size(); ++i) sb.off.push_back(b0 + piece.seq_off[i]);
        if (sb.bases.size() >= (256u << 20)) flush();
    });
    flush();
    return ok;
}

// Block-gzip input through the device FASTQ front end: the members go up compressed (the reader's spare threads inflate their share),
// records are cut, masked and packed on the device and their k-mers go straight into the set — no text comes back.  false: the input
// is not for this path (CID_ERR_UNSUPPORTED: lower-case bases, reads longer than a segment) and `ks` holds a partial count.
static bool count_bgzf_on_device(cid_ctx *ctx, cid_kmerset *ks, const std::vector<std::string> &fq, uint8_t q) {
    const size_t n_files = fq.size();
    cid_fastq *fr = nullptr;
    CID_TRY(cid_fastq_create(ctx, (int)n_files, q, &fr));
    const size_t target = read_id_mt_pe::device_fastq_stretch_bytes(0);
    std::unique_ptr<BgzfMemberReader> rd[2];
    for (size_t i = 0; i < n_files; ++i)
        rd[i] = BgzfMemberReader::open(fq[i], target, read_id_mt_pe::device_fastq_host_share(), read_id_mt_pe::device_fastq_host_threads(n_files));
    BgzfStretch st[2][2];   // per file two stretches in turn: the one pushed last stays untouched while its text is still on the bus
    size_t turn[2] = {0, 0}, pending[2] = {0, 0};
    bool more[2] = {true, n_files == 2};
    double ms_read = 0, ms_push = 0, ms_count = 0;
    auto push_next = [&](size_t i) {
        if (!more[i]) return;
        const auto tr = Clock::now();
        BgzfStretch &sx = st[i][turn[i]++ & 1];
        const bool got = rd[i]->next(sx);
        ms_read += ms_since(tr);
        if (!got) { more[i] = false; return; }
        const auto tp = Clock::now();
        const bool host_part = sx.host_text_bytes > 0;
        CID_TRY(cid_fastq_push_bgzf(fr, (int)i, sx.bytes.data(), sx.bytes.size(), sx.off.data(), sx.len.data(), sx.text_len.data(), sx.device_members,
                                    sx.last && !host_part ? CID_FASTQ_LAST : 0));
        if (host_part)
            CID_TRY(cid_fastq_push_text(fr, (int)i, sx.host_text.p, sx.host_text_bytes, (sx.last ? CID_FASTQ_LAST : 0) | (sx.host_text.pinned ? CID_FASTQ_KEEP : 0)));
        if (sx.last) more[i] = false;
        ms_push += ms_since(tp);
        ++pending[i];
    };
    for (size_t i = 0; i < n_files; ++i) push_next(i);
    bool ok = true;
    while (ok && (pending[0] || pending[1])) {
        for (size_t i = 0; i < n_files; ++i) push_next(i);   // the stretch after this one: inflated while this one is counted
        uint64_t n = 0;
        const auto tc = Clock::now();
        const int rc = cid_fastq_count_kmers(fr, ks, 2, &n);
        ms_count += ms_since(tc);
        for (size_t i = 0; i < n_files; ++i) if (pending[i]) --pending[i];
        if (rc == CID_ERR_UNSUPPORTED) ok = false;
        else if (rc != CID_OK) die("%s", cid_last_error());
    }
    cid_fastq_destroy(fr);
    if (g_timing) fprintf(stderr, "timing: device front end: waiting for the file reader %.0f ms, push (H2D of the members) %.0f ms, counting %.0f ms\n", ms_read, ms_push, ms_count);
    return ok;
}

// nullptr = the file holds lower-case bases: count it on the host
cid_kmerset *count_fastq_gpu(cid_ctx *ctx, uint64_t k, const std::string &f1, const std::string *f2, uint8_t q) {
    cid_kmerset *ks = nullptr;
    CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
    {
        std::vector<std::string> fq{f1};
        if (f2) fq.push_back(*f2);
        if (k <= 32 && read_id_mt_pe::device_fastq_wanted(fq, fq.size())) {
            const auto t_dev = Clock::now();
            if (count_bgzf_on_device(ctx, ks, fq, q)) {
                CID_TRY(cid_kmerset_finalize(ks, nullptr));
                if (g_timing) fprintf(stderr, "timing: query k-mers counted through the device front end in %.0f ms\n", ms_since(t_dev));
                return ks;
            }
            cid_kmerset_destroy(ks);   // (a partial count) — the host reads the files again
            CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
        }
    }
    const bool ok = stream_fastq_batches(f1, f2, q, [&](const SeqBatch &sb) { return cid_kmerset_add_seqs(ks, sb.bases.data(), sb.off.data(), sb.n(), 1); });
    if (!ok) { cid_kmerset_destroy(ks); return nullptr; }
    CID_TRY(cid_kmerset_finalize(ks, nullptr));
    return ks;
}

int64_t auto_cutoff_gpu(cid_kmerset *ks) {
    size_t nb = 0;
    CID_TRY(cid_kmerset_count_histogram(ks, nullptr, nullptr, 0, &nb));
    std::vector<uint32_t> mult(nb);
    std::vector<uint64_t> cnt(nb);
    CID_TRY(cid_kmerset_count_histogram(ks, mult.data(), cnt.data(), nb, &nb));
    std::map<uint64_t, uint64_t> hm;
    for (size_t i = 0; i < nb; ++i) hm[mult[i]] = cnt[i];
    uint64_t n = 0;
    CID_TRY(cid_kmerset_size(ks, &n));
    return auto_cutoff_from_histogram(hm, n);
}

// The query's k-mer map on the GPU side of the search drivers: one cid_kmerset (one GPU, or a striped index whose ranks all need
// the whole set), or — a replicated group, k <= 32 — a cid_group_kmerset counted over all ranks, each of which then searches its
// own range of the set (COLORID_ONE_GPU_KMERS=1 keeps the counting on rank 0).
struct GpuSet {
    cid_kmerset *one = nullptr;
    cid_group_kmerset *many = nullptr;
    explicit operator bool() const { return one || many; }
    uint64_t size() const {
        uint64_t n = 0;
        if (many) CID_TRY(cid_group_kmerset_size(many, &n)); else CID_TRY(cid_kmerset_size(one, &n));
        return n;
    }
    void clean(uint64_t t) { if (many) CID_TRY(cid_group_kmerset_clean(many, t)); else CID_TRY(cid_kmerset_clean(one, t)); }
    void counts(uint32_t *out) const { if (many) CID_TRY(cid_group_kmerset_download(many, nullptr, out)); else CID_TRY(cid_kmerset_download(one, nullptr, out)); }
    int64_t auto_cutoff() const {
        if (!many) return auto_cutoff_gpu(one);
        size_t nb = 0;
        CID_TRY(cid_group_kmerset_count_histogram(many, nullptr, nullptr, 0, &nb));
        std::vector<uint32_t> mult(nb);
        std::vector<uint64_t> cnt(nb);
        CID_TRY(cid_group_kmerset_count_histogram(many, mult.data(), cnt.data(), nb, &nb));
        std::map<uint64_t, uint64_t> hm;
        for (size_t i = 0; i < nb; ++i) hm[mult[i]] = cnt[i];
        return auto_cutoff_from_histogram(hm, size());
    }
    void destroy() { if (many) cid_group_kmerset_destroy(many); if (one) cid_kmerset_destroy(one); many = nullptr; one = nullptr; }
};
static bool count_over_group(uint64_t k) { return g_group && !g_striped && k <= 32 && !getenv("COLORID_ONE_GPU_KMERS"); }

static GpuSet count_fasta_set(cid_ctx *ctx, uint64_t k, const std::vector<std::string> &seqs) {
    GpuSet gs;
    if (!count_over_group(k)) { gs.one = count_fasta_gpu(ctx, k, seqs); return gs; }
    CID_TRY(cid_group_kmerset_create(g_group, (uint32_t)k, &gs.many));
    SeqBatch sb;
    for (const std::string &s : seqs) sb.push(s);
    CID_TRY(cid_group_kmerset_add_seqs(gs.many, sb.bases.data(), sb.off.data(), sb.n(), 0));
    CID_TRY(cid_group_kmerset_finalize(gs.many, nullptr));
    return gs;
}
static GpuSet count_fastq_set(cid_ctx *ctx, uint64_t k, const std::string &f1, const std::string *f2, uint8_t q) {
    GpuSet gs;
    if (!count_over_group(k)) { gs.one = count_fastq_gpu(ctx, k, f1, f2, q); return gs; }
    CID_TRY(cid_group_kmerset_create(g_group, (uint32_t)k, &gs.many));
    const bool ok = stream_fastq_batches(f1, f2, q, [&](const SeqBatch &sb) { return cid_group_kmerset_add_seqs(gs.many, sb.bases.data(), sb.off.data(), sb.n(), 1); });
    if (!ok) { gs.destroy(); return gs; }
    CID_TRY(cid_group_kmerset_finalize(gs.many, nullptr));
    return gs;
}
static int hot_search_count_gpuset(cid_ctx *ctx, const Bigsi &b, const GpuSet &gs, uint64_t *hits, uint64_t *nu, uint64_t *sf, uint32_t *uc) {
    if (gs.many) return cid_group_search_count_parts(g_group, g_replicas.data(), gs.many, hits, nu, sf, uc);
    return hot_search_count_set(ctx, b, gs.one, hits, nu, sf, uc);
}
static int hot_search_perfect_gpuset(cid_ctx *ctx, const Bigsi &b, const GpuSet &gs, uint32_t *words, int *missing) {
    if (gs.many) return cid_group_search_perfect_parts(g_group, g_replicas.data(), gs.many, words, missing);
    return hot_search_perfect_set(ctx, b, gs.one, words, missing);
}

static void print_perfect(const Bigsi &b, const std::string &label, size_t n_kmers, const std::vector<uint32_t> &words, int missing);

static void perfect_one(cid_ctx *ctx, const Bigsi &b, const std::string &label, const KmerMap &km) {
    const uint32_t w32 = (uint32_t)((b.colors.size() + 31) / 32);
    std::vector<uint32_t> words(w32);
    int missing = 0;
    CID_TRY(hot_search_perfect(ctx, b, km.keys(), km.size(), words.data(), &missing));
    print_perfect(b, label, km.size(), words, missing);
}

static void print_perfect(const Bigsi &b, const std::string &label, size_t n_kmers, const std::vector<uint32_t> &words, int missing) {
    if (missing) {
        fprintf(stderr, "No perfect hits!\n");
        return;
    }
    size_t n_hits = 0;
    for (size_t c = 0; c < b.colors.size(); ++c) n_hits += (words[c / 32] >> (c % 32)) & 1u;
    fprintf(stderr, "%zu hits\n", n_hits);
    for (size_t c = 0; c < b.colors.size(); ++c)
        if ((words[c / 32] >> (c % 32)) & 1u) printf("%s\t%s\t%zu\t1.00\n", label.c_str(), b.colors[c].c_str(), n_kmers);
}

void perfect_search::batch_search(cid_ctx *ctx, const std::vector<std::string> &files, const Bigsi &b) {
    for (const std::string &file : files) {
        fprintf(stderr, "Counting k-mers, this may take a while!\n");
        if (gpu_counting(b)) {
            GpuSet ks = count_fasta_set(ctx, b.k_size, read_fasta(file));
            const uint64_t n = ks.size();
            fprintf(stderr, "%llu kmers in query\n", (unsigned long long)n);
            if (n == 0) {
                fprintf(stderr, "Warning! no kmers in query; maybe your kmer length is larger than your query length?\n");
            } else {
                std::vector<uint32_t> words((b.colors.size() + 31) / 32);
                int missing = 0;
                CID_TRY(hot_search_perfect_gpuset(ctx, b, ks, words.data(), &missing));
                print_perfect(b, file, n, words, missing);
            }
            ks.destroy();
            continue;
        }
        KmerMap km((uint32_t)b.k_size);
        kmerize_vector(read_fasta(file), 1, km);
        fprintf(stderr, "%zu kmers in query\n", km.size());
        if (km.size() == 0) {
            fprintf(stderr, "Warning! no kmers in query; maybe your kmer length is larger than your query length?\n");
            continue;
        }
        perfect_one(ctx, b, file, km);
    }
}

void perfect_search::batch_search_mf(cid_ctx *ctx, const std::vector<std::string> &files, const Bigsi &b) {
    for (const std::string &file : files) {
        std::vector<std::string> labels, seqs;
        read_fasta_mf(file, labels, seqs);
        for (size_t i = 0; i < labels.size(); ++i) {
            if (i >= seqs.size()) die("index out of bounds: the len is %zu but the index is %zu", seqs.size(), i);  // sequences[i]
            KmerMap km((uint32_t)b.k_size);
            if (!kmerize_string(seqs[i], km)) {
                printf("Warning! no kmers in query '%s'; maybe your kmer length is larger than your query length?\n", labels[i].c_str());
                continue;
            }
            fprintf(stderr, "%zu kmers in query\n", km.size());
            perfect_one(ctx, b, labels[i], km);
        }
    }
}

// ---------------------------------------------------------------------------------------------- batch_search_pe.rs

void batch_search_pe::batch_search(cid_ctx *ctx, const std::vector<std::string> &files1, const std::vector<std::string> &files2,
                                   const Bigsi &b, int64_t filter, double cov, bool gene_search, uint8_t qual_offset) {
    const size_t C = b.colors.size();
    for (size_t i = 0; i < files1.size(); ++i) {
        const std::string &file1 = files1[i];
        const bool gz = file1.size() >= 2 && file1.compare(file1.size() - 2, 2, "gz") == 0;
        if (gz && !files2.empty() && i >= files2.size()) die("index out of bounds: the len is %zu but the index is %zu", files2.size(), i);
        if (gz && files2.empty()) fprintf(stderr, "%s\nCounting k-mers, this may take a while!\n", file1.c_str());
        else if (gz) fprintf(stderr, "Paired end: %s %s\nCounting k-mers, this may take a while!\n", file1.c_str(), files2[i].c_str());
        else fprintf(stderr, "%s\nCounting k-mers, this may take a while!\n", file1.c_str());  // anything else is FASTA (:106-123)
        // which cutoff applies (batch_search_pe.rs:34-39 fastq, :111-121 FASTA): -1 = auto_cutoff
        const bool fasta_gene = !gz && gene_search;
        const bool use_auto = !fasta_gene && filter < 0;
        if (!gz && !gene_search && filter < 0) fprintf(stderr, "no gene search\n");

        std::vector<uint64_t> hits(C), n_unique(C), sum_freq(C), modes;
        bool have_modes = false;
        std::vector<uint32_t> uc, counts;
        size_t n_kmers = 0;
        GpuSet ks;
        if (gpu_counting(b)) {
            ks = gz ? count_fastq_set(ctx, b.k_size, file1, files2.empty() ? nullptr : &files2[i], qual_offset)
                    : count_fasta_set(ctx, b.k_size, read_fasta(file1));
        }
        if (ks) {  // the k-mer map lives on the device(s)
            uint64_t t = fasta_gene ? 0 : (uint64_t)(filter < 0 ? 0 : filter);
            if (use_auto) { const int64_t a = ks.auto_cutoff(); if (a < 0) die("auto_cutoff: histogram too short"); t = (uint64_t)a; }
            ks.clean(t);
            n_kmers = ks.size();
            fprintf(stderr, "%zu k-mers in query\n", n_kmers);
            const auto t0 = Clock::now();
            if (!gene_search && !g_group) {   // the whole report on the device: nothing per k-mer comes back
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_search_count_set_report(ctx, b.index, ks.one, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else if (!gene_search && g_striped && ks.one) {   // the same over colour stripes: the report is finished on rank 0's device
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_group_stripes_search_count_set_report(g_group, g_replicas.data(), ks.one, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else if (!gene_search && ks.many) {   // the same over the ranks' parts: the (colour, multiplicity) histograms add up
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_group_search_count_parts_report(g_group, g_replicas.data(), ks.many, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else {
                if (!gene_search) { uc.resize(n_kmers); counts.resize(n_kmers); }
                CID_TRY(hot_search_count_gpuset(ctx, b, ks, hits.data(), gene_search ? nullptr : n_unique.data(),
                                                gene_search ? nullptr : sum_freq.data(), gene_search ? nullptr : uc.data()));
                if (!gene_search) ks.counts(counts.data());
            }
            if (gz) fprintf(stderr, "Search: %ld sec\n", secs_since(t0));
            ks.destroy();
        } else {  // host k-mer map (k > 32, lower-case fastq, or COLORID_HOST_KMERS)
            fprintf(stderr, "k-mer map on the host\n");
            KmerMap km((uint32_t)b.k_size);
            if (gz && files2.empty()) kmers_from_fq_qual(file1, qual_offset, km);
            else if (gz) kmers_fq_pe_qual(file1, files2[i], qual_offset, km);
            else kmerize_vector(read_fasta(file1), 1, km);
            uint64_t t = fasta_gene ? 0 : (uint64_t)(filter < 0 ? 0 : filter);
            if (use_auto) { const int64_t a = km.auto_cutoff(); if (a < 0) die("auto_cutoff: histogram too short"); t = (uint64_t)a; }
            km.clean(t);
            n_kmers = km.size();
            fprintf(stderr, "%zu k-mers in query\n", n_kmers);
            const auto t0 = Clock::now();
            if (!gene_search) uc.resize(n_kmers);
            CID_TRY(hot_search_count(ctx, b, km.keys(), km.counts().data(), km.size(), hits.data(),
                                     gene_search ? nullptr : n_unique.data(), gene_search ? nullptr : sum_freq.data(),
                                     gene_search ? nullptr : uc.data()));
            if (gz) fprintf(stderr, "Search: %ld sec\n", secs_since(t0));
            counts = km.counts();
        }
        if (!gene_search) generate_report(file1, b, hits, n_unique, sum_freq, uc, counts.data(), n_kmers, cov, have_modes ? &modes : nullptr);
        else generate_report_gene(file1, b, hits, n_kmers, cov);
    }
}

// ---------------------------------------------------------------------------------------------- read_id_mt_pe.rs

namespace {

// parallel_vec (read_id_mt_pe.rs:282-363) in two stages: counts on the GPU ...
struct Counted {   // one batch after the GPU stage: each read's non-zero (colour, count) entries
    ReadBatch rb;
    std::vector<uint32_t> nk;
    std::vector<uint8_t> status;
    std::vector<uint64_t> row_start;
    std::vector<uint32_t> colours, counts;
};
void count_batch(cid_ctx *ctx, const Bigsi &b, Counted &c, size_t d, size_t start_sample) {
    ReadBatch &rb = c.rb;
    const size_t n = rb.size();
    // counts stay on the GPU as dense rows; only each read's non-zero (colour, count) entries come back
    const auto t_gpu = Clock::now();
    c.nk.resize(n);
    c.status.resize(n);
    uint64_t n_entries = 0;
    if (g_striped)
        CID_TRY(cid_group_stripes_readid_count_sparse(g_group, g_replicas.data(), rb.bases.data(), rb.seq_off.data(), rb.seq_off.size() - 1,
                                                      rb.read_seq0.data(), n, (uint32_t)d, (uint32_t)start_sample, c.nk.data(), c.status.data(), &n_entries));
    else if (g_group)
        CID_TRY(cid_group_readid_count_sparse(g_group, g_replicas.data(), rb.bases.data(), rb.seq_off.data(), rb.seq_off.size() - 1,
                                              rb.read_seq0.data(), n, (uint32_t)d, (uint32_t)start_sample, c.nk.data(), c.status.data(), &n_entries));
    else
        CID_TRY(cid_readid_count_sparse(ctx, b.index, rb.bases.data(), rb.seq_off.data(), rb.seq_off.size() - 1, rb.read_seq0.data(), n,
                                        (uint32_t)d, (uint32_t)start_sample, c.nk.data(), c.status.data(), &n_entries));
    g_ms_gpu_count += ms_since(t_gpu);
    c.row_start.resize(n + 1);
    c.colours.resize(n_entries);
    c.counts.resize(n_entries);
    g_entries += n_entries;
    if (g_group) CID_TRY(cid_group_readid_sparse_fetch(g_group, c.row_start.data(), c.colours.data(), c.counts.data()));
    else CID_TRY(cid_readid_sparse_fetch(ctx, c.row_start.data(), c.colours.data(), c.counts.data()));
    g_ms_gpu += ms_since(t_gpu);
}

// ... and the poll (kmer_poll_plus per read, read_id_mt_pe.rs:168-251) + the rows of <prefix>_reads.txt on the host: the reads of a
// batch are independent, so COLORID_POLL_THREADS (default 8) threads format contiguous slices of it and the slices are written in order
// (default 2: the poll of a million reads is 50 ms on one thread, 30 ms on two — enough to stay ahead of the GPU stage)
static int g_poll_threads = [] {
    const char *e = getenv("COLORID_POLL_THREADS");
    const int v = e ? atoi(e) : std::min(2, std::max(1, cpu_budget() / 8));
    return v < 1 ? 1 : v;
}();
static inline void append_u64(std::string &o, uint64_t v) {
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(t[--n]);
}
void poll_batch(const Bigsi &b, const Counted &c, double fp_correct, const std::vector<double> &fp, FILE *out,
                std::map<std::string, uint64_t> &tally, bool &tally_ok) {
    const size_t n = c.rb.size(), C = b.colors.size();
    const auto t_poll = Clock::now();
    const size_t nt = std::min<size_t>((size_t)g_poll_threads, (n + 4095) / 4096);
    // (the slices' text buffers and tallies live across batches: a fresh 2 MB string per slice and batch is 600 page faults, and page
    // faults of several threads at once serialise in the kernel)
    static thread_local std::vector<std::string> text_tl;
    static thread_local std::vector<std::vector<uint64_t>> acc_tl;
    std::vector<std::string> &text = text_tl;              // (references: the slices' threads must see THIS thread's vectors, and a
    std::vector<std::vector<uint64_t>> &acc = acc_tl;      //  thread_local named inside their lambda would be their own)
    if (text.size() < nt) text.resize(nt);
    if (acc.size() < nt) acc.resize(nt);
    // the tally of <prefix>_counts.txt: an accepted read counts under its label — one accession, "no_hits" or "too_short" — every other under "reject"
    for (size_t t = 0; t < nt; ++t) { text[t].clear(); acc[t].assign(C + 3, 0); }   // [C] no_hits, [C+1] too_short, [C+2] reject
    if (memchr(c.rb.id_chars.data(), '\t', c.rb.id_chars.size())) tally_ok = false;
    auto work = [&](size_t t) {
        // (the slice's string is moved onto this thread's stack while it grows: the headers of text[0], text[1], ... share cache lines,
        // and every append writes its string's size — two slices polled side by side took twice as long as one after the other)
        std::string o;
        o.swap(text[t]);
        const size_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        o.reserve((r1 - r0) * 48 + (c.rb.id_off[r1 - 1] - c.rb.id_off[r0]) + 64);
        std::vector<uint8_t> sig(64);
        std::vector<uint64_t> &a = acc[t];
        for (size_t r = r0; r < r1; ++r) {
            o += c.rb.id(r);
            if (c.status[r] == 1) { o += "\ttoo_short\t0\t0\taccept\t0\n"; ++a[C + 1]; continue; }
            const size_t e0 = (size_t)c.row_start[r], ne = (size_t)(c.row_start[r + 1] - c.row_start[r]);
            if (sig.size() < ne + 1) sig.resize(ne + 1);
            const Poll p = poll_core(c.colours.data() + e0, c.counts.data() + e0, ne, c.nk[r], C, fp, fp_correct, sig.data());
            o += '\t';
            if (p.kind == 0) { o += "no_hits"; ++a[C]; }
            else if (p.kind == 1) { o += "no_significant_hits"; ++a[C + 2]; }
            else {
                bool first = true;
                for (size_t e = 0; e < ne; ++e)
                    if (sig[e]) { if (!first) o += ','; first = false; o += b.colors[c.colours[e0 + e]]; }
                ++a[p.n_top == 1 ? p.first_top : C + 2];
            }
            o += '\t'; append_u64(o, p.best);
            o += '\t'; append_u64(o, c.nk[r]);
            o += (p.kind == 0 || (p.kind == 2 && p.n_top == 1)) ? "\taccept\t" : "\treject\t";
            append_u64(o, p.n_top);
            o += '\n';
        }
        o.swap(text[t]);
    };
    static thread_local std::unique_ptr<TaskPool> pool;   // the polling thread's helpers
    if (nt > 1 && !pool) pool.reset(new TaskPool(g_poll_threads - 1));
    if (nt > 1) pool->parallel_for(nt, work);
    else if (nt) work(0);
    const auto t_write = Clock::now();
    for (size_t t = 0; t < nt; ++t) fwrite(text[t].data(), 1, text[t].size(), out);
    g_ms_write += ms_since(t_write);
    for (size_t t = 0; t < nt; ++t) {
        const std::vector<uint64_t> &a = acc[t];
        for (size_t col = 0; col < C; ++col) if (a[col]) tally[b.colors[col]] += a[col];
        if (a[C]) tally["no_hits"] += a[C];
        if (a[C + 1]) tally["too_short"] += a[C + 1];
        if (a[C + 2]) tally["reject"] += a[C + 2];
    }
    g_ms_poll += ms_since(t_poll);
}

std::vector<double> false_prob_map(const Bigsi &b) {  // read_id_mt_pe.rs:18-38
    std::vector<double> fp(b.colors.size());
    for (size_t c = 0; c < fp.size(); ++c) fp[c] = false_prob((double)b.bloom_size, (double)b.num_hash, (double)b.n_ref_kmers[c]);
    return fp;
}

// Three stages beside the caller's parsing of batch i+2 (and the LineReaders inflating further ahead): the GPU call of batch i+1 on
// one thread, the poll + output of batch i on another — the phases the reference runs back to back (read_id_mt_pe.rs:864-907).
// Rows leave in submission order.  The counting thread is the only one that touches `ctx` while it runs.
class BatchClassifier {
  public:
    BatchClassifier(cid_ctx *ctx, const Bigsi &b, size_t d, double fp_correct, size_t start_sample, const std::vector<double> &fp, FILE *out,
                    const char *progress_fmt)
        : ctx_(ctx), b_(b), d_(d), fp_correct_(fp_correct), start_sample_(start_sample), fp_(fp), out_(out), progress_fmt_(progress_fmt),
          counter_([this] { run_count(); }), poller_([this] { run_poll(); }) {
        for (const std::string &name : b.colors) if (name.find('\t') != std::string::npos) tally_ok_ = false;
    }
    // hands `rb` over and leaves an empty batch in its place; waits while kDepth batches are queued
    void submit(ReadBatch &rb) {
        if (rb.size() == 0) return;
        std::unique_lock<std::mutex> lk(mu_);
        const auto tw = Clock::now();
        cv_room_.wait(lk, [&] { return queue_.size() < kDepth; });
        g_ms_wait[0] += ms_since(tw);
        queue_.push_back(std::move(rb));
        if (!spare_.empty()) { rb = std::move(spare_.back()); spare_.pop_back(); }
        else rb = ReadBatch();
        rb.clear();
        cv_work_.notify_one();
    }
    // a batch whose buffers are already grown (its reads were polled), for the record packers; an empty one when none is free
    ReadBatch spare() {
        std::lock_guard<std::mutex> lk(mu_);
        if (spare_.empty()) return ReadBatch();
        ReadBatch rb = std::move(spare_.back());
        spare_.pop_back();
        return rb;
    }
    // the device front end (cid_fastq) hands over batches that are counted already: an empty Counted to fill, and its way to the poll
    std::unique_ptr<Counted> take_counted() {
        std::lock_guard<std::mutex> lk(mu_);
        if (free_counted_.empty()) return std::unique_ptr<Counted>(new Counted);
        std::unique_ptr<Counted> c = std::move(free_counted_.back());
        free_counted_.pop_back();
        return c;
    }
    void push_counted(std::unique_ptr<Counted> c) {
        std::unique_lock<std::mutex> lk(mu_);
        const auto tw = Clock::now();
        cv_polled_.wait(lk, [&] { return counted_.size() < kDepth; });
        g_ms_wait[2] += ms_since(tw);
        counted_.push_back(std::move(c));
        cv_counted_.notify_one();
    }
    uint64_t finish() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            done_ = true;
        }
        cv_work_.notify_one();
        counter_.join();
        poller_.join();
        g_read_counts.swap(tally_);
        g_read_counts_valid = tally_ok_;
        return n_reads_;
    }
  private:
    static constexpr size_t kDepth = 2;
    void run_count() {
        for (;;) {
            std::unique_ptr<Counted> c;
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!free_counted_.empty()) { c = std::move(free_counted_.back()); free_counted_.pop_back(); }
                else c.reset(new Counted);
                const auto tw = Clock::now();
                cv_work_.wait(lk, [&] { return done_ || !queue_.empty(); });
                g_ms_wait[1] += ms_since(tw);
                if (queue_.empty()) break;
                c->rb = std::move(queue_.front());
                queue_.pop_front();
                cv_room_.notify_one();
            }
            count_batch(ctx_, b_, *c, d_, start_sample_);
            std::unique_lock<std::mutex> lk(mu_);
            const auto tw = Clock::now();
            cv_polled_.wait(lk, [&] { return counted_.size() < kDepth; });
            g_ms_wait[2] += ms_since(tw);
            counted_.push_back(std::move(c));
            cv_counted_.notify_one();
        }
        std::lock_guard<std::mutex> lk(mu_);
        count_done_ = true;
        cv_counted_.notify_one();
    }
    void run_poll() {
        for (;;) {
            std::unique_ptr<Counted> c;
            {
                std::unique_lock<std::mutex> lk(mu_);
                const auto tw = Clock::now();
                cv_counted_.wait(lk, [&] { return count_done_ || !counted_.empty(); });
                g_ms_wait[3] += ms_since(tw);
                if (counted_.empty()) return;
                c = std::move(counted_.front());
                counted_.pop_front();
                cv_polled_.notify_one();
            }
            poll_batch(b_, *c, fp_correct_, fp_, out_, tally_, tally_ok_);
            n_reads_ += c->rb.size();
            fprintf(stderr, progress_fmt_, (unsigned long long)n_reads_);
            c->rb.clear();
            std::lock_guard<std::mutex> lk(mu_);
            if (spare_.size() < 16) spare_.push_back(std::move(c->rb));
            c->rb = ReadBatch();
            free_counted_.push_back(std::move(c));   // its result vectors keep their capacity for a later batch
        }
    }
    cid_ctx *ctx_;
    const Bigsi &b_;
    size_t d_;
    double fp_correct_;
    size_t start_sample_;
    const std::vector<double> &fp_;
    FILE *out_;
    const char *progress_fmt_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_room_, cv_counted_, cv_polled_;
    std::deque<ReadBatch> queue_;
    std::deque<std::unique_ptr<Counted>> counted_;
    std::vector<std::unique_ptr<Counted>> free_counted_;
    std::vector<ReadBatch> spare_;
    bool done_ = false, count_done_ = false;
    uint64_t n_reads_ = 0;   // the polling thread's, read by finish() after the join
    std::map<std::string, uint64_t> tally_;   // (label or "reject") -> reads, for <prefix>_counts.txt
    bool tally_ok_ = true;
    std::thread counter_, poller_;   // last members: start after everything above is initialised
};

}  // namespace

static void print_read_id_timing(const Clock::time_point &t0) {
    if (!g_timing) return;
    fprintf(stderr,
            "timing: total %.0f ms, GPU calls (copies + kernels) %.0f ms, poll + write %.0f ms; waits: parser on a full queue %.0f ms, "
            "GPU stage idle %.0f ms, GPU stage on the poll %.0f ms, poll idle %.0f ms; of the GPU calls: counting %.0f ms, "
            "%llu (colour, count) entries fetched; of poll + write: writing %.0f ms\n",
            ms_since(t0), g_ms_gpu, g_ms_poll, g_ms_wait[0], g_ms_wait[1], g_ms_wait[2], g_ms_wait[3], g_ms_gpu_count, (unsigned long long)g_entries,
            g_ms_write);
}

// ---- block-gzip input through the device front end (cid_fastq_*): the members go up compressed — read from the file a stretch ahead by
// BgzfMemberReader — and come back classified: per read the id line, n_kmers, status and its (colour, count) entries, ready for the
// poll.  No inflating threads, no record packers: the host reads the file, polls and writes.  COLORID_DEVICE_FASTQ=0 keeps the host
// front end; several GPUs (--gpus / --placement) use it too.
bool read_id_mt_pe::device_fastq_wanted(const std::vector<std::string> &fq, size_t n_files) {
    // default: on, for single-end input (16 M reads: 0.66-0.70 s against 1.05-1.11 s through the host front end on a 16-CPU share) and
    // for pairs (4 M pairs: 0.33-0.34 s against 0.36-0.38 s) — profiles/r03_frontend_16m.txt; COLORID_DEVICE_FASTQ=0 keeps the host's
    const char *e = getenv("COLORID_DEVICE_FASTQ");
    if ((e && atoi(e) == 0) || g_group) return false;
    (void)n_files;
    for (size_t i = 0; i < n_files; ++i)
        if (!BgzfMemberReader::is_bgzf(fq[i])) return false;
    return true;
}
// a stretch: ~256 MiB of text (800 000 reads of 150 bp: a DEFLATE stream decodes serially, so a launch takes ~14 ms however few members
// it holds), fewer when the dense report rows of its reads would pass 8 GiB
size_t read_id_mt_pe::device_fastq_stretch_bytes(size_t n_colors) {
    size_t target = (size_t)(getenv("COLORID_DEVICE_FASTQ_MB") ? atoi(getenv("COLORID_DEVICE_FASTQ_MB")) : 256) << 20;
    const size_t by_rows = ((size_t)8 << 30) / ((n_colors + 1) * 4) * 300;
    if (target > by_rows) target = by_rows < ((size_t)1 << 20) ? ((size_t)1 << 20) : by_rows;
    return target;
}
// DEFLATE decodes serially inside a member, so the device inflates a member per lane at ~10 GB/s of text in all, while the cores the
// host front end would spend on inflating and packing are idle: the reader's threads inflate this share of every stretch (the rest
// goes up compressed).  Measured on a 16-CPU share (tools/exp_frontend.sh): see DESIGN.md.
double read_id_mt_pe::device_fastq_host_share() {
    if (const char *e = getenv("COLORID_DEVICE_FASTQ_HOST_SHARE")) { const double v = atof(e); return v < 0.0 ? 0.0 : v > 1.0 ? 1.0 : v; }
    // a host thread inflates ~0.7 GB/s of text, the device ~10 GB/s beside the classification it also runs: with the eight threads a
    // 16-CPU share leaves, an even split keeps both sides busy (16 M reads, tools/exp_frontend_16m.sh: 0.71-0.73 s at share 0.5, 0.84-0.90 s
    // with the host inflating everything, 0.89-0.96 s with the device inflating everything; host front end 1.02-1.21 s)
    const double v = (double)device_fastq_host_threads(1) * 0.0625;
    return v > 1.0 ? 1.0 : v;
}
int read_id_mt_pe::device_fastq_host_threads(size_t n_files) {
    // the process's CPU share (cgroup quota) minus the threads that are busy anyway — four polling, the readers, the GPU stage, the
    // writer: a process whose runnable threads exceed its quota is throttled as a whole, GPU stage included
    const char *e = getenv("COLORID_GZ_THREADS");
    const int v = e ? atoi(e) : std::max(1, (cpu_budget() - 8) / (int)(n_files ? n_files : 1));
    return v < 1 ? 1 : v > 12 ? 12 : v;
}
namespace {
// false: the input is not this path's (reads too long for the LDS kernels) and nothing has been written yet — the caller falls back
bool classify_bgzf_on_device(cid_ctx *ctx, const std::vector<std::string> &fq, size_t n_files, const Bigsi &b, size_t d, size_t start_sample,
                             uint8_t qual_offset, BatchClassifier &classifier) {
    const auto t_enter = Clock::now();
    cid_fastq *fr = nullptr;
    CID_TRY(cid_fastq_create(ctx, (int)n_files, qual_offset, &fr));
    const size_t target = read_id_mt_pe::device_fastq_stretch_bytes(b.colors.size());
    std::unique_ptr<BgzfMemberReader> rd[2];
    for (size_t i = 0; i < n_files; ++i)   // (reading since before the index load, main.cpp)
        rd[i] = BgzfMemberReader::open(fq[i], target, read_id_mt_pe::device_fastq_host_share(), read_id_mt_pe::device_fastq_host_threads(n_files));
    // stretches pushed ahead of the one being classified: their inflate launches (alternating streams in the library) overlap, which
    // matters because a launch cannot be shorter than the decoding of one member (~14 ms) however few members it holds
    const size_t ahead = [] { const char *e = getenv("COLORID_DEVICE_FASTQ_AHEAD"); const long v = e ? atol(e) : 1; return (size_t)(v < 1 ? 1 : v > 6 ? 6 : v); }();
    std::vector<BgzfStretch> st[2];   // per file ahead + 1 stretches in turn: the one pushed last stays untouched while its text is still on the bus
    st[0].resize(ahead + 1); st[1].resize(ahead + 1);
    size_t turn[2] = {0, 0};
    bool more[2] = {true, n_files == 2};
    size_t pending[2] = {0, 0};   // stretches pushed and not yet taken by a classify call
    double ms_read = 0, ms_push = 0, ms_classify = 0, ms_fetch = 0;
    auto push_next = [&](size_t i) {
        if (!more[i]) return;
        const auto tr = Clock::now();
        BgzfStretch &sx = st[i][turn[i]++ % (ahead + 1)];
        const bool got = rd[i]->next(sx);
        ms_read += ms_since(tr);
        if (!got) { more[i] = false; return; }
        const auto tp = Clock::now();
        // the device's members first, then the text the reader's threads inflated (two pushes: classify takes both)
        const bool host_part = sx.host_text_bytes > 0;
        CID_TRY(cid_fastq_push_bgzf(fr, (int)i, sx.bytes.data(), sx.bytes.size(), sx.off.data(), sx.len.data(), sx.text_len.data(), sx.device_members,
                                    sx.last && !host_part ? CID_FASTQ_LAST : 0));
        if (host_part)
            CID_TRY(cid_fastq_push_text(fr, (int)i, sx.host_text.p, sx.host_text_bytes, (sx.last ? CID_FASTQ_LAST : 0) | (sx.host_text.pinned ? CID_FASTQ_KEEP : 0)));
        if (sx.last) more[i] = false;
        ms_push += ms_since(tp);
        ++pending[i];
    };
    auto t_gpu = Clock::now();
    for (size_t i = 0; i < n_files; ++i) push_next(i);
    const double ms_setup = ms_since(t_enter);
    bool first = true;
    while (pending[0] || pending[1]) {
        for (size_t i = 0; i < n_files; ++i)   // the stretches after this one: inflated while this one is classified
            while (more[i] && pending[i] < ahead + 1) push_next(i);
        uint64_t n = 0, ne = 0, idb = 0;
        const auto tc = Clock::now();
        const int rc = cid_fastq_classify(fr, b.index, (uint32_t)d, (uint32_t)start_sample, 2, &n, &ne, &idb);
        ms_classify += ms_since(tc);
        for (size_t i = 0; i < n_files; ++i) if (pending[i]) --pending[i];
        if (rc == CID_ERR_UNSUPPORTED && first) {
            fprintf(stderr, "note: %s — using the host front end\n", cid_last_error());
            cid_fastq_destroy(fr);
            return false;
        }
        if (rc != CID_OK) die("%s%s", cid_last_error(), rc == CID_ERR_UNSUPPORTED ? " (rerun with COLORID_DEVICE_FASTQ=0)" : "");
        g_ms_gpu_count += ms_since(t_gpu);
        first = false;
        if (n) {
            std::unique_ptr<Counted> c = classifier.take_counted();
            c->rb.clear();
            c->nk.resize(n); c->status.resize(n); c->row_start.resize(n + 1); c->colours.resize(ne); c->counts.resize(ne);
            c->rb.id_off.resize(n + 1);
            c->rb.id_chars.resize(idb);
            const auto tf = Clock::now();
            CID_TRY(cid_fastq_fetch(fr, c->nk.data(), c->status.data(), c->row_start.data(), c->colours.data(), c->counts.data(), c->rb.id_off.data(),
                                    &c->rb.id_chars[0]));
            ms_fetch += ms_since(tf);
            c->rb.id_off.resize(n);   // (ReadBatch counts its reads by the ids)
            g_entries += ne;
            g_ms_gpu += ms_since(t_gpu);
            classifier.push_counted(std::move(c));
        } else g_ms_gpu += ms_since(t_gpu);
        t_gpu = Clock::now();
    }
    cid_fastq_destroy(fr);
    if (g_timing)
        fprintf(stderr, "timing: device front end: waiting for the file reader %.0f ms, push (H2D of the members) %.0f ms, classify %.0f ms, fetch %.0f ms; "
                "%.0f ms until the first stretch was pushed, %.0f ms in all\n", ms_read, ms_push, ms_classify, ms_fetch, ms_setup, ms_since(t_enter));
    return true;
}
}  // namespace

void read_id_mt_pe::per_read_stream_se(cid_ctx *ctx, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct,
                                       size_t batch, const std::string &prefix, uint8_t qual_offset, size_t start_sample) {
    const auto t0 = Clock::now();
    const std::vector<double> fp = false_prob_map(b);
    FILE *out = fopen((prefix + "_reads.txt").c_str(), "w");
    if (!out) die("could not create outfile!");
    ReadBatch rb;
    const bool on_device = device_fastq_wanted(fq, 1);
    if (on_device && !getenv("COLORID_POLL_THREADS")) g_poll_threads = std::max(g_poll_threads, std::min(4, cpu_budget() / 4));   // no packing threads beside them
    BatchClassifier classifier(ctx, b, d, fp_correct, start_sample, fp, out, "%llu read pairs classified\r");
    if (g_timing) fprintf(stderr, "timing: %.0f ms of set-up before the first read\n", ms_since(t0));
    if (!on_device || !classify_bgzf_on_device(ctx, fq, 1, b, d, start_sample, qual_offset, classifier))
    stream_fastq_records(fq[0], nullptr, qual_offset, true, [&](ReadBatch &&piece) {
        if (rb.size() == 0) rb = std::move(piece); else rb.append(piece);
        if (rb.size() >= batch || rb.heavy()) classifier.submit(rb);   // (batches close on piece boundaries: at least `batch` reads each)
    }, [&] { return classifier.spare(); });
    classifier.submit(rb);
    const auto t_drain = Clock::now();
    const uint64_t read_count = classifier.finish();
    const double ms_drain = ms_since(t_drain);
    const auto t_close = Clock::now();
    fclose(out);
    if (g_timing) fprintf(stderr, "timing: %.0f ms from the end of the input to the last row written, %.0f ms closing the output\n", ms_drain, ms_since(t_close));
    fprintf(stderr, "Classified %llu reads in %ld seconds\n", (unsigned long long)read_count, secs_since(t0));
    print_read_id_timing(t0);
}

void read_id_mt_pe::per_read_stream_pe(cid_ctx *ctx, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct,
                                       size_t batch, const std::string &prefix, uint8_t qual_offset, size_t start_sample) {
    const auto t0 = Clock::now();
    const std::vector<double> fp = false_prob_map(b);
    FILE *out = fopen((prefix + "_reads.txt").c_str(), "w");
    if (!out) die("could not create outfile!");
    ReadBatch rb;
    const bool on_device = device_fastq_wanted(fq, 2);
    if (on_device && !getenv("COLORID_POLL_THREADS")) g_poll_threads = std::max(g_poll_threads, std::min(4, cpu_budget() / 4));
    BatchClassifier classifier(ctx, b, d, fp_correct, start_sample, fp, out, "%llu read pairs classified\r");
    if (!on_device || !classify_bgzf_on_device(ctx, fq, 2, b, d, start_sample, qual_offset, classifier))
    stream_fastq_records(fq[0], &fq[1], qual_offset, true, [&](ReadBatch &&piece) {
        if (rb.size() == 0) rb = std::move(piece); else rb.append(piece);
        if (rb.size() >= batch || rb.heavy()) classifier.submit(rb);
    }, [&] { return classifier.spare(); });
    classifier.submit(rb);
    const uint64_t read_count = classifier.finish();
    fclose(out);
    fprintf(stderr, "Classified %llu read pairs in %ld seconds\n", (unsigned long long)read_count, secs_since(t0));
    print_read_id_timing(t0);
}

void read_id_mt_pe::stream_fasta(cid_ctx *ctx, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct,
                                 size_t batch, const std::string &prefix, size_t start_sample) {
    // read_line() keeps the '\n' inside the sequence (k-mers across a line break fail has_no_n); ids keep the '>'
    const auto t0 = Clock::now();
    const std::vector<double> fp = false_prob_map(b);
    FILE *out = fopen((prefix + "_reads.txt").c_str(), "w");
    if (!out) die("could not create outfile!");
    FILE *f = fopen(fq[0].c_str(), "rb");
    if (!f) die("file not found: %s", fq[0].c_str());
    ReadBatch rb;
    BatchClassifier classifier(ctx, b, d, fp_correct, start_sample, fp, out, " %llu reads classified\r");
    std::string sub, id, l;
    uint64_t count = 0;
    char *lineptr = nullptr;
    size_t cap = 0;
    ssize_t got;
    while ((got = getline(&lineptr, &cap, f)) > 0) {
        l.assign(lineptr, (size_t)got);
        if (count == 0) {
            id = l.substr(0, l.size() - 1);
        } else if (l.find('>') != std::string::npos) {
            if (!sub.empty()) {
                rb.push(id, &sub, 1);
                id = l.substr(0, l.size() - 1);
                sub.clear();
            }
        } else {
            sub += l;
        }
        ++count;
        if (rb.size() > 0 && (rb.size() % batch == 0 || rb.heavy())) classifier.submit(rb);
    }
    free(lineptr);
    fclose(f);
    rb.push(id, &sub, 1);
    classifier.submit(rb);
    const uint64_t read_count = classifier.finish();
    fclose(out);
    fprintf(stderr, "Classified %llu reads in %ld seconds\n", (unsigned long long)read_count, secs_since(t0));
}

}  // namespace colorid
