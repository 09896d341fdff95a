// The reference's read_id drivers with the hot loop replaced by C-ABI calls:
//   read_id_mt_pe::per_read_stream_se/_pe, stream_fasta (src/read_id_mt_pe.rs) -> cid_readid_count* / cid_fastq_*
// plus the CPU-side tail (kmer_poll_plus, the counts file of src/reports.rs).  File formats follow the reference.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "drivers_common.hpp"
#include <unistd.h>

namespace colorid {

static double g_ms_gpu = 0, g_ms_poll = 0, g_ms_gpu_count = 0, g_ms_write = 0;
static uint64_t g_entries = 0;
static double g_ms_wait[4] = {0, 0, 0, 0};   // parser blocked by a full queue | GPU stage idle | GPU stage blocked by the poll | poll idle

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
    const char *e = cli_env("COLORID_POLL_THREADS");
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
    void ids_stay_with_counted(bool on) { std::lock_guard<std::mutex> lk(mu_); ids_stay_ = on; }
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
            std::lock_guard<std::mutex> lk(mu_);
            if (!ids_stay_) {   // (the device front end fills a Counted's ids itself: they stay with it, sized, for the next stretch)
                c->rb.clear();
                if (spare_.size() < 16) spare_.push_back(std::move(c->rb));
                c->rb = ReadBatch();
            }
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
    bool done_ = false, count_done_ = false, ids_stay_ = false;
    uint64_t n_reads_ = 0;   // the polling thread's, read by finish() after the join
    std::map<std::string, uint64_t> tally_;   // (label or "reject") -> reads, for <prefix>_counts.txt
    bool tally_ok_ = true;
    std::thread counter_, poller_;   // last members: start after everything above is initialised
};

}  // namespace

static void reset_read_id_timing() {
    g_ms_gpu = g_ms_poll = g_ms_gpu_count = g_ms_write = 0;
    g_entries = 0;
    for (double &w : g_ms_wait) w = 0;
}
static void print_read_id_timing(const Clock::time_point &t0) {
    if (!g_timing) return;
    fprintf(stderr,
            "timing: total %.0f ms, GPU calls (copies + kernels) %.0f ms, poll + write %.0f ms; waits: parser on a full queue %.0f ms, "
            "GPU stage idle %.0f ms, GPU stage on the poll %.0f ms, poll idle %.0f ms; of the GPU calls: counting %.0f ms, "
            "%llu (colour, count) entries fetched; of poll + write: writing %.0f ms\n",
            ms_since(t0), g_ms_gpu, g_ms_poll, g_ms_wait[0], g_ms_wait[1], g_ms_wait[2], g_ms_wait[3], g_ms_gpu_count, (unsigned long long)g_entries,
            g_ms_write);
    fprintf(stderr, "timing: the input's decoding threads waited %.0f ms for the parser to take their blocks\n", LineReader::blocked_ms());
    // batch_id runs one stream per sample in the same process: the next one's line starts from zero
    reset_read_id_timing();
}

// ---- block-gzip input through the device front end (cid_fastq_*): the members go up compressed — read from the file a stretch ahead by
// BgzfMemberReader — and come back classified: per read the id line, n_kmers, status and its (colour, count) entries, ready for the
// poll.  No inflating threads, no record packers: the host reads the file, polls and writes.  COLORID_DEVICE_FASTQ=0 keeps the host
// front end; several GPUs (--gpus / --placement) use it too.
bool read_id_mt_pe::device_fastq_wanted(const std::vector<std::string> &fq, size_t n_files) {
    // default: on, for single-end input (16 M reads: 0.66-0.70 s against 1.05-1.11 s through the host front end on a 16-CPU share) and
    // for pairs (4 M pairs: 0.33-0.34 s against 0.36-0.38 s) — profiles/r03_frontend_16m.txt; COLORID_DEVICE_FASTQ=0 keeps the host's
    const char *e = cli_env("COLORID_DEVICE_FASTQ");
    if ((e && atoi(e) == 0) || g_group) return false;
    (void)n_files;
    for (size_t i = 0; i < n_files; ++i)
        if (!BgzfMemberReader::is_bgzf(fq[i])) return false;
    return true;
}
// a stretch: ~256 MiB of text (800 000 reads of 150 bp: a DEFLATE stream decodes serially, so a launch takes ~14 ms however few members
// it holds), fewer when the dense report rows of its reads would pass 8 GiB
size_t read_id_mt_pe::device_fastq_stretch_bytes(size_t n_colors) {
    size_t target = (size_t)(cli_env("COLORID_DEVICE_FASTQ_MB") ? atoi(cli_env("COLORID_DEVICE_FASTQ_MB")) : 256) << 20;
    const size_t by_rows = ((size_t)8 << 30) / ((n_colors + 1) * 4) * 300;
    if (target > by_rows) target = by_rows < ((size_t)1 << 20) ? ((size_t)1 << 20) : by_rows;
    return target;
}
// DEFLATE decodes serially inside a member, so the device inflates a member per lane (rounds 2-4) or per wave (round 5 on), while the
// cores the host front end would spend on inflating and packing are idle: the reader's threads inflate this share of every stretch (the
// rest goes up compressed): 0.0375 per spare thread, 0.3 on a 16-CPU share (round 3, tools/exp_frontend_poll.sh: 16 M reads 0.44-0.51 s
// at 0.5, 0.41-0.44 s at 0.3).  Round 6 measured it again (tools/exp_frontend_share.sh, profiles/r06_frontend_share.txt): with everything
// inflated on the device and the freed threads in the poll (ten instead of six) 16 M reads take the same 313-335 ms as at 0.3 (297-352),
// 4 M pairs 101-102 ms instead of 117-122, one million reads 39-41 instead of 33-35 — the loop waits for the GPU either way (the inflate
// and the classifier of a 256 MiB stretch, 13.3 ms side by side: profiles/r06_frontend_gpu_bound.txt).  The default stays.
double read_id_mt_pe::device_fastq_host_share() {
    if (const char *e = cli_env("COLORID_DEVICE_FASTQ_HOST_SHARE")) { const double v = atof(e); return v < 0.0 ? 0.0 : v > 1.0 ? 1.0 : v; }
    const double v = (double)device_fastq_host_threads(1) * 0.0375;
    return v > 1.0 ? 1.0 : v;
}
int read_id_mt_pe::device_fastq_host_threads(size_t n_files) {
    // the process's CPU share (cgroup quota) minus the threads that are busy anyway — four polling, the readers, the GPU stage, the
    // writer: a process whose runnable threads exceed its quota is throttled as a whole, GPU stage included
    const char *e = cli_env("COLORID_GZ_THREADS");
    const int v = e ? atoi(e) : std::max(1, (cpu_budget() - 8) / (int)(n_files ? n_files : 1));
    return v < 1 ? 1 : v > 12 ? 12 : v;
}
namespace {
// kFrontEndDone: every read went through the device front end.  kFrontEndNotMine: the input is not this path's (reads too long for the
// LDS kernels, a step over the dense-row limit) — in the FIRST step, nothing has been written: the caller takes the host front end.
// kFrontEndRestart: the same in a LATER step, rows of the earlier stretches are already on their way to the output: the caller
// finishes the classifier, empties the output and runs the whole input through the host front end (which takes such reads) — an input
// that the host path completes is never a hard failure here.
enum FrontEnd { kFrontEndDone, kFrontEndNotMine, kFrontEndRestart };
FrontEnd classify_bgzf_on_device(cid_ctx *ctx, const std::vector<std::string> &fq, size_t n_files, const Bigsi &b, size_t d, size_t start_sample,
                             uint8_t qual_offset, BatchClassifier &classifier) {
    const auto t_enter = Clock::now();
    cid_fastq *fr = nullptr;
    CID_TRY(cid_fastq_create(ctx, (int)n_files, qual_offset, &fr));
    classifier.ids_stay_with_counted(true);
    const size_t target = read_id_mt_pe::device_fastq_stretch_bytes(b.colors.size());
    std::unique_ptr<BgzfMemberReader> rd[2];
    for (size_t i = 0; i < n_files; ++i)   // (reading since before the index load, main.cpp)
        rd[i] = BgzfMemberReader::open(fq[i], target, read_id_mt_pe::device_fastq_host_share(), read_id_mt_pe::device_fastq_host_threads(n_files));
    // stretches pushed ahead of the one being classified: their inflate launches (alternating streams in the library) overlap, which
    // matters because a launch cannot be shorter than the decoding of one member (~14 ms) however few members it holds
    const size_t ahead = [] { const char *e = cli_env("COLORID_DEVICE_FASTQ_AHEAD"); const long v = e ? atol(e) : 1; return (size_t)(v < 1 ? 1 : v > 6 ? 6 : v); }();
    std::vector<BgzfStretch> st[2];   // per file ahead + 1 stretches in turn: the one pushed last stays untouched while its text is still on the bus
    st[0].resize(ahead + 1); st[1].resize(ahead + 1);
    size_t turn[2] = {0, 0};
    bool more[2] = {true, n_files == 2};
    size_t pending[2] = {0, 0};   // stretches pushed and not yet taken by a classify call
    double ms_read = 0, ms_push = 0, ms_classify = 0, ms_fetch = 0, ms_buffers = 0, ms_handover = 0;
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
        // (only the device's members travel: they come first in the stretch; the rest was inflated here)
        const size_t dev_bytes = sx.device_members ? (size_t)sx.off[sx.device_members - 1] + sx.len[sx.device_members - 1] : 0;
        // (page-locked bytes: the copy runs on beside this thread; the library waits for it in the NEXT push of this file, which comes
        // before the reader gets this stretch's buffer back — st[i] holds ahead + 1 >= 2 of them in turn)
        CID_TRY(cid_fastq_push_bgzf(fr, (int)i, sx.bytes.data(), dev_bytes, sx.off.data(), sx.len.data(), sx.text_len.data(), sx.device_members,
                                    sx.bytes.pinned ? CID_FASTQ_KEEP : 0));
        // ALWAYS the second push, also when it is empty: a classify step takes two pushes per file = exactly one stretch, and the wait
        // for the copy of the stretch before (CID_FASTQ_KEEP) happens here, before the reader gets that stretch's buffer back
        CID_TRY(cid_fastq_push_text(fr, (int)i, host_part ? sx.host_text.p : nullptr, host_part ? sx.host_text_bytes : 0,
                                    (sx.last ? CID_FASTQ_LAST : 0) | (host_part && sx.host_text.pinned ? CID_FASTQ_KEEP : 0)));
        if (sx.last) more[i] = false;
        ms_push += ms_since(tp);
        ++pending[i];
    };
    auto t_gpu = Clock::now();
    for (size_t i = 0; i < n_files; ++i) push_next(i);
    const double ms_setup = ms_since(t_enter);
    // One step = cid_fastq_classify_begin (records cut and packed, the classifier launched) ... cid_fastq_classify_end (its report
    // compacted).  Between the two the classifier runs for ~10 ms per stretch and this thread has the GPU-free work to do: fetching
    // the results of the step BEFORE (they stay valid until the next _end, and travel on a stream of their own) and pushing the
    // stretch after the next.  Done the other way round — classify, fetch, push, in a row — the GPU idled ~5 ms of every 18.
    bool first = true;
    size_t n_steps = 0;
    uint64_t have_n = 0, have_ne = 0, have_idb = 0;   // an ended step whose results are still on the device
    bool have = false;
    for (;;) {
        for (size_t i = 0; i < n_files; ++i)   // (only the start: ahead + 1 stretches on their way before the first step)
            while (first && more[i] && pending[i] < ahead + 1) push_next(i);
        const bool begin = pending[0] || pending[1];
        if (begin) {
            const auto tc = Clock::now();
            const int rc = cid_fastq_classify_begin(fr, b.index, (uint32_t)d, (uint32_t)start_sample, 2);
            ms_classify += ms_since(tc);
            for (size_t i = 0; i < n_files; ++i) if (pending[i]) --pending[i];
            if (rc == CID_ERR_UNSUPPORTED) {   // (tests inject one through the library: cid_ctx_tune fastq_refuse_at_step / CID_FASTQ_REFUSE_AT_STEP)
                fprintf(stderr, "note: %s — %s\n", cid_last_error(), first ? "using the host front end" : "starting over with the host front end");
                cid_fastq_destroy(fr);
                classifier.ids_stay_with_counted(false);
                return first ? kFrontEndNotMine : kFrontEndRestart;
            }
            if (rc != CID_OK) die("%s", cid_last_error());
            first = false;
            ++n_steps;
        }
        if (have && have_n) {
            const auto tb = Clock::now();
            std::unique_ptr<Counted> c = classifier.take_counted();
            c->rb.bases.clear(); c->rb.seq_off.assign(1, 0); c->rb.read_seq0.assign(1, 0);   // (the ids keep their size: a recycled batch is not filled with zeros again)
            c->nk.resize(have_n); c->status.resize(have_n); c->row_start.resize(have_n + 1); c->colours.resize(have_ne); c->counts.resize(have_ne);
            c->rb.id_off.resize(have_n + 1);
            c->rb.id_chars.resize(have_idb);
            const auto tf = Clock::now();
            ms_buffers += ms_since(tb);
            CID_TRY(cid_fastq_fetch(fr, c->nk.data(), c->status.data(), c->row_start.data(), c->colours.data(), c->counts.data(), c->rb.id_off.data(),
                                    &c->rb.id_chars[0]));
            ms_fetch += ms_since(tf);
            c->rb.id_off.resize(have_n);   // (ReadBatch counts its reads by the ids)
            g_entries += have_ne;
            const auto th = Clock::now();
            classifier.push_counted(std::move(c));
            ms_handover += ms_since(th);
        }
        have = false;
        for (size_t i = 0; i < n_files; ++i)   // the stretches after this one: inflated while this one is classified
            while (more[i] && pending[i] < ahead + 1) push_next(i);
        if (!begin) break;
        const auto tc = Clock::now();
        const int rc = cid_fastq_classify_end(fr, &have_n, &have_ne, &have_idb);
        ms_classify += ms_since(tc);
        if (rc != CID_OK) die("%s", cid_last_error());
        have = true;
        g_ms_gpu_count += ms_since(t_gpu);
        g_ms_gpu += ms_since(t_gpu);
        t_gpu = Clock::now();
    }
    cid_fastq_destroy(fr);
    if (g_timing)
        fprintf(stderr, "timing: device front end: waiting for the file reader %.0f ms, push (H2D of the members) %.0f ms, classify %.0f ms, fetch %.0f ms "
                "(+ %.0f ms sizing its buffers), handing the rows to the poll %.0f ms; %.0f ms until the first stretch was pushed, %.0f ms in all\n",
                ms_read, ms_push, ms_classify, ms_fetch, ms_buffers, ms_handover, ms_setup, ms_since(t_enter));
    return kFrontEndDone;
}

// the rows written so far belong to a run that is being started over: an empty output again, and the abandoned run's share of the
// `timing:` lines forgotten.  An output that cannot be emptied (a FIFO, a character device: what was written is gone) ends the run.
void empty_output(FILE *out, const std::string &prefix) {
    struct stat sb;
    if (fflush(out) != 0 || fstat(fileno(out), &sb) != 0) die("could not empty the outfile for the restart");
    if (!S_ISREG(sb.st_mode)) die("the device front end gave up mid-run and %s_reads.txt is not a regular file: rows already written cannot be taken back; "
                                  "run again with COLORID_DEVICE_FASTQ=0", prefix.c_str());
    if (ftruncate(fileno(out), 0) != 0) die("could not empty the outfile for the restart");
    rewind(out);
    reset_read_id_timing();
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
    if (on_device && !cli_env("COLORID_POLL_THREADS"))
        g_poll_threads = std::max(g_poll_threads, read_id_mt_pe::device_fastq_host_share() > 0.0 ? std::min(6, cpu_budget() * 3 / 8) : std::min(12, cpu_budget() * 5 / 8));   // no packing threads beside them: a stretch's poll on 4 threads takes 18 ms, the GPU side 13
    auto make_classifier = [&] { return std::unique_ptr<BatchClassifier>(new BatchClassifier(ctx, b, d, fp_correct, start_sample, fp, out, "%llu read pairs classified\r")); };
    std::unique_ptr<BatchClassifier> classifier = make_classifier();
    if (g_timing) fprintf(stderr, "timing: %.0f ms of set-up before the first read\n", ms_since(t0));
    FrontEnd fe = on_device ? classify_bgzf_on_device(ctx, fq, 1, b, d, start_sample, qual_offset, *classifier) : kFrontEndNotMine;
    if (fe == kFrontEndRestart) {
        classifier->finish();
        empty_output(out, prefix);
        classifier = make_classifier();
    }
    if (fe != kFrontEndDone)
    stream_fastq_records(fq[0], nullptr, qual_offset, true, [&](ReadBatch &&piece) {
        if (rb.size() == 0) rb = std::move(piece); else rb.append(piece);
        if (rb.size() >= batch || rb.heavy()) classifier->submit(rb);   // (batches close on piece boundaries: at least `batch` reads each)
    }, [&] { return classifier->spare(); });
    classifier->submit(rb);
    const auto t_drain = Clock::now();
    const uint64_t read_count = classifier->finish();
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
    if (on_device && !cli_env("COLORID_POLL_THREADS"))
        g_poll_threads = std::max(g_poll_threads, read_id_mt_pe::device_fastq_host_share() > 0.0 ? std::min(6, cpu_budget() * 3 / 8) : std::min(12, cpu_budget() * 5 / 8));
    auto make_classifier = [&] { return std::unique_ptr<BatchClassifier>(new BatchClassifier(ctx, b, d, fp_correct, start_sample, fp, out, "%llu read pairs classified\r")); };
    std::unique_ptr<BatchClassifier> classifier = make_classifier();
    FrontEnd fe = on_device ? classify_bgzf_on_device(ctx, fq, 2, b, d, start_sample, qual_offset, *classifier) : kFrontEndNotMine;
    if (fe == kFrontEndRestart) {
        classifier->finish();
        empty_output(out, prefix);
        classifier = make_classifier();
    }
    if (fe != kFrontEndDone)
    stream_fastq_records(fq[0], &fq[1], qual_offset, true, [&](ReadBatch &&piece) {
        if (rb.size() == 0) rb = std::move(piece); else rb.append(piece);
        if (rb.size() >= batch || rb.heavy()) classifier->submit(rb);
    }, [&] { return classifier->spare(); });
    classifier->submit(rb);
    const uint64_t read_count = classifier->finish();
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
    ReadBatch rb;
    BatchClassifier classifier(ctx, b, d, fp_correct, start_sample, fp, out, " %llu reads classified\r");
    // The reference's loop (read_id_mt_pe.rs:450-569), line by line: the first line is the first id (its last character dropped); a
    // later line that holds a '>' anywhere closes the record before it IF that record has any sequence yet (else the line is dropped and
    // the id stays); every other line joins the record's sequence WITH its line end; the end of the file closes the last record
    // whatever it holds.  A record's lines go straight into the batch's bases (round 6: through getline's buffer, a string of the
    // record's own and then the batch, a 10 kb read was copied three times and scanned once: 125-140 ms per 150 Mbases, the GPU's 7).
    std::string id;
    uint64_t count = 0;
    size_t rec_start = 0;   // where the open record's sequence begins in rb.bases
    auto close_record = [&] {
        rb.begin(id);
        rb.seq_off.push_back(rb.bases.size());
        rb.end();
        rec_start = rb.bases.size();
    };
    auto take_line = [&](const char *line, size_t n) {   // n > 0, the line end included when the file has one
        if (count == 0) {
            id.assign(line, n - 1);
        } else if (memchr(line, '>', n) != nullptr) {
            if (rb.bases.size() > rec_start) {
                close_record();
                id.assign(line, n - 1);
                if (rb.size() % batch == 0 || rb.heavy()) { classifier.submit(rb); rec_start = rb.bases.size(); }
            }
        } else {
            rb.bases.insert(rb.bases.end(), reinterpret_cast<const uint8_t *>(line), reinterpret_cast<const uint8_t *>(line) + n);
        }
        ++count;
    };
    const int fd = ::open(fq[0].c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) die("file not found: %s", fq[0].c_str());
    struct stat sb;
    void *map = MAP_FAILED;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map != MAP_FAILED) {   // the file's pages as they lie in the page cache
        (void)madvise(map, (size_t)sb.st_size, MADV_SEQUENTIAL);
        const char *p = static_cast<const char *>(map), *end = p + sb.st_size;
        while (p < end) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            const char *next = nl ? nl + 1 : end;
            take_line(p, (size_t)(next - p));
            p = next;
        }
        munmap(map, (size_t)sb.st_size);
        ::close(fd);
    } else {                   // a pipe, an empty file: line by line
        FILE *f = fdopen(fd, "rb");
        if (!f) die("file not found: %s", fq[0].c_str());
        char *lineptr = nullptr;
        size_t cap = 0;
        ssize_t got;
        setvbuf(f, nullptr, _IOFBF, 8u << 20);
        while ((got = getline(&lineptr, &cap, f)) > 0) take_line(lineptr, (size_t)got);
        free(lineptr);
        fclose(f);
    }
    close_record();
    classifier.submit(rb);
    const uint64_t read_count = classifier.finish();
    fclose(out);
    fprintf(stderr, "Classified %llu reads in %ld seconds\n", (unsigned long long)read_count, secs_since(t0));
    print_read_id_timing(t0);
}

}  // namespace colorid
