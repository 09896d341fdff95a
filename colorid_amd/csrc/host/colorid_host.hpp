// Host side of colorid's query path, C++17, above the C ABI (include/colorid_hip.h).
// The reference host is Rust (no toolchain in this image); names and argument meaning follow the reference
// functions cited at each declaration so the call sites read like src/main.rs.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <thread>
#include <mutex>
#include <functional>
#include <deque>
#include <condition_variable>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/colorid_hip.h"

namespace colorid {

[[noreturn]] void die(const char *fmt, ...);  // the reference panics (expect/unwrap): message + exit code 101

// ---------------------------------------------------------------- seq.rs / kmer.rs (query side)
std::vector<std::string> read_fasta(const std::string &path);                                          // kmer.rs:10-45
void read_fasta_mf(const std::string &path, std::vector<std::string> &labels, std::vector<std::string> &seqs);  // kmer.rs:47-84
void qual_mask(std::string &seq, const std::string &qual, uint8_t q);                                  // seq.rs:36-56 (in place)

// FnvHashMap<String, usize> of canonical k-mers: packed keys (n * k bytes) + counts.  Iteration order is
// insertion order (the reference's is arbitrary).
class KmerMap {
  public:
    explicit KmerMap(uint32_t k);
    uint32_t k() const { return k_; }
    size_t size() const { return counts_.size(); }
    const uint8_t *keys() const { return keys_.data(); }
    const std::vector<uint32_t> &counts() const { return counts_; }
    void add(const uint8_t *key, uint32_t n = 1);
    void clean(uint64_t t);          // kmer.rs:826-837: keep count > t
    int64_t auto_cutoff() const;     // kmer.rs:866-942; -1 where the reference would panic
  private:
    void rehash();
    uint32_t k_;
    std::vector<uint8_t> keys_;
    std::vector<uint32_t> counts_;
    std::vector<uint32_t> table_;  // entry + 1
};
// kmer.rs:866-942 on the histogram {multiplicity -> number of distinct k-mers}; -1 where the reference would panic
int64_t auto_cutoff_from_histogram(const std::map<uint64_t, uint64_t> &histo, uint64_t n_distinct);
void kmerize_vector(const std::vector<std::string> &v, size_t d, KmerMap &out);   // kmer.rs:87-125
bool kmerize_string(const std::string &l, KmerMap &out);                          // kmer.rs:271-299 (false = None)
void kmers_from_fq_qual(const std::string &path, uint8_t q, KmerMap &out);        // kmer.rs:461-510
void kmers_fq_pe_qual(const std::string &p1, const std::string &p2, uint8_t q, KmerMap &out);  // kmer.rs:581-655

// CPUs this process may keep busy: the cgroup's CPU quota (cpu.max) when there is one, else the hardware's thread count.  The
// host pipeline sizes its pools from it (a third for inflating block-gzip members, a fifth for packing records): more runnable threads than the quota get the whole process throttled for the rest of the scheduling period, which on a
// 16-CPU share of a GPU box cost more than the extra threads brought (tools/exp_readid_stages.sh).
int cpu_budget();

// A few persistent worker threads.  The pipeline used to start threads per batch (inflating a batch's members, packing a chunk's
// records, polling a batch's slices): a thread's start and end map and unmap its stack, and in a process whose other threads fault
// pages all the time those calls queue on the address-space lock — two poll threads per batch were twice as SLOW as one.
// The command line's environment switches (cli_switches.def lists them; README.md's table is generated from it): cli_env("NAME") returns
// the variable's value as it was when the first switch was asked for — one snapshot of the listed names, whoever asks — or nullptr.
inline const char *cli_env(const char *name) {
    struct Snapshot {
        std::vector<std::pair<std::string, std::string>> set;   // the listed variables that are set
        std::vector<std::string> known;
        Snapshot() {
#define CLI_SWITCH(id, env, kind, dflt, doc) known.push_back(env); if (const char *e = getenv(env)) set.emplace_back(env, e);
#include "cli_switches.def"
#undef CLI_SWITCH
        }
    };
    static const Snapshot snap;
    for (const auto &kv : snap.set) if (kv.first == name) return kv.second.c_str();
    for (const auto &k : snap.known) if (k == name) return nullptr;
    fprintf(stderr, "cli_env: '%s' is not in cli_switches.def\n", name);
    abort();
}

class TaskPool {
  public:
    explicit TaskPool(int n_threads) {
        for (int i = 0; i < n_threads; ++i) threads_.emplace_back([this] { run(); });
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    TaskPool(const TaskPool &) = delete;
    TaskPool &operator=(const TaskPool &) = delete;
    void submit(std::function<void()> fn) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(fn)); }
        cv_.notify_one();
    }
    // fn(0) .. fn(n-1), the caller taking part; returns when all are done
    void parallel_for(size_t n, const std::function<void(size_t)> &fn) {
        if (n == 0) return;
        struct Latch { std::mutex m; std::condition_variable c; size_t left; } latch;
        latch.left = n - 1;
        for (size_t i = 1; i < n; ++i)
            submit([&fn, &latch, i] {
                fn(i);
                std::lock_guard<std::mutex> lk(latch.m);
                if (--latch.left == 0) latch.c.notify_one();
            });
        fn(0);
        std::unique_lock<std::mutex> lk(latch.m);
        latch.c.wait(lk, [&] { return latch.left == 0; });
    }
  private:
    void run() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

// gz/plain line reader with BufRead::lines() semantics (strips \n and \r\n)
class LineReader {   // inflates on its own thread, a few MiB ahead of the caller (two mates of a pair decode in parallel)
  public:
    explicit LineReader(const std::string &path);
    ~LineReader();
    // Start inflating `path` now, up to ~256 MiB of text ahead: the next LineReader opened on the same path takes the stream
    // over.  The CLI calls this before it loads the index, so that gzip decoding runs beside the index load instead of after it.
    static void prefetch(const std::string &path);
    static void drop_prefetched();   // end of a command: stop the streams nobody took
    static double blocked_ms();      // COLORID_TIMING: time the decoding threads waited for their consumer (all readers so far)
    // Block-gzip (BGZF) input is inflated on GPU `device` (cid_bgzf_inflate: one wave per member) instead of on COLORID_GZ_THREADS
    // host threads; < 0 = on the host.  Applies to readers opened afterwards; each reader thread uses a context of its own.
    static void inflate_on_gpu(int device);
    LineReader(const LineReader &) = delete;
    LineReader &operator=(const LineReader &) = delete;
    bool next(std::string &line);
    // the same line as a view: valid until the next call on this reader (it points into the decoded block, or into the reader's
    // own buffer for a line that straddles two blocks) — no per-line copy
    bool next(const char *&ptr, size_t &len);
    // Block interface, for a caller that splits the text itself (RecordChunker; not to be mixed with next() on one reader): the
    // next decoded block, its text at [kHeadroom, blk.size()) — the bytes in front are the caller's to fill (the tail of the
    // previous block).  Blocks go back with recycle().
    static constexpr size_t kHeadroom = 1u << 16;
    bool next_block(std::vector<char> &blk);
    void recycle(std::vector<char> &&blk);
    struct Impl;   // (defined in fastx_kmers.cpp)
  private:
    Impl *p_;
};

// Whole FASTQ records (groups of four lines, counted from the start of the file as the reference's `line_count % 4` does) of one
// Block-gzip (BGZF) input for the device front end (cid_fastq_*): the file's members still COMPRESSED, in stretches worth about
// `text_target` bytes of text — the reading thread only walks the members' headers (their sizes are in the "BC" extra field, the
// text sizes in the trailers) a stretch or two ahead of the caller; inflate, line split, record split and masking happen on the GPU.
struct ByteBuf {   // a growable byte buffer that never zero-fills what it is about to overwrite (fread targets of tens of MB)
    unsigned char *p = nullptr;
    size_t n = 0, cap = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf &) = delete;
    ByteBuf &operator=(const ByteBuf &) = delete;
    ByteBuf(ByteBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ByteBuf &operator=(ByteBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    ~ByteBuf() { free(p); }
    void reserve(size_t want) {
        if (want <= cap) return;
        size_t c = cap ? cap : (size_t)1 << 20;
        while (c < want) c += c / 2;
        unsigned char *q = static_cast<unsigned char *>(realloc(p, c));
        if (!q) die("out of memory (%zu bytes)", c);
        p = q; cap = c;
    }
    unsigned char *data() { return p; }
    const unsigned char *data() const { return p; }
    size_t size() const { return n; }
};
struct PinnedBuf {   // page-locked bytes from the library (cid_pinned_alloc; plain malloc when that fails), grown, never shrunk
    unsigned char *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    PinnedBuf(PinnedBuf &&o) noexcept : p(o.p), cap(o.cap), pinned(o.pinned) { o.p = nullptr; o.cap = 0; }
    PinnedBuf &operator=(PinnedBuf &&o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; pinned = o.pinned; o.p = nullptr; o.cap = 0; } return *this; }
    ~PinnedBuf() { release(); }
    void release();
    void reserve(size_t want);   // (contents are not kept)
};
struct StretchBuf {   // a stretch's compressed bytes: page-locked when the library hands such memory out, grown with its contents kept
    unsigned char *p = nullptr;
    size_t n = 0, cap = 0;
    bool pinned = false;
    StretchBuf() = default;
    StretchBuf(const StretchBuf &) = delete;
    StretchBuf &operator=(const StretchBuf &) = delete;
    StretchBuf(StretchBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), pinned(o.pinned) { o.p = nullptr; o.n = o.cap = 0; }
    StretchBuf &operator=(StretchBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; pinned = o.pinned; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    ~StretchBuf() { release(); }
    void release();
    void reserve(size_t want);   // (the first n bytes are kept)
    unsigned char *data() { return p; }
    const unsigned char *data() const { return p; }
    size_t size() const { return n; }
};
struct BgzfStretch {
    // Whole members, back to back.  COLORID_DEVICE_FASTQ_PINNED=1 puts them in page-locked memory, sized once per stretch (from the file's
    // size, then from the stretch before), and their copy to the device then runs beside the loop (CID_FASTQ_KEEP): the push of 16 M
    // reads falls from 150-190 ms to 14-36 — and the loop stays where it was, 296-310 ms against 307-315, because the GPU is what it
    // waits for by then (k_bgzf_inflate_wave 10 ms and k_readid 9.6 ms per 256 MiB stretch, side by side 13.3: rocprofv3 of the command
    // line, profiles/r06_frontend_gpu_bound.txt), while two files of pairs start 25-35 ms later behind their larger first buffers
    // (4 M pairs 139-142 ms against 97-104).  Off by default.
    StretchBuf bytes;
    std::vector<uint32_t> off, len, text_len;      // member i = bytes[off[i], +len[i]), its text has text_len[i] bytes
    uint64_t text_bytes = 0;
    bool last = false;                             // the file ends with this stretch
    // the members [device_members, off.size()) are inflated already, by the reader's host threads: their text, back to back
    size_t device_members = 0;
    PinnedBuf host_text;
    size_t host_text_bytes = 0;
};
class BgzfMemberReader {
  public:
    // host_share: the fraction of every stretch's text that `host_threads` threads of the reader inflate themselves (0: none)
    BgzfMemberReader(const std::string &path, size_t text_target, double host_share = 0.0, int host_threads = 0);
    ~BgzfMemberReader();
    BgzfMemberReader(const BgzfMemberReader &) = delete;
    BgzfMemberReader &operator=(const BgzfMemberReader &) = delete;
    bool next(BgzfStretch &s);                     // false after the stretch that had last == true; `s`'s buffers are recycled
    static bool is_bgzf(const std::string &path);
    // Start reading `path` now (the CLI calls this before the GPU context and the index exist); open() hands the running reader over
    // — or starts one — to whoever classifies the file.  drop_prefetched(): readers nobody took.
    static void prefetch(const std::string &path, size_t text_target, double host_share, int host_threads);
    static std::unique_ptr<BgzfMemberReader> open(const std::string &path, size_t text_target, double host_share, int host_threads);
    static void drop_prefetched();
    size_t text_target() const;
    struct Impl;
  private:
    Impl *p_;
};

// input, a decoded block at a time: the text [begin, rec_end.back()) of buf, record r ending just past its fourth newline at
// rec_end[r].  Lines that do not complete a record at the end of the input are dropped, as the line loops never push them
// (read_id_mt_pe.rs:862-895).  Finding the boundaries is one memchr per line; the records of a chunk are then parsed in parallel.
struct RecChunk {
    std::vector<char> buf;
    size_t begin = 0;
    std::vector<uint32_t> rec_end;
    size_t records() const { return rec_end.size(); }
    size_t rec_begin(size_t r) const { return r ? rec_end[r - 1] : begin; }
};
class RecordChunker {
  public:
    explicit RecordChunker(LineReader &r) : r_(r) {}
    bool next(RecChunk &c);              // false at the end of the input
    void recycle(RecChunk &c) { if (!c.buf.empty()) r_.recycle(std::move(c.buf)); c.buf = std::vector<char>(); c.rec_end.clear(); }
  private:
    LineReader &r_;
    std::string carry_;                  // the lines after the last whole record of the previous block
    uint64_t carry_lines_ = 0;           // newlines inside carry_
    bool done_ = false;
};

void debug_records(const std::string &f1, const std::string *f2, uint8_t q);   // CPU tests: the reads as the record pipeline packs them

// GPU k-mer maps (cid_kmerset, k <= 32; COLORID_HOST_KMERS=1 forces the host map).  count_fastq_gpu returns nullptr when
// the file holds lower-case bases (their case is kept, so they cannot be packed): the caller counts on the host.
bool gpu_counting_enabled(uint64_t k);
// `target`: the index the set is going to be searched in (cid_kmerset_set_target_index: ordered for it at no extra pass); null = code order
cid_kmerset *count_fasta_gpu(cid_ctx *ctx, uint64_t k, const std::vector<std::string> &seqs, const cid_index *target = nullptr);
cid_kmerset *count_fastq_gpu(cid_ctx *ctx, uint64_t k, const std::string &f1, const std::string *f2, uint8_t q, const cid_index *target = nullptr);
int64_t auto_cutoff_gpu(cid_kmerset *ks);   // kmer.rs:866-942 from the device histogram

// ---------------------------------------------------------------- bigsi.rs
struct Bigsi {  // BigsyMapNew minus the map, which lives on the device
    uint64_t bloom_size = 0, num_hash = 0, k_size = 0;
    uint64_t m_size = 0;                    // > 0: BigsyMapMiniNew (.mxi), Bloom keys are minimizers of this length
    std::vector<std::string> colors;        // colour id -> accession
    std::vector<uint64_t> n_ref_kmers;      // by colour id
    cid_index *index = nullptr;
};
// stripes != nullptr: the index goes to the ranks of `group` as colour stripes (cid_group_stripes_*) instead of to one GPU; b.index stays null
Bigsi read_bigsi(cid_ctx *ctx, const std::string &path, int hash_variant, bool meta_only = false, cid_group *group = nullptr,
                 std::vector<cid_index *> *stripes = nullptr);  // bigsi.rs:59-69
// map the index file and start touching its pages — call it before the GPU context is made; read_bigsi then uploads from the mapping
void bigsi_read_ahead(const std::string &path);
void save_bigsi(const std::string &path, const Bigsi &b);                                           // bigsi.rs:51-57
Bigsi build_single(cid_ctx *ctx, const std::string &ref_tsv, uint64_t bloom, uint64_t hashes, uint64_t k, uint8_t quality,
                   int64_t cutoff, int hash_variant, uint64_t m_size = 0);   // build.rs:15-130; m_size > 0: build_single_mini :396-492

// For every hash variant v and every accession of ref_tsv that is a colour of b: the fraction of the accession's k-mers (counted
// as build does, build.rs:54-99) whose n rows are all set in its colour.  Prints one line per (variant, accession); worst[v] = the
// smallest fraction under variant v.  Returns the number of accessions checked.
// build.rs:15-31: `name \t file [\t file2]` per line, a later line of the same name replaces the earlier one (the sample sheet of
// `build -r`, `hashcheck -r` and `batch_id -q`); iterated in name order
std::map<std::string, std::vector<std::string>> tab_to_map(const std::string &tsv);
size_t hashcheck(cid_ctx *ctx, Bigsi &b, const std::string &ref_tsv, uint8_t quality, const char *const *variant_names, std::vector<double> &worst);

// ---------------------------------------------------------------- reports.rs / read_id tail
double false_prob(double m, double k, double n);                                                     // read_id_mt_pe.rs:695-698
struct Classification { std::string label; uint64_t count; uint64_t kmer_length; const char *verdict; uint64_t n_top; };
// read_id_mt_pe.rs:187-251 on the report's non-zero entries (ascending colour id; colour C = no_hits_num)
Classification kmer_poll_plus(const uint32_t *colours, const uint32_t *counts, size_t n_entries, uint64_t kmer_length, const Bigsi &b,
                              const std::vector<double> &fp, double fp_correct);
void read_counts_five_fields(const std::string &reads_file, const std::string &prefix);            // reports.rs:98-120

// ---------------------------------------------------------------- several GPUs (--gpus N / --devices a,b,..)
// When set, the drivers below shard every query over the group's ranks (cid_group_*); results are identical.
void set_group(cid_group *group, const std::vector<cid_index *> &replicas);
void set_stripes(cid_group *group, const std::vector<cid_index *> &stripes);   // the same calls over a colour-striped index

// ---------------------------------------------------------------- drivers (same names as the reference modules)
namespace perfect_search {
void batch_search(cid_ctx *, const std::vector<std::string> &files, const Bigsi &b);      // perfect_search.rs:6-60
void batch_search_mf(cid_ctx *, const std::vector<std::string> &files, const Bigsi &b);   // perfect_search.rs:62-120
}
namespace batch_search_pe {
void batch_search(cid_ctx *, const std::vector<std::string> &files1, const std::vector<std::string> &files2, const Bigsi &b,
                  int64_t filter, double cov, bool gene_search, uint8_t qual_offset);     // batch_search_pe.rs:9-179
}
namespace read_id_mt_pe {
// block-gzip (BGZF) fastq input on one GPU takes the device front end (cid_fastq_*) unless COLORID_DEVICE_FASTQ=0
bool device_fastq_wanted(const std::vector<std::string> &fq, size_t n_files);
size_t device_fastq_stretch_bytes(size_t n_colors);   // text per stretch (n_colors == 0: before the index is known)
double device_fastq_host_share();                     // share of a stretch's text inflated by the reader's host threads
int device_fastq_host_threads(size_t n_files);
void per_read_stream_se(cid_ctx *, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct, size_t batch,
                        const std::string &prefix, uint8_t qual_offset, size_t start_sample);   // read_id_mt_pe.rs:835-951
void per_read_stream_pe(cid_ctx *, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct, size_t batch,
                        const std::string &prefix, uint8_t qual_offset, size_t start_sample);   // read_id_mt_pe.rs:701-832
void stream_fasta(cid_ctx *, const std::vector<std::string> &fq, const Bigsi &b, size_t d, double fp_correct, size_t batch,
                  const std::string &prefix, size_t start_sample);                              // read_id_mt_pe.rs:450-569
}

}  // namespace colorid
