// Internals shared by the two halves of the CLI's workload drivers (drivers.cpp: build-side counting, search, perfect search;
// drivers_readid.cpp: read_id): timing helpers, the group the drivers run on, and the FASTQ record pipeline's batch types.
#pragma once
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
inline long secs_since(Clock::time_point t0) { return (long)std::chrono::duration_cast<std::chrono::seconds>(Clock::now() - t0).count(); }
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
// COLORID_TIMING=1: sub-second phase times on stderr (the reference's own timers print whole seconds)
extern bool g_timing;
// several GPUs (set_group / set_stripes, drivers.cpp): NULL = one GPU
extern cid_group *g_group;
extern std::vector<cid_index *> g_replicas;   // one handle per rank: replicas of the index, or (g_striped) its colour stripes
extern bool g_striped;

// qual_mask (seq.rs:36-56) applied while the read is appended to a batch: q == 0 keeps the sequence as it is; otherwise the output
// has one base per quality character, 'N' where the quality is below q + 33
inline void append_masked(std::vector<uint8_t> &bases, const char *seq, size_t slen, const char *qual, size_t qlen, uint8_t q) {
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
    // long reads: a batch also closes once it holds this many bases (batch boundaries never change a read's result).  48 MiB: 150 Mbases of
    // 10 kb reads are 15 000 reads — fewer than `-c` — and went up as ONE batch behind the whole file's reading (256 MiB until round 6:
    // 125-140 ms per 150 Mbases of FASTA); in pieces the upload and the GPU's 2 ms per piece run beside the reading of the next
    bool heavy() const { return bases.size() >= (48u << 20); }
    void clear() { id_chars.clear(); id_off.clear(); bases.clear(); seq_off.assign(1, 0); read_seq0.assign(1, 0); }
};

// ---- FASTQ text -> packed batches on several threads.  RecordChunker cuts each input's decoded blocks at record boundaries (a
// newline scan on the calling thread); the records of a chunk — for pairs: as many records of either file's current chunk as
// both have — are split into lines, quality-masked (seq.rs:36-56) and packed by COLORID_PARSE_THREADS threads; the
// pieces reach `sink` in input order.  The same reads in the same order as the line loops of read_id_mt_pe.rs:862-895 / :927-975
// and kmer.rs:481-503 / :619-647: a record is pushed at its fourth line; for pairs the walk ends with the shorter file.
// (default: a fifth of cpu_budget(), at most 4 — 3 on a 16-CPU share of a GPU box)
const int g_parse_threads = [] { const char *e = cli_env("COLORID_PARSE_THREADS"); const int v = e ? atoi(e) : std::min(4, cpu_budget() / 5); return v < 1 ? 1 : v; }();

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
inline ReadBatch pack_records(ReadBatch rb, const RecChunk *c1, size_t a0, const RecChunk *c2, size_t b0, size_t n, uint8_t q, bool want_ids) {
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

}  // namespace colorid
