// Sequence input and canonical k-mer counting on the host (src/seq.rs, the query-side functions of src/kmer.rs).
// This is host staging for the GPU path: the reference counts k-mers on the CPU too (String keys + FNV);
// here keys live packed in one arena and are hashed 8 bytes at a time.
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <cerrno>
#include <dlfcn.h>
#include <zlib.h>

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <atomic>
#include <chrono>

#include "colorid_host.hpp"
#include "fast_inflate.hpp"
#include "par_gunzip.hpp"

namespace colorid {

void die(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "colorid: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    // a Rust panic exits with 101.  Reader / classifier threads may be running (and may be the caller): flush what was written
    // and leave without running static destructors under them.
    fflush(nullptr);
    _exit(101);
}

// ---------------------------------------------------------------------------------------------- files

static std::string slurp(const std::string &path, const char *what) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) die("%s: %s", what, path.c_str());
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// str::lines(): split at \n, drop one trailing \r, no empty line after a final \n
template <typename F>
static void for_each_line(const std::string &c, F &&fn) {
    size_t pos = 0, n = c.size();
    while (pos < n) {
        size_t e = c.find('\n', pos);
        size_t next = e == std::string::npos ? n : e + 1;
        if (e == std::string::npos) e = n;
        size_t le = e;
        if (le > pos && c[le - 1] == '\r') --le;
        fn(c.data() + pos, le - pos);
        pos = next;
    }
}

static void read_fasta_impl(const std::string &path, std::vector<std::string> *labels, std::vector<std::string> &seqs) {
    const std::string c = slurp(path, "file not found");
    size_t n_lines = 0;
    for_each_line(c, [&](const char *, size_t) { ++n_lines; });
    std::string sub;
    size_t count = 0;
    for_each_line(c, [&](const char *p, size_t len) {
        ++count;
        if (memchr(p, '>', len)) {  // any line containing '>' is a header (kmer.rs:26)
            if (labels) labels->emplace_back(len ? p + 1 : p, len ? len - 1 : 0);  // line[1..]
            if (!sub.empty()) seqs.push_back(sub);
            sub.clear();
        } else if (count == n_lines) {
            sub.append(p, len);
            if (!sub.empty()) seqs.push_back(sub);
        } else {
            sub.append(p, len);
        }
    });
}

std::vector<std::string> read_fasta(const std::string &path) {
    std::vector<std::string> v;
    read_fasta_impl(path, nullptr, v);
    return v;
}
void read_fasta_mf(const std::string &path, std::vector<std::string> &labels, std::vector<std::string> &seqs) {
    read_fasta_impl(path, &labels, seqs);
}

int cpu_budget() {
    static const int budget = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");   // cgroup v2: "<quota> <period>" in microseconds, or "max <period>"
        if (f) {
            char q[32];
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long cpus = (atol(q) + period - 1) / period;
                if (cpus >= 1 && cpus < hw) hw = (int)cpus;
            }
            fclose(f);
        }
        return hw;
    }();
    return budget;
}

// libdeflate (whole-buffer DEFLATE: a BGZF member is one) inflates 2-3x faster than zlib; it is used when the host has it
// (dlopen of libdeflate.so.0 — no build-time dependency), checks CRC-32 and ISIZE like zlib, and COLORID_LIBDEFLATE=0 turns it off
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*gzip)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        const char *e = cli_env("COLORID_LIBDEFLATE");
        if (e && atoi(e) == 0) return;
        void *lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        alloc = reinterpret_cast<void *(*)()>(dlsym(lib, "libdeflate_alloc_decompressor"));
        gzip = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(lib, "libdeflate_gzip_decompress"));
        release = reinterpret_cast<void (*)(void *)>(dlsym(lib, "libdeflate_free_decompressor"));
        crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(dlsym(lib, "libdeflate_crc32"));
        ok = alloc && gzip && release && crc32;
    }
};
static const LibDeflate &libdeflate() { static const LibDeflate l; return l; }

// Decoded text travels in blocks through a small bounded queue: zlib runs on the reader's own thread while the caller
// splits lines, masks qualities and feeds the GPU.  gzread() continues across gzip members like MultiGzDecoder and reads
// plain files as they are.
// BGZF input (block gzip: every member <= 64 KiB and its compressed size in a "BC" extra field — bgzip, htslib, Illumina's
// converters): the members of a batch are independent, so the reader thread only walks the headers and hands the batch's members to
// COLORID_GZ_THREADS (default 8) inflating threads, each writing at its member's offset of the output block (the members'
// uncompressed sizes are in their trailers).  Single-stream gzip has no such boundaries and stays on one zlib thread.
// COLORID_TIMING: how long the decoding threads of all readers stood still because their consumer had not taken the blocks before
static std::atomic<uint64_t> g_reader_blocked_us{0};
static std::atomic<int> g_gzip_streams{0};   // gzip-stream readers alive: the mates of a pair share the threads their streams are decoded on
double LineReader::blocked_ms() { return (double)g_reader_blocked_us.load() / 1e3; }

struct LineReader::Impl {
    static constexpr size_t kBlock = 4u << 20, kHead = LineReader::kHeadroom;   // a block's text starts at kHead
    size_t depth = 4;              // blocks in flight (prefetched streams: ~256 MiB worth)
    gzFile gz = nullptr;
    FILE *raw = nullptr;           // BGZF mode: the compressed file itself
    int gz_threads = std::min(8, std::max(1, cpu_budget() / 3));   // COLORID_GZ_THREADS overrides
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_full, cv_free;
    std::deque<std::vector<char>> full, free_blocks;
    bool eof = false, stop = false;
    std::vector<char> cur;   // block being split by next()
    size_t pos = 0;
    std::string carry;       // next(ptr, len): a line that straddles blocks

    // a block-gzip member header: 1f 8b 08 04 | mtime xfl os | XLEN | ... 'B' 'C' 02 00 BSIZE ... ; returns the member's total size or 0
    static size_t bgzf_member_size(const unsigned char *h, size_t have) {
        if (have < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return 0;
        const size_t xlen = h[10] | ((size_t)h[11] << 8);
        if (have < 12 + xlen) return 0;
        for (size_t o = 12; o + 4 <= 12 + xlen;) {
            const size_t slen = h[o + 2] | ((size_t)h[o + 3] << 8);
            if (h[o] == 'B' && h[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return (size_t)(h[o + 4] | ((size_t)h[o + 5] << 8)) + 1;
            o += 4 + slen;
        }
        return 0;
    }
    static bool is_gzip(const std::string &path) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        unsigned char h[3] = {0, 0, 0};
        const size_t n = fread(h, 1, 3, f);
        fclose(f);
        return n == 3 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8;
    }
    static bool is_bgzf(const std::string &path) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        unsigned char h[64];
        const size_t n = fread(h, 1, sizeof h, f);
        fclose(f);
        return bgzf_member_size(h, n) >= 26;
    }

    void push(std::vector<char> &&blk, bool last) {
        std::lock_guard<std::mutex> lk(mu);
        if (blk.size() > kHead) full.push_back(std::move(blk));
        if (last) eof = true;
        cv_full.notify_one();
    }
    bool take_free(std::vector<char> &blk) {   // false: asked to stop
        std::unique_lock<std::mutex> lk(mu);
        const auto tw = std::chrono::steady_clock::now();
        cv_free.wait(lk, [&] { return stop || full.size() < depth; });
        g_reader_blocked_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw).count();
        if (stop) return false;
        if (!free_blocks.empty()) { blk = std::move(free_blocks.front()); free_blocks.pop_front(); }
        return true;
    }

    void run() {
        for (;;) {
            std::vector<char> blk;
            if (!take_free(blk)) return;
            blk.resize(kHead + kBlock);
            size_t got = 0;
            while (got < kBlock) {   // gzread may return short counts at member boundaries
                const int n = gzread(gz, blk.data() + kHead + got, (unsigned)(kBlock - got));
                if (n <= 0) break;
                got += (size_t)n;
            }
            blk.resize(kHead + got);
            const bool last = got < kBlock;
            push(std::move(blk), last);
            if (last) return;
        }
    }

    // Single-stream (or multi-member) gzip when the host has libdeflate: the members' DEFLATE data goes through zlib's raw inflate and
    // the CRC-32 of the text through libdeflate's (zlib 1.2.11 computes it at 0.9 GB/s inside gzread, a third of the stream's decoding
    // time; libdeflate's runs at memory speed).  The container is parsed here (RFC 1952: header with optional extra / name / comment /
    // header CRC, trailer CRC-32 + ISIZE, both checked); further members follow as in MultiGzDecoder; bytes after the last member that
    // are no gzip header end the stream, as with gzread.
    struct RawIn {
        FILE *f;
        std::vector<unsigned char> buf;
        size_t pos = 0, end = 0;
        bool eof = false;
        static constexpr size_t kPad = 64;   // readable bytes behind the data (FastInflate loads eight at a time)
        explicit RawIn(FILE *file) : f(file), buf((1u << 20) + kPad) {}
        size_t avail() const { return end - pos; }
        bool fill() {   // more bytes behind the unread ones; false at the end of the file
            if (eof) return false;
            if (pos && pos == end) pos = end = 0;
            if (end == buf.size() - kPad) {
                if (pos == 0) return false;
                memmove(buf.data(), buf.data() + pos, end - pos);
                end -= pos; pos = 0;
            }
            const size_t n = fread(buf.data() + end, 1, buf.size() - kPad - end, f);
            if (n == 0) { eof = true; return false; }
            end += n;
            memset(buf.data() + end, 0, kPad);
            return true;
        }
        bool need(size_t n) { while (avail() < n) if (!fill()) return false; return true; }
        int byte() { if (!need(1)) return -1; return buf[pos++]; }
        bool skip(size_t n) { while (n) { if (!need(1)) return false; const size_t k = std::min(n, avail()); pos += k; n -= k; } return true; }
    };
    void run_gzip() {
        const LibDeflate &ld = libdeflate();
        RawIn in(raw);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) die("zlib: inflateInit2 failed");
        std::vector<char> blk;
        size_t got = 0;
        bool have_blk = false, first = true;
        const bool use_fast = !(cli_env("COLORID_FAST_INFLATE") && atoi(cli_env("COLORID_FAST_INFLATE")) == 0);   // 0: zlib's inflate
        FastInflate fz;
        // COLORID_PAR_GZIP=0: one thread decodes a gzip stream (FastInflate) as before
        // threads: COLORID_GZ_THREADS, else as many as the process has CPUs (at most 16), shared between the gzip streams open at once (the
        // mates of a pair) — they run in lockstep with idle stretches, beside the packing and polling threads: measured on a 16-CPU share,
        // 4 M reads with forty quality letters, 0.64-0.67 s on 8 threads, 0.50-0.57 s on 12, 0.46-0.48 s on 16 (serial 1.8-1.9 s, zlib
        // 2.8-2.95 s); 4 M pairs 0.67-0.68 s on 8 + 8, 1.18-1.30 s on 4 + 4
        struct Alive { ~Alive() { --g_gzip_streams; } } alive;   // (counted in at open_stream)
        int par_threads = cli_env("COLORID_GZ_THREADS") ? gz_threads : std::min(16, std::max(gz_threads, cpu_budget()));
        bool par_on = use_fast && par_threads >= 3 && !(cli_env("COLORID_PAR_GZIP") && atoi(cli_env("COLORID_PAR_GZIP")) == 0);
        int par_small = 0;
        std::unique_ptr<TaskPool> par_pool;
        std::vector<char> hist(32768);
        size_t hist_n = 0;
        // With the fast decoder the CRC-32 of the text (a tenth of the decoding thread's time even at libdeflate's speed) runs on a thread of
        // its own: a block goes to it with the list of its members' pieces, and from there to the consumer.  As before, a member's blocks
        // are handed on as they fill; the mismatch is reported when the member's last piece has been summed.
        struct CrcPiece { size_t off, len; bool ends_member; uint32_t want; };
        struct CrcJob { std::vector<char> blk; std::vector<CrcPiece> pieces; bool last = false, stop = false; };
        std::mutex cmu;
        std::condition_variable ccv_job, ccv_room;
        std::deque<CrcJob> cq;
        std::vector<CrcPiece> pieces;   // of the block being filled
        size_t piece_start = 0;         // where the current member's text starts in it
        std::thread crc_thread;
        if (use_fast)
            crc_thread = std::thread([&] {
                uint32_t c = 0;
                for (;;) {
                    CrcJob job;
                    {
                        std::unique_lock<std::mutex> lk(cmu);
                        ccv_job.wait(lk, [&] { return !cq.empty(); });
                        job = std::move(cq.front());
                        cq.pop_front();
                    }
                    ccv_room.notify_one();
                    if (job.stop) return;
                    for (const CrcPiece &pc : job.pieces) {
                        c = ld.crc32(c, job.blk.data() + kHead + pc.off, pc.len);
                        if (pc.ends_member) {
                            if (c != pc.want) die("corrupt gzip member (inflate / CRC-32 failed)");
                            c = 0;
                        }
                    }
                    push(std::move(job.blk), job.last);
                }
            });
        struct CrcJoin {   // (every way out of this function ends the CRC thread first)
            std::thread &t; std::mutex &mu; std::condition_variable &cv; std::deque<CrcJob> &q;
            ~CrcJoin() {
                if (!t.joinable()) return;
                { std::lock_guard<std::mutex> lk(mu); CrcJob j; j.stop = true; q.push_back(std::move(j)); }
                cv.notify_one();
                t.join();
            }
        } crc_join{crc_thread, cmu, ccv_job, cq};
        auto flush_block = [&](bool last) {
            blk.resize(kHead + got);
            if (use_fast) {
                CrcJob job;
                job.blk = std::move(blk);
                job.pieces.swap(pieces);
                job.last = last;
                {
                    std::unique_lock<std::mutex> lk(cmu);
                    ccv_room.wait(lk, [&] { return cq.size() < 2; });
                    cq.push_back(std::move(job));
                }
                ccv_job.notify_one();
            } else {
                push(std::move(blk), last);
            }
            blk = std::vector<char>();
            got = 0;
            piece_start = 0;
            have_blk = false;
        };
        for (;;) {   // one member per round
            if (!in.need(1)) break;                                    // clean end of the file
            if (!in.need(10) || in.buf[in.pos] != 0x1f || in.buf[in.pos + 1] != 0x8b) {
                if (first) die("not a gzip stream");
                break;                                                   // trailing bytes that are no member
            }
            if (in.buf[in.pos + 2] != 8) die("corrupt gzip member (unknown compression method)");
            const unsigned flg = in.buf[in.pos + 3];
            in.pos += 10;
            if (flg & 4) { const int a = in.byte(), b = in.byte(); if (a < 0 || b < 0 || !in.skip((size_t)a | ((size_t)b << 8))) die("truncated gzip member header"); }
            if (flg & 8) { int c; while ((c = in.byte()) > 0) {} if (c < 0) die("truncated gzip member header"); }
            if (flg & 16) { int c; while ((c = in.byte()) > 0) {} if (c < 0) die("truncated gzip member header"); }
            if (flg & 2) { if (!in.skip(2)) die("truncated gzip member header"); }
            first = false;
            inflateReset(&zs);
            uint32_t crc = 0;
            uint64_t total = 0;
            if (use_fast && par_on) {
                // the member's DEFLATE stream on gz_threads threads (par_gunzip.hpp: chunks that find a block boundary of their own and decode
                // from it with the text before them unknown, put right in order afterwards).  The text's CRC-32 stays with the CRC thread.
                if (!par_pool) {   // (the first member's body: both readers of a pair exist by now)
                    if (!cli_env("COLORID_GZ_THREADS")) par_threads = std::max(3, par_threads / std::max(1, g_gzip_streams.load()));
                    par_pool.reset(new TaskPool(par_threads - 1));
                }
                // (COLORID_GZ_CHUNK_KB / COLORID_GZ_CHUNKS_PER_THREAD: the experiment's knobs)
                const size_t chunk_kb = cli_env("COLORID_GZ_CHUNK_KB") ? (size_t)atol(cli_env("COLORID_GZ_CHUNK_KB")) : 1024;
                const size_t per_thread = cli_env("COLORID_GZ_CHUNKS_PER_THREAD") ? (size_t)atol(cli_env("COLORID_GZ_CHUNKS_PER_THREAD")) : 2;
                ParallelInflate pi(chunk_kb << 10, (size_t)par_threads * (per_thread ? per_thread : 1));
                bool stopped = false;
                auto reader = [&](uint8_t *dst, size_t cap) -> size_t { return fread(dst, 1, cap, raw); };
                auto sink = [&](const uint8_t *t, size_t n) -> bool {
                    while (n) {
                        if (!have_blk) {
                            if (!take_free(blk)) { stopped = true; return false; }
                            blk.resize(kHead + kBlock);
                            got = 0;
                            have_blk = true;
                        }
                        const size_t k = std::min(n, kBlock - got);
                        memcpy(blk.data() + kHead + got, t, k);
                        got += k; t += k; n -= k; total += k;
                        if (got == kBlock) {
                            pieces.push_back(CrcPiece{piece_start, got - piece_start, false, 0});
                            flush_block(false);
                        }
                    }
                    return true;
                };
                auto pfor = [&](size_t n, const std::function<void(size_t)> &fn) { par_pool->parallel_for(n, fn); };
                auto no_crc = [](uint32_t, const uint8_t *, size_t) -> uint32_t { return 0; };
                auto no_comb = [](uint32_t, uint32_t, uint64_t) -> uint32_t { return 0; };
                const bool ok = pi.run(in.buf.data() + in.pos, in.avail(), reader, sink, pfor, no_crc, no_comb);
                if (stopped) { inflateEnd(&zs); return; }
                if (!ok) die("corrupt gzip member (inflate failed: %s)", pi.error());
                if (cli_env("COLORID_TIMING"))
                    fprintf(stderr, "timing: gzip member of %llu bytes of text decoded on %d threads: %llu chunks in %llu rounds, %llu started inside the stream and were taken, "
                            "%llu stretches decoded again serially\n", (unsigned long long)pi.total(), par_threads, (unsigned long long)pi.stats.chunks,
                            (unsigned long long)pi.stats.rounds, (unsigned long long)pi.stats.accepted, (unsigned long long)pi.stats.serial);
                // what was read beyond the stream (the trailer, the members behind it) goes back to the reader
                const std::vector<uint8_t> &rest = pi.leftover();
                in.buf.assign(std::max<size_t>(rest.size() + (1u << 20), 2u << 20) + RawIn::kPad, 0);
                if (!rest.empty()) memcpy(in.buf.data(), rest.data(), rest.size());
                in.pos = 0; in.end = rest.size(); in.eof = false;
                hist_n = 0;
                // a file of many small members gains nothing here (each one's chunks start from scratch): the serial decoder takes over
                if (pi.stats.rounds <= 1 && ++par_small >= 4) par_on = false;
            } else if (use_fast) {   // the member's DEFLATE stream through FastInflate (fast_inflate.hpp): 1.6x zlib's inflate on FASTQ text
                fz.reset();
                for (;;) {
                    if (!have_blk) {
                        if (!take_free(blk)) { inflateEnd(&zs); return; }
                        blk.resize(kHead + kBlock);
                        got = 0;
                        have_blk = true;
                        if (hist_n) memcpy(blk.data() + kHead - hist_n, hist.data(), hist_n);   // the member's last 32 KiB in front of the block's text
                    }
                    const uint8_t *ip = in.buf.data() + in.pos;
                    uint8_t *op = reinterpret_cast<uint8_t *>(blk.data() + kHead + got), *const op0 = op;
                    const FastInflate::Result r = fz.run(ip, in.buf.data() + in.end, in.eof, op, reinterpret_cast<uint8_t *>(blk.data() + kHead + kBlock));
                    in.pos = (size_t)(ip - in.buf.data());
                    if (r == FastInflate::kError) die("corrupt gzip member (inflate failed: %s)", fz.error());
                    const size_t made = (size_t)(op - op0);
                    got += made;
                    total += made;
                    if (r == FastInflate::kStreamEnd) break;
                    if (r == FastInflate::kNeedInput) {
                        if (!in.fill() && !in.eof) die("truncated gzip member");
                        continue;
                    }
                    // the block is full (to within a longest match): its last 32 KiB stay with the reader for the next block's matches
                    hist_n = got < hist.size() ? got : hist.size();
                    memcpy(hist.data(), blk.data() + kHead + got - hist_n, hist_n);
                    pieces.push_back(CrcPiece{piece_start, got - piece_start, false, 0});
                    flush_block(false);
                }
                hist_n = 0;   // (the next member's matches cannot reach in front of its own text)
            } else
            for (;;) {   // the member's DEFLATE stream
                if (!have_blk) {
                    if (!take_free(blk)) { inflateEnd(&zs); return; }
                    blk.resize(kHead + kBlock);
                    got = 0;
                    have_blk = true;
                }
                if (in.avail() == 0 && !in.fill()) die("truncated gzip member");
                zs.next_in = in.buf.data() + in.pos;
                zs.avail_in = (uInt)std::min<size_t>(in.avail(), 1u << 30);
                zs.next_out = reinterpret_cast<Bytef *>(blk.data() + kHead + got);
                zs.avail_out = (uInt)(kBlock - got);
                const uInt in0 = zs.avail_in, out0 = zs.avail_out;
                const int rc = inflate(&zs, Z_NO_FLUSH);
                if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) die("corrupt gzip member (inflate failed)");
                in.pos += in0 - zs.avail_in;
                const size_t made = out0 - zs.avail_out;
                crc = ld.crc32(crc, blk.data() + kHead + got, made);
                got += made;
                total += made;
                if (got == kBlock) flush_block(false);
                if (rc == Z_STREAM_END) break;
                if (rc == Z_BUF_ERROR && in0 != 0 && in0 == zs.avail_in && made == 0 && zs.avail_out != 0) die("corrupt gzip member (inflate made no progress)");
            }
            if (!in.need(8)) die("truncated gzip member (no trailer)");
            const unsigned char *t = in.buf.data() + in.pos;
            const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            in.pos += 8;
            if (use_fast) {   // the sum is the CRC thread's to compare: the member's last piece goes with the block it lies in
                if (isize != (uint32_t)total) die("corrupt gzip member (inflate / CRC-32 failed)");
                if (!have_blk) {   // (a member that ended exactly with a block, or an empty one: the piece needs a block to travel in)
                    if (!take_free(blk)) { inflateEnd(&zs); return; }
                    blk.resize(kHead + kBlock);
                    got = 0;
                    piece_start = 0;
                    have_blk = true;
                }
                pieces.push_back(CrcPiece{piece_start, got - piece_start, true, want_crc});
                piece_start = got;
            } else if (want_crc != crc || isize != (uint32_t)total) die("corrupt gzip member (inflate / CRC-32 failed)");
        }
        inflateEnd(&zs);
        if (!have_blk) { blk = std::vector<char>(); got = 0; }
        flush_block(true);
    }

    // BGZF: batches of members worth ~16 MiB of text, inflated by gz_threads threads — or, with a GPU set (inflate_on_gpu), batches of
    // ~64 MiB (a thousand members: two rounds of the 512 the chip decodes at a time) by cid_bgzf_inflate on a context of this thread
    struct Member { size_t in_off, in_len, out_off, out_len; };
    std::unique_ptr<TaskPool> pool;   // the inflating threads of a BGZF reader (created by its worker thread, which is one of them)
    int gpu_device = -1;
    cid_ctx *gpu_ctx = nullptr, *gpu_ctx2 = nullptr;   // two contexts take turns: a batch is decoded and copied back while the next one is read and submitted
    struct Pending { std::vector<char> blk; size_t out_total; bool last; cid_ctx *ctx; };
    void finish_pending(Pending &p) {
        size_t bad = 0;
        if (cid_bgzf_inflate_finish(p.ctx, reinterpret_cast<uint8_t *>(p.blk.data() + kHead), p.out_total, &bad) != CID_OK)
            die("corrupt gzip member (inflate / CRC-32 failed): %s", cid_last_error());
        push(std::move(p.blk), p.last);
    }
    void run_bgzf() {
        if (gpu_device >= 0 && (cid_ctx_create(gpu_device, &gpu_ctx) != CID_OK || cid_ctx_create(gpu_device, &gpu_ctx2) != CID_OK)) {
            fprintf(stderr, "note: no GPU context for the gzip members (%s): inflating on the host\n", cid_last_error());
            if (gpu_ctx) cid_ctx_destroy(gpu_ctx);
            gpu_ctx = gpu_ctx2 = nullptr;
        }
        if (gpu_ctx) (void)cid_warmup(gpu_ctx, CID_WARM_INFLATE);
        const size_t kBatchOut = gpu_ctx ? ((size_t)(cli_env("COLORID_GPU_INFLATE_MB") ? atoi(cli_env("COLORID_GPU_INFLATE_MB")) : 64) << 20) : (16u << 20);
        std::deque<Pending> pending;
        size_t turn = 0;
        std::vector<unsigned char> in;        // compressed bytes of the batch (plus the unread tail of the last fread)
        size_t in_have = 0, in_pos = 0;
        bool file_end = false;
        auto need = [&](size_t bytes) {       // make in[in_pos, in_pos + bytes) available if the file has them (grows, never moves)
            while (in_have - in_pos < bytes && !file_end) {
                if (in.size() < in_have + (4u << 20)) in.resize(in_have + (4u << 20));
                const size_t n = fread(in.data() + in_have, 1, in.size() - in_have, raw);
                if (n == 0) file_end = true;
                in_have += n;
            }
            return in_have - in_pos >= bytes;
        };
        for (;;) {
            std::vector<char> blk;
            if (!take_free(blk)) break;
            std::vector<Member> mem;
            size_t out_total = 0;
            // the unread tail moves to the front once per batch (never inside one: members are addressed by offset)
            if (in_pos) { memmove(in.data(), in.data() + in_pos, in_have - in_pos); in_have -= in_pos; in_pos = 0; }
            size_t scan = in_pos;
            bool last = false;
            while (out_total < kBatchOut) {
                in_pos = scan;
                if (!need(18)) { if (in_have - in_pos != 0) die("truncated gzip member header"); last = true; break; }
                {   // the whole extra field (it may hold other subfields before "BC") before it is parsed
                    const unsigned char *h = in.data() + in_pos;
                    const size_t xlen = h[10] | ((size_t)h[11] << 8);
                    if (h[0] == 0x1f && h[1] == 0x8b && (h[3] & 4) && !need(12 + xlen)) die("truncated gzip member header");
                }
                const size_t msz = bgzf_member_size(in.data() + in_pos, in_have - in_pos);
                if (msz < 26) die("not a block-gzip (BGZF) member inside a BGZF file: mixed gzip streams are not supported in one file");
                if (!need(msz)) die("truncated gzip member");
                const unsigned char *t = in.data() + in_pos + msz - 4;
                const size_t isize = t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                if (isize > (1u << 16)) die("BGZF member larger than 64 KiB");
                mem.push_back(Member{in_pos, msz, out_total, isize});   // (empty members — the BGZF end marker — too: their DEFLATE data and CRC-32 are checked like any other's)
                out_total += isize;
                scan = in_pos + msz;
            }
            in_pos = scan;
            blk.resize(kHead + out_total);
            if (gpu_ctx) {   // submit this batch, then hand over the one submitted before it (its kernel ran meanwhile)
                std::vector<uint32_t> mo(mem.size()), ml(mem.size()), to(mem.size()), tl(mem.size());
                for (size_t i = 0; i < mem.size(); ++i) { mo[i] = (uint32_t)mem[i].in_off; ml[i] = (uint32_t)mem[i].in_len; to[i] = (uint32_t)mem[i].out_off; tl[i] = (uint32_t)mem[i].out_len; }
                cid_ctx *cx = (turn++ & 1) ? gpu_ctx2 : gpu_ctx;
                if (cid_bgzf_inflate_start(cx, in.data(), in_pos, mo.data(), ml.data(), to.data(), tl.data(), mem.size(), out_total) != CID_OK)
                    die("corrupt gzip member (inflate / CRC-32 failed): %s", cid_last_error());
                pending.push_back(Pending{std::move(blk), out_total, last, cx});
                while (pending.size() > (last ? 0u : 1u)) { finish_pending(pending.front()); pending.pop_front(); }
                if (last) break;
                continue;
            } else if (!mem.empty()) {
                const int nt = (int)std::min<size_t>((size_t)gz_threads, mem.size());
                if (!pool && gz_threads > 1) pool.reset(new TaskPool(gz_threads - 1));   // (this thread takes a share too)
                std::vector<int> bad(nt, 0);
                auto work = [&](int t) {
                    const LibDeflate &ld = libdeflate();
                    if (ld.ok) {
                        void *dc = ld.alloc();
                        if (dc) {
                            for (size_t i = (size_t)t; i < mem.size(); i += (size_t)nt) {
                                const Member &m = mem[i];   // exact output size given, no "actual" pointer: anything else than ISIZE bytes is an error
                                if (ld.gzip(dc, in.data() + m.in_off, m.in_len, blk.data() + kHead + m.out_off, m.out_len, nullptr) != 0) { bad[t] = 1; break; }
                            }
                            ld.release(dc);
                            return;
                        }
                    }
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    if (inflateInit2(&zs, 15 + 16) != Z_OK) { bad[t] = 1; return; }
                    for (size_t i = (size_t)t; i < mem.size(); i += (size_t)nt) {
                        const Member &m = mem[i];
                        inflateReset(&zs);
                        zs.next_in = in.data() + m.in_off; zs.avail_in = (uInt)m.in_len;
                        zs.next_out = reinterpret_cast<Bytef *>(blk.data() + kHead + m.out_off); zs.avail_out = (uInt)m.out_len;
                        const int rc = inflate(&zs, Z_FINISH);
                        if (rc != Z_STREAM_END || zs.total_out != m.out_len) { bad[t] = 1; break; }   // zlib checked the member's CRC-32
                    }
                    inflateEnd(&zs);
                };
                if (pool) pool->parallel_for((size_t)nt, [&](size_t t) { work((int)t); });
                else work(0);
                for (int t = 0; t < nt; ++t) if (bad[t]) die("corrupt gzip member (inflate / CRC-32 failed)");
            }
            push(std::move(blk), last);
            if (last) break;
        }
        // (asked to stop with a batch in flight: its context is finished before it is destroyed)
        for (Pending &p : pending) { size_t bad = 0; (void)cid_bgzf_inflate_finish(p.ctx, reinterpret_cast<uint8_t *>(p.blk.data() + kHead), p.out_total, &bad); }
        if (gpu_ctx) { cid_ctx_destroy(gpu_ctx); gpu_ctx = nullptr; }
        if (gpu_ctx2) { cid_ctx_destroy(gpu_ctx2); gpu_ctx2 = nullptr; }
    }
    bool refill() {   // false at end of input
        std::unique_lock<std::mutex> lk(mu);
        if (cur.capacity()) free_blocks.push_back(std::move(cur));
        cv_full.wait(lk, [&] { return eof || !full.empty(); });
        if (full.empty()) { cur = std::vector<char>(); pos = 0; return false; }
        cur = std::move(full.front());
        full.pop_front();
        pos = kHead;
        cv_free.notify_one();
        return true;
    }
    bool take_block(std::vector<char> &blk) {   // the block interface: ownership moves to the caller
        std::unique_lock<std::mutex> lk(mu);
        cv_full.wait(lk, [&] { return eof || !full.empty(); });
        if (full.empty()) return false;
        blk = std::move(full.front());
        full.pop_front();
        cv_free.notify_one();
        return true;
    }
    void give_back(std::vector<char> &&blk) {
        std::lock_guard<std::mutex> lk(mu);
        if (free_blocks.size() < depth + 8) free_blocks.push_back(std::move(blk));
    }
};

static int g_inflate_device = [] { const char *e = cli_env("COLORID_GPU_INFLATE"); return e && atoi(e) > 0 ? 0 : -1; }();
void LineReader::inflate_on_gpu(int device) { g_inflate_device = device; }

static std::mutex g_prefetch_mu;
static std::vector<std::pair<std::string, LineReader::Impl *>> g_prefetched;

static LineReader::Impl *open_stream(const std::string &path, bool ahead) {
    LineReader::Impl *p = new LineReader::Impl;
    const char *gt = cli_env("COLORID_GZ_THREADS");
    if (gt) p->gz_threads = atoi(gt);
    if (p->gz_threads < 1) p->gz_threads = 1;
    if (LineReader::Impl::is_bgzf(path)) {   // (one thread too: whole members through libdeflate beat a zlib stream)
        p->raw = fopen(path.c_str(), "rb");
        if (!p->raw) die("file not found: %s", path.c_str());
        if (ahead) p->depth = 16;   // batches of 16 MiB of text
        p->gpu_device = g_inflate_device;
        if (p->gpu_device >= 0) p->depth = 4;   // (batches of 64 MiB)
        p->worker = std::thread([p] { p->run_bgzf(); });
        return p;
    }
    if (ahead) p->depth = 64;       // blocks of 4 MiB
    if (libdeflate().ok && LineReader::Impl::is_gzip(path)) {   // zlib's raw inflate + libdeflate's CRC-32 (run_gzip)
        p->raw = fopen(path.c_str(), "rb");
        if (!p->raw) die("file not found: %s", path.c_str());
        ++g_gzip_streams;   // (run_gzip counts itself out)
        p->worker = std::thread([p] { p->run_gzip(); });
        return p;
    }
    p->gz = gzopen(path.c_str(), "rb");
    if (!p->gz) die("file not found: %s", path.c_str());
    gzbuffer(p->gz, 1 << 20);
    p->worker = std::thread([p] { p->run(); });
    return p;
}
static void close_stream(LineReader::Impl *p) {
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_free.notify_all();
    if (p->worker.joinable()) p->worker.join();
    if (p->gz) gzclose(p->gz);
    if (p->raw) fclose(p->raw);
    delete p;
}

void LineReader::prefetch(const std::string &path) {
    Impl *p = open_stream(path, true);
    std::lock_guard<std::mutex> lk(g_prefetch_mu);
    g_prefetched.emplace_back(path, p);
}
void LineReader::drop_prefetched() {
    std::vector<std::pair<std::string, Impl *>> left;
    {
        std::lock_guard<std::mutex> lk(g_prefetch_mu);
        left.swap(g_prefetched);
    }
    for (auto &e : left) close_stream(e.second);
}

LineReader::LineReader(const std::string &path) : p_(nullptr) {
    {
        std::lock_guard<std::mutex> lk(g_prefetch_mu);
        for (size_t i = 0; i < g_prefetched.size(); ++i)
            if (g_prefetched[i].first == path) {
                p_ = g_prefetched[i].second;
                g_prefetched.erase(g_prefetched.begin() + (long)i);
                break;
            }
    }
    if (!p_) p_ = open_stream(path, false);
}
LineReader::~LineReader() { close_stream(p_); }
bool LineReader::next(std::string &line) {
    line.clear();
    bool got = false;
    for (;;) {
        if (p_->pos == p_->cur.size() && !p_->refill()) break;
        got = true;
        const char *b = p_->cur.data() + p_->pos;
        const size_t left = p_->cur.size() - p_->pos;
        const char *nl = static_cast<const char *>(memchr(b, '\n', left));
        if (nl) {
            line.append(b, (size_t)(nl - b));
            p_->pos += (size_t)(nl - b) + 1;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            return true;
        }
        line.append(b, left);   // the line continues in the next block
        p_->pos = p_->cur.size();
    }
    if (!got) return false;
    if (!line.empty() && line.back() == '\r') line.pop_back();   // last line without a newline
    return true;
}

bool LineReader::next(const char *&ptr, size_t &len) {
    bool got = false, carried = false;
    for (;;) {
        if (p_->pos == p_->cur.size() && !p_->refill()) break;
        got = true;
        const char *b = p_->cur.data() + p_->pos;
        const size_t left = p_->cur.size() - p_->pos;
        const char *nl = static_cast<const char *>(memchr(b, '\n', left));
        if (nl) {
            p_->pos += (size_t)(nl - b) + 1;
            if (!carried) { ptr = b; len = (size_t)(nl - b); }
            else { p_->carry.append(b, (size_t)(nl - b)); ptr = p_->carry.data(); len = p_->carry.size(); }
            if (len && ptr[len - 1] == '\r') --len;
            return true;
        }
        if (!carried) { p_->carry.clear(); carried = true; }
        p_->carry.append(b, left);   // the line continues in the next block (which recycles this one)
        p_->pos = p_->cur.size();
    }
    if (!got) return false;
    ptr = p_->carry.data(); len = p_->carry.size();   // last line without a newline
    if (len && ptr[len - 1] == '\r') --len;
    return true;
}

bool LineReader::next_block(std::vector<char> &blk) { return p_->take_block(blk); }
void LineReader::recycle(std::vector<char> &&blk) { p_->give_back(std::move(blk)); }

// Newlines of [p, end): every fourth one (counting from `lines` seen before) closes a FASTQ record, whose end offset (relative to
// `base`) is appended.  FASTQ lines are short — an id of ten bytes, 150 bases — so one memchr call per line is mostly call overhead
// (4 M calls per million reads, ~90 ms on the reading thread); 32 bytes at a time with AVX2 compare + movemask, the set bits walked
// with tzcnt, takes a quarter of that.  Returns the new line count.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static uint64_t scan_records_avx2(const char *base, const char *p, const char *end, uint64_t lines, std::vector<uint32_t> &rec_end) {
    const __m256i nl = _mm256_set1_epi8('\n');
    for (; p + 32 <= end; p += 32) {
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p)), nl));
        while (m) {
            const uint32_t b = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            if ((++lines & 3u) == 0) rec_end.push_back((uint32_t)(p + b + 1 - base));
        }
    }
    for (; p < end; ++p)
        if (*p == '\n' && (++lines & 3u) == 0) rec_end.push_back((uint32_t)(p + 1 - base));
    return lines;
}
#endif
static uint64_t scan_records(const char *base, const char *p, const char *end, uint64_t lines, std::vector<uint32_t> &rec_end) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && !cli_env("COLORID_NO_AVX2");
    if (avx2) return scan_records_avx2(base, p, end, lines, rec_end);
#endif
    while (p < end) {
        const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!nl) break;
        p = nl + 1;
        if ((++lines & 3u) == 0) rec_end.push_back((uint32_t)(p - base));
    }
    return lines;
}

// ---------------------------------------------------------------------------------------------- BgzfMemberReader
// page-locked memory is slow to get and to give back (hipHostMalloc / hipHostFree of 256 MiB: tens of milliseconds each): buffers a
// reader lets go of wait in a pool for the next one — never returned to the runtime, the process ends with the command
namespace {
std::mutex g_pin_mu;
std::vector<std::pair<unsigned char *, size_t>> g_pin_pool;
}
void PinnedBuf::release() {
    if (!p) return;
    if (pinned) { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_pool.emplace_back(p, cap); }
    else free(p);
    p = nullptr; cap = 0;
}
void PinnedBuf::reserve(size_t want) {
    if (want <= cap) return;
    release();
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_pool.size(); ++i)
            if (g_pin_pool[i].second >= want) {
                p = g_pin_pool[i].first; cap = g_pin_pool[i].second; pinned = true;
                g_pin_pool.erase(g_pin_pool.begin() + (long)i);
                return;
            }
    }
    size_t c = want + want / 8 + 4096;
    void *q = nullptr;
    if (cid_pinned_alloc(c, &q) == CID_OK) { p = static_cast<unsigned char *>(q); pinned = true; }
    else { p = static_cast<unsigned char *>(malloc(c)); pinned = false; if (!p) die("out of memory (%zu bytes)", c); }
    cap = c;
}

void StretchBuf::release() {
    if (!p) return;
    if (pinned) { std::lock_guard<std::mutex> lk(g_pin_mu); g_pin_pool.emplace_back(p, cap); }
    else free(p);
    p = nullptr; n = cap = 0;
}
void StretchBuf::reserve(size_t want) {
    if (want <= cap) return;
    static const bool lock_pages = [] { const char *e = cli_env("COLORID_DEVICE_FASTQ_PINNED"); return e && atoi(e) != 0; }();   // (off: see BgzfStretch)
    size_t c = cap + cap / 2;
    if (c < want) c = want;
    unsigned char *q = nullptr;
    size_t qcap = c;
    bool qpinned = false;
    if (lock_pages) {
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            for (size_t i = 0; i < g_pin_pool.size(); ++i)
                if (g_pin_pool[i].second >= c) {
                    q = g_pin_pool[i].first; qcap = g_pin_pool[i].second; qpinned = true;
                    g_pin_pool.erase(g_pin_pool.begin() + (long)i);
                    break;
                }
        }
        void *v = nullptr;
        if (!q && cid_pinned_alloc(c, &v) == CID_OK) { q = static_cast<unsigned char *>(v); qpinned = true; }
    }
    if (!q) { q = static_cast<unsigned char *>(malloc(c)); if (!q) die("out of memory (%zu bytes)", c); }
    if (n) memcpy(q, p, n);
    const size_t keep = n;
    release();
    p = q; n = keep; cap = qcap; pinned = qpinned;
}

// members [m0, m1) of `s` inflated into out (their texts back to back) by the pool's threads and this one; every member's CRC-32 and
// ISIZE are checked (libdeflate when the host has it, else zlib)
static void inflate_members_host(TaskPool *pool, int n_threads, const BgzfStretch &s, size_t m0, size_t m1, unsigned char *out) {
    if (m1 <= m0) return;
    std::vector<size_t> at(m1 - m0 + 1, 0);
    for (size_t i = m0; i < m1; ++i) at[i - m0 + 1] = at[i - m0] + s.text_len[i];
    const int nt = (int)std::min<size_t>((size_t)std::max(1, n_threads), m1 - m0);
    std::vector<int> bad(nt, 0);
    auto work = [&](int t) {
        const LibDeflate &ld = libdeflate();
        void *dc = ld.ok ? ld.alloc() : nullptr;
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (!dc && inflateInit2(&zs, 15 + 16) != Z_OK) { bad[t] = 1; return; }
        for (size_t i = m0 + (size_t)t; i < m1; i += (size_t)nt) {
            unsigned char *dst = out + at[i - m0];
            if (dc) {   // exact output size given, no "actual" pointer: anything else than ISIZE bytes is an error
                if (ld.gzip(dc, s.bytes.p + s.off[i], s.len[i], dst, s.text_len[i], nullptr) != 0) { bad[t] = 1; break; }
            } else {
                inflateReset(&zs);
                zs.next_in = const_cast<Bytef *>(s.bytes.p + s.off[i]); zs.avail_in = (uInt)s.len[i];
                zs.next_out = dst; zs.avail_out = (uInt)s.text_len[i];
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.total_out != s.text_len[i]) { bad[t] = 1; break; }   // zlib checked the member's CRC-32
            }
        }
        if (dc) ld.release(dc); else inflateEnd(&zs);
    };
    if (pool && nt > 1) pool->parallel_for((size_t)nt, [&](size_t t) { work((int)t); });
    else for (int t = 0; t < nt; ++t) work(t);
    for (int t = 0; t < nt; ++t) if (bad[t]) die("corrupt gzip member (inflate / CRC-32 failed)");
}

// Two stages (round 6; round 5's one thread read a stretch, walked its members and inflated the host's share before it went back for the
// next read: the device waited 87-168 ms of a 16 M-read file's 310-375 ms for it):
//   reader   the file's next bytes by several preads side by side (the page cache copies at a few GB/s per thread; the members' sizes are
//            in their headers, so the walk from member to member needs the bytes and nothing else), the stretch cut at whole members
//   inflater the host's share of the stretch (its LAST members' text) on the reader's pool
// A stretch is handed out when both are through; the stages work on different stretches at the same time.
struct BgzfMemberReader::Impl {
    double host_share = 0.0;
    int host_threads = 0;
    std::unique_ptr<TaskPool> pool, read_pool;
    int fd = -1;
    uint64_t fpos = 0;                 // the next byte of the file not read yet
    uint64_t file_size = ~0ull;        // (a pipe: unknown)
    size_t text_target;
    std::thread worker, inflater;
    std::mutex mu;
    std::condition_variable cv_full, cv_free, cv_walked;
    std::deque<BgzfStretch> full, spare, walked;
    bool eof = false, stop = false, handed_last = false, walked_last = false;
    std::vector<unsigned char> tail;   // bytes read from the file behind the last whole member of the stretch before
    static constexpr size_t kDepth = 2;
    static constexpr int kReadThreads = 4;

    // up to `chunk` bytes of the file behind fpos to dst; returns what the file had
    size_t read_some(unsigned char *dst, size_t chunk) {
        const size_t piece = (size_t)4 << 20;
        const size_t n_parts = chunk >= 2 * piece ? std::min<size_t>((size_t)kReadThreads, chunk / piece) : 1;
        std::vector<size_t> got(n_parts, 0);
        auto part = [&](size_t i) {
            const size_t a = chunk * i / n_parts, b = chunk * (i + 1) / n_parts;
            size_t done = 0;
            while (a + done < b) {
                const ssize_t n = pread(fd, dst + a + done, b - a - done, (off_t)(fpos + a + done));
                if (n < 0) { if (errno == EINTR) continue; die("read error on a block-gzip file: %s", strerror(errno)); }
                if (n == 0) break;
                done += (size_t)n;
            }
            got[i] = done;
        };
        if (n_parts > 1) {
            if (!read_pool) read_pool.reset(new TaskPool(kReadThreads - 1));
            read_pool->parallel_for(n_parts, part);
        } else part(0);
        size_t total = 0;
        for (size_t i = 0; i < n_parts; ++i) {   // (a short part is the end of the file: nothing behind it was read)
            const size_t want = chunk * (i + 1) / n_parts - chunk * i / n_parts;
            total += got[i];
            if (got[i] < want) break;
        }
        fpos += total;
        return total;
    }

    void run() {
        bool file_end = false;
        size_t guess = 0;   // compressed bytes of the stretch before: the next one is read with one round of reads of about that size
        for (;;) {
            BgzfStretch s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_free.wait(lk, [&] { return stop || full.size() + walked.size() < kDepth + 1; });
                if (stop) return;
                if (!spare.empty()) { s = std::move(spare.front()); spare.pop_front(); }
            }
            s.off.clear(); s.len.clear(); s.text_len.clear(); s.text_bytes = 0; s.last = false;
            {   // ONE allocation per buffer: the stretch before tells the size, the first one takes the file's (compressed FASTQ is a
                // quarter to a half of its text), and the file's end bounds both
                const size_t left = file_size > fpos ? (size_t)(file_size - fpos) : 0;
                size_t room = guess ? guess + guess / 8 : text_target / 2 + ((size_t)1 << 20);
                if (file_size != ~0ull && room > left + 4096) room = left + 4096;
                s.bytes.reserve(tail.size() + room);
            }
            if (!tail.empty()) memcpy(s.bytes.p, tail.data(), tail.size());
            s.bytes.n = tail.size();
            tail.clear();
            size_t pos = 0;   // the first byte not yet assigned to a member
            auto need = [&](size_t bytes) {   // make s.bytes[pos, pos + bytes) available if the file has them
                while (s.bytes.n - pos < bytes && !file_end) {
                    size_t chunk = (size_t)8 << 20;
                    if (guess > s.bytes.n + chunk) chunk = guess - s.bytes.n;   // the bulk of a stretch in one round of reads
                    if (s.bytes.n - pos + chunk < bytes) chunk = bytes;
                    s.bytes.reserve(s.bytes.n + chunk);
                    const size_t n = read_some(s.bytes.p + s.bytes.n, chunk);
                    s.bytes.n += n;
                    if (n < chunk) file_end = true;
                }
                return s.bytes.n - pos >= bytes;
            };
            while (s.text_bytes < text_target) {
                if (!need(18)) {
                    if (s.bytes.n - pos != 0) die("truncated gzip member header");
                    s.last = true;
                    break;
                }
                {
                    const unsigned char *h = s.bytes.p + pos;
                    const size_t xlen = h[10] | ((size_t)h[11] << 8);
                    if (h[0] == 0x1f && h[1] == 0x8b && (h[3] & 4) && !need(12 + xlen)) die("truncated gzip member header");
                }
                const size_t msz = LineReader::Impl::bgzf_member_size(s.bytes.p + pos, s.bytes.n - pos);
                if (msz < 26) die("not a block-gzip (BGZF) member inside a BGZF file: mixed gzip streams are not supported in one file");
                if (!need(msz)) die("truncated gzip member");
                const unsigned char *t = s.bytes.p + pos + msz - 4;
                const size_t isize = t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                if (isize > (1u << 16)) die("BGZF member larger than 64 KiB");
                s.off.push_back((uint32_t)pos); s.len.push_back((uint32_t)msz); s.text_len.push_back((uint32_t)isize);
                s.text_bytes += isize;
                pos += msz;
                if (pos >= (3ull << 30)) break;   // (offsets are 32-bit)
            }
            tail.assign(s.bytes.p + pos, s.bytes.p + s.bytes.n);
            s.bytes.n = pos;
            guess = pos;
            const bool last = s.last;
            {
                std::lock_guard<std::mutex> lk(mu);
                walked.push_back(std::move(s));
                if (last) walked_last = true;
            }
            cv_walked.notify_one();
            if (last) return;
        }
    }

    void run_inflater() {
        for (;;) {
            BgzfStretch s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_walked.wait(lk, [&] { return stop || !walked.empty(); });
                if (stop) return;
                s = std::move(walked.front());
                walked.pop_front();
            }
            // the host's share of the stretch: the LAST members' text (the device's members come first in the text)
            s.device_members = s.off.size();
            s.host_text_bytes = 0;
            if (host_share > 0.0 && !s.off.empty()) {
                const uint64_t want = (uint64_t)((double)s.text_bytes * host_share);
                size_t m0 = s.off.size();
                uint64_t got = 0;
                while (m0 > 0 && got < want) { --m0; got += s.text_len[m0]; }
                s.device_members = m0;
                s.host_text_bytes = (size_t)got;
                s.host_text.reserve(s.host_text_bytes + 64);
                if (!pool && host_threads > 1) pool.reset(new TaskPool(host_threads - 1));
                inflate_members_host(pool.get(), host_threads, s, m0, s.off.size(), s.host_text.p);
            }
            const bool last = s.last;
            {
                std::lock_guard<std::mutex> lk(mu);
                full.push_back(std::move(s));
                if (last) eof = true;
            }
            cv_full.notify_one();
            if (last) return;
        }
    }
};

BgzfMemberReader::BgzfMemberReader(const std::string &path, size_t text_target, double host_share, int host_threads) : p_(new Impl) {
    p_->text_target = text_target ? text_target : 1;
    p_->host_share = host_share < 0.0 ? 0.0 : host_share > 1.0 ? 1.0 : host_share;
    p_->host_threads = host_threads;
    p_->fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (p_->fd < 0) die("file not found: %s", path.c_str());
    (void)posix_fadvise(p_->fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    {
        struct stat sb;
        if (fstat(p_->fd, &sb) == 0 && S_ISREG(sb.st_mode)) p_->file_size = (uint64_t)sb.st_size;
    }
    p_->worker = std::thread([this] { p_->run(); });
    p_->inflater = std::thread([this] { p_->run_inflater(); });
}
BgzfMemberReader::~BgzfMemberReader() {
    {
        std::lock_guard<std::mutex> lk(p_->mu);
        p_->stop = true;
    }
    p_->cv_free.notify_all();
    p_->cv_walked.notify_all();
    if (p_->worker.joinable()) p_->worker.join();
    if (p_->inflater.joinable()) p_->inflater.join();
    if (p_->fd >= 0) ::close(p_->fd);
    delete p_;
}
bool BgzfMemberReader::next(BgzfStretch &s) {
    std::unique_lock<std::mutex> lk(p_->mu);
    if (p_->handed_last) return false;
    if (s.bytes.cap) { p_->spare.push_back(std::move(s)); s = BgzfStretch(); }
    p_->cv_full.wait(lk, [&] { return !p_->full.empty(); });
    s = std::move(p_->full.front());
    p_->full.pop_front();
    if (s.last) p_->handed_last = true;
    lk.unlock();
    p_->cv_free.notify_one();
    return true;
}
bool BgzfMemberReader::is_bgzf(const std::string &path) { return LineReader::Impl::is_bgzf(path); }
size_t BgzfMemberReader::text_target() const { return p_->text_target; }

namespace {
std::mutex g_bgzf_mu;
std::vector<std::pair<std::string, std::unique_ptr<BgzfMemberReader>>> g_bgzf_ahead;
}
void BgzfMemberReader::prefetch(const std::string &path, size_t text_target, double host_share, int host_threads) {
    std::unique_ptr<BgzfMemberReader> r(new BgzfMemberReader(path, text_target, host_share, host_threads));
    std::lock_guard<std::mutex> lk(g_bgzf_mu);
    g_bgzf_ahead.emplace_back(path, std::move(r));
}
std::unique_ptr<BgzfMemberReader> BgzfMemberReader::open(const std::string &path, size_t text_target, double host_share, int host_threads) {
    {
        std::lock_guard<std::mutex> lk(g_bgzf_mu);
        for (size_t i = 0; i < g_bgzf_ahead.size(); ++i)
            if (g_bgzf_ahead[i].first == path && g_bgzf_ahead[i].second->text_target() <= text_target) {   // (a larger stretch than allowed: start over)
                std::unique_ptr<BgzfMemberReader> r = std::move(g_bgzf_ahead[i].second);
                g_bgzf_ahead.erase(g_bgzf_ahead.begin() + (long)i);
                return r;
            }
    }
    return std::unique_ptr<BgzfMemberReader>(new BgzfMemberReader(path, text_target, host_share, host_threads));
}
void BgzfMemberReader::drop_prefetched() {
    std::lock_guard<std::mutex> lk(g_bgzf_mu);
    g_bgzf_ahead.clear();
}

bool RecordChunker::next(RecChunk &c) {
    c.rec_end.clear();
    for (;;) {
        if (done_) return false;
        std::vector<char> blk;
        const bool have = r_.next_block(blk);
        if (!have) {   // end of input: what is left is at most three lines and perhaps a fourth without its newline
            done_ = true;
            if (carry_.empty()) return false;
            if (carry_.back() != '\n') carry_.push_back('\n');   // BufRead::lines() yields an unterminated last line too
            blk.assign(LineReader::kHeadroom, 0);
        }
        // the previous block's tail goes in front of this block's text: into the headroom, or (a record longer than that) into a new buffer
        size_t begin = LineReader::kHeadroom;
        if (!carry_.empty()) {
            if (carry_.size() <= LineReader::kHeadroom && have) {
                begin -= carry_.size();
                memcpy(blk.data() + begin, carry_.data(), carry_.size());
            } else {
                std::vector<char> joined(LineReader::kHeadroom + carry_.size() + (blk.size() - LineReader::kHeadroom));
                memcpy(joined.data() + LineReader::kHeadroom, carry_.data(), carry_.size());
                memcpy(joined.data() + LineReader::kHeadroom + carry_.size(), blk.data() + LineReader::kHeadroom, blk.size() - LineReader::kHeadroom);
                if (have) r_.recycle(std::move(blk));
                blk.swap(joined);
            }
        }
        if (blk.size() >= (1ull << 32)) die("a FASTQ record of more than 4 GiB");
        // newlines from where the carried lines end; every fourth one closes a record
        const char *base = blk.data(), *end = base + blk.size();
        const char *p = base + begin + carry_.size();
        uint64_t lines = carry_lines_;
        if (!have) { p = base + begin; lines = 0; }   // (the final carry is scanned whole: its last newline was just added)
        lines = scan_records(base, p, end, lines, c.rec_end);
        const size_t boundary = c.rec_end.empty() ? begin : c.rec_end.back();
        carry_.assign(base + boundary, (size_t)(end - (base + boundary)));
        carry_lines_ = lines & 3u;
        if (!have) carry_.clear();
        if (c.rec_end.empty()) {   // no whole record yet (very long reads): keep reading
            if (have) r_.recycle(std::move(blk));
            continue;
        }
        c.buf = std::move(blk);
        c.begin = begin;
        return true;
    }
}

void qual_mask(std::string &seq, const std::string &qual, uint8_t q) {
    if (q == 0) return;
    // the reference walks qual.chars() and takes one base per quality char: output length = qual length
    if (seq.size() < qual.size()) die("ERROR: could not get the next nt in the sequence");
    seq.resize(qual.size());
    const uint8_t max_quality = (uint8_t)(q + 33);
    for (size_t i = 0; i < qual.size(); ++i)
        if ((uint8_t)qual[i] < max_quality) seq[i] = 'N';
}

// ---------------------------------------------------------------------------------------------- KmerMap

static inline uint64_t key_hash(const uint8_t *p, uint32_t k) {
    uint64_t h = 0x9E3779B185EBCA87ULL ^ k;
    uint32_t i = 0;
    for (; i + 8 <= k; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0xC2B2AE3D27D4EB4FULL;
        h ^= h >> 29;
    }
    uint64_t w = 0;
    if (i < k) memcpy(&w, p + i, k - i);
    h = (h ^ w) * 0x165667B19E3779F9ULL;
    return h ^ (h >> 32);
}

KmerMap::KmerMap(uint32_t k) : k_(k), table_(1 << 16, 0) {}

void KmerMap::rehash() {
    std::vector<uint32_t> t(table_.size() * 2, 0);
    const size_t mask = t.size() - 1;
    for (size_t e = 0; e < counts_.size(); ++e) {
        size_t h = key_hash(keys_.data() + e * k_, k_) & mask;
        while (t[h]) h = (h + 1) & mask;
        t[h] = (uint32_t)e + 1;
    }
    table_.swap(t);
}

void KmerMap::add(const uint8_t *key, uint32_t n) {
    size_t mask = table_.size() - 1;
    size_t h = key_hash(key, k_) & mask;
    while (table_[h]) {
        const size_t e = table_[h] - 1;
        if (memcmp(keys_.data() + e * k_, key, k_) == 0) {
            if (counts_[e] > UINT32_MAX - n) die("k-mer multiplicity overflows 32 bits");
            counts_[e] += n;
            return;
        }
        h = (h + 1) & mask;
    }
    if (counts_.size() >= UINT32_MAX - 1) die("more than 2^32 distinct k-mers in one query");
    keys_.insert(keys_.end(), key, key + k_);
    counts_.push_back(n);
    table_[h] = (uint32_t)counts_.size();
    if (counts_.size() * 2 > table_.size()) rehash();
}

void KmerMap::clean(uint64_t t) {
    size_t w = 0;
    for (size_t e = 0; e < counts_.size(); ++e) {
        if (counts_[e] > t) {
            if (w != e) {
                memmove(keys_.data() + w * k_, keys_.data() + e * k_, k_);
                counts_[w] = counts_[e];
            }
            ++w;
        }
    }
    counts_.resize(w);
    keys_.resize(w * k_);
    std::fill(table_.begin(), table_.end(), 0);
    const size_t mask = table_.size() - 1;
    for (size_t e = 0; e < w; ++e) {
        size_t h = key_hash(keys_.data() + e * k_, k_) & mask;
        while (table_[h]) h = (h + 1) & mask;
        table_[h] = (uint32_t)e + 1;
    }
}

int64_t auto_cutoff_from_histogram(const std::map<uint64_t, uint64_t> &hm, uint64_t n_distinct) {  // kmer.rs:866-942
    uint64_t max_cov = 0, sum = 0;
    for (auto &kv : hm) { max_cov = kv.first > max_cov ? kv.first : max_cov; sum += kv.first * kv.second; }
    const double total_mean = (double)sum / (double)n_distinct;
    if (total_mean < 1.5) return 0;
    const size_t ncov = max_cov >= 1 ? (size_t)(max_cov - 1) : 0;  // coverages[j] = #k-mers with multiplicity j+1, j < max_cov-1
    if (ncov == 0) return -1;                                       // `coverages.len() - 1` underflows: panic
    auto cov = [&](size_t j) -> uint64_t { auto it = hm.find(j + 1); return it == hm.end() ? 0 : it->second; };
    const size_t nd1 = ncov >= 2 ? ncov - 2 : 0;
    if (nd1 == 0) return -1;                                        // `d1.len() - 1` underflows: panic
    // max_cov can be huge (poly-A): d1/d2 only matter up to their first element < 1, so walk them lazily
    size_t first_d1 = 0, first_d2 = 0;
    auto d1 = [&](size_t i) { return (double)cov(i + 1) / (double)cov(i + 2); };
    for (size_t i = 0; i < nd1; ++i)
        if (d1(i) < 1.0) { first_d1 = i + 1; break; }
    for (size_t i = 0; i + 1 < nd1; ++i)
        if (d1(i) / d1(i + 1) < 1.0) { first_d2 = i + 1; break; }
    uint64_t bigsum = 0, num = 0;
    for (auto &kv : hm) {  // i * coverages[1+i] over i = 0..ncov-2, coverages[1+i] = histo[i+2]
        if (kv.first >= 2 && kv.first - 2 + 1 < ncov) { bigsum += (kv.first - 2) * kv.second; num += kv.second; }
    }
    const double mean = (double)bigsum / (double)num;
    if (first_d1 > 0 && (double)first_d1 < mean * 0.75) return (int64_t)first_d1;
    if (first_d2 > 0) return (int64_t)first_d2;
    const double c = std::ceil(mean / 2.0);
    const uint64_t cu = (c != c || c <= 0.0) ? 0 : (uint64_t)c;
    return (int64_t)(cu > 1 ? cu : 1);
}

int64_t KmerMap::auto_cutoff() const {
    std::map<uint64_t, uint64_t> hm;
    for (uint32_t c : counts_) hm[c] += 1;
    return auto_cutoff_from_histogram(hm, counts_.size());
}

// ---------------------------------------------------------------------------------------------- window walk

static inline bool good_base(uint8_t c) {
    const uint8_t u = c & 0xDF;
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}
static inline uint8_t switch_base(uint8_t c) {  // kmer.rs:847-863
    switch (c) {
    case 'a': return 't'; case 'c': return 'g'; case 't': return 'a'; case 'g': return 'c';
    case 'u': return 'a'; case 'n': return 'n';
    case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C';
    case 'U': return 'A'; case 'N': return 'N';
    default: return 'N';
    }
}

// One string: windows i = 0, d, 2d...; keep iff (no filter or all k bases in ACGTacgt); choose fwd < rc ? fwd : rc on
// the raw bytes; optionally upper-case the chosen string (FASTA paths), then count it.
static void walk(const uint8_t *l, size_t len, size_t d, bool filter_n, bool upper, KmerMap &out) {
    const size_t k = out.k();
    if (len < k) return;
    std::vector<uint8_t> rcbuf(len), tmp(k);
    for (size_t i = 0; i < len; ++i) rcbuf[i] = switch_base(l[len - 1 - i]);
    // bad[i] = number of non-ACGT bases in l[0..i)
    std::vector<uint32_t> bad(len + 1, 0);
    for (size_t i = 0; i < len; ++i) bad[i + 1] = bad[i] + (good_base(l[i]) ? 0u : 1u);
    for (size_t i = 0; i + k <= len; i += d) {
        if (filter_n && bad[i + k] != bad[i]) continue;
        const uint8_t *fwd = l + i, *rc = rcbuf.data() + (len - (i + k));
        const uint8_t *pick = memcmp(fwd, rc, k) < 0 ? fwd : rc;
        if (upper) {
            for (size_t t = 0; t < k; ++t) tmp[t] = (pick[t] >= 'a' && pick[t] <= 'z') ? (uint8_t)(pick[t] - 32) : pick[t];
            out.add(tmp.data());
        } else {
            out.add(pick);
        }
    }
}

void kmerize_vector(const std::vector<std::string> &v, size_t d, KmerMap &out) {
    for (const std::string &l : v) walk(reinterpret_cast<const uint8_t *>(l.data()), l.size(), d, true, true, out);
}

bool kmerize_string(const std::string &l, KmerMap &out) {
    if (l.size() < out.k()) return false;
    walk(reinterpret_cast<const uint8_t *>(l.data()), l.size(), 1, false, true, out);
    return true;
}

void kmers_from_fq_qual(const std::string &path, uint8_t q, KmerMap &out) {
    LineReader r(path);
    std::string line, seq;
    uint64_t line_count = 1;
    while (r.next(line)) {
        if (line_count % 4 == 2) seq = line;
        else if (line_count % 4 == 0) {
            qual_mask(seq, line, q);
            walk(reinterpret_cast<const uint8_t *>(seq.data()), seq.size(), 1, true, false, out);
        }
        ++line_count;
    }
}

void kmers_fq_pe_qual(const std::string &p1, const std::string &p2, uint8_t q, KmerMap &out) {
    LineReader r1(p1), r2(p2);
    std::string l1, l2, s1, s2;
    uint64_t line_count = 1;
    while (r1.next(l1)) {
        if (!r2.next(l2)) break;
        if (line_count % 4 == 2) { s1 = l1; s2 = l2; }
        else if (line_count % 4 == 0) {
            qual_mask(s1, l1, q);
            qual_mask(s2, l2, q);
            walk(reinterpret_cast<const uint8_t *>(s1.data()), s1.size(), 1, true, false, out);
            walk(reinterpret_cast<const uint8_t *>(s2.data()), s2.size(), 1, true, false, out);
        }
        ++line_count;
    }
}

}  // namespace colorid
