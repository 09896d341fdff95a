// The FASTQ front end of read_id on the device (include/colorid_hip.h "cid_fastq"): what the reference does per read on
// the host before its search — inflate the gzip stream (flate2 MultiGzDecoder, src/read_id_mt_pe.rs:848-856), walk the lines four at a
// time (:862-879: header, sequence, '+', quality), mask low-quality bases (seq::qual_mask, src/seq.rs:36-56), collect (id, [seq(, mate)])
// (:868-879, :927-975) — happens here for a whole stretch of the file at once, and the reads never exist in host memory:
//   block-gzip members (compressed) --H2D--> k_bgzf_inflate --> text in HBM (behind the unfinished tail of the stretch before)
//   newline positions (one stream compaction) --> every fourth line ends a record --> per record: id / sequence / quality spans, the
//   masked length, the read's bytes and k-mer windows --> exclusive scans --> bases | seq_off | read_seq0 exactly as cid_readid_count_dev
//   takes them, and the id lines NUL-terminated back to back --> k_readid ... --> sparse (colour, count) rows.
// Up go the compressed bytes (~60 MB per million 150-bp reads), down come per read: the id, n_kmers, status and a handful of entries.
// Two files = read pairs: record r of either file makes read r; a call takes as many records as both hold, the rest waits on the device.
// Host code + a few elementwise kernels; inflate and classification are the existing kernels.
#include <cstring>

#include <chrono>
#include <deque>
#include <new>
#include <vector>

#include "cid_api_common.hpp"
#include "cid_scan.hpp"

using cid::fail;

namespace cid {

// ---- line ends of a text.  One wave walks a chunk of kNlChunk bytes, 1 KiB a round (64 lanes x 16 aligned bytes); a lane's newlines
// are the set bits of a 16-bit mask.  k_nl_count leaves the chunks' counts (then one exclusive scan over the chunks), k_nl_positions
// writes the positions in ascending order: chunk base + the rounds before + the lanes before (a wave prefix sum) + the bits before.
// (rocPRIM's select over a counting iterator with a byte predicate took 1.0 ms per 256 MB of text; these two passes read the text at
// HBM speed.)
constexpr uint32_t kNlChunk = 8192;
__device__ __forceinline__ uint32_t newline_mask16(const uint8_t *text, uint32_t at, uint32_t len) {
    if (at >= len) return 0u;
    const uint4 v = *reinterpret_cast<const uint4 *>(text + at);   // (the text buffers carry 64 bytes of slack behind len)
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t y = w[i] ^ 0x0A0A0A0Au;                                            // a newline is a zero byte of y
        const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);        // 0x80 exactly at the zero bytes
        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * i);
    }
    const uint32_t valid = len - at;   // bytes of this piece inside the text
    return valid >= 16 ? m : (m & ((1u << valid) - 1u));
}
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *text, uint32_t len, uint32_t n_chunks, uint32_t *chunk_count) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t ch = wave; ch <= n_chunks; ch += n_waves) {
        if (ch == n_chunks) { if (lane == 0) chunk_count[ch] = 0; continue; }   // slot n_chunks receives the total
        uint32_t c = 0;
        for (uint32_t r = 0; r < kNlChunk / 1024; ++r) c += (uint32_t)__builtin_popcount(newline_mask16(text, ch * kNlChunk + r * 1024 + lane * 16, len));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) chunk_count[ch] = c;
    }
}
__global__ __launch_bounds__(256) void k_nl_positions(const uint8_t *text, uint32_t len, uint32_t n_chunks, const uint32_t *chunk_off, uint32_t *nl, uint64_t *n_nl) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_nl = chunk_off[n_chunks];
    for (uint32_t ch = wave; ch < n_chunks; ch += n_waves) {
        uint32_t base = chunk_off[ch];
        for (uint32_t r = 0; r < kNlChunk / 1024; ++r) {
            const uint32_t at = ch * kNlChunk + r * 1024 + lane * 16;
            uint32_t m = newline_mask16(text, at, len);
            const uint32_t c = (uint32_t)__builtin_popcount(m);
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += u;
            }
            uint32_t to = base + incl - c;
            while (m) {
                nl[to++] = at + (uint32_t)__builtin_ctz(m);
                m &= m - 1;
            }
            base += (uint32_t)__shfl(incl, 63, 64);
        }
    }
}

struct FqFile {   // device view of one input's text for the kernels
    const uint8_t *text;
    const uint32_t *nl;       // positions of the line ends, ascending
    const uint64_t *n_nl;     // how many
    uint32_t len;             // bytes of text
};

// BufRead::lines() yields an unterminated last line too: at the end of the input a text that does not end in '\n' gets a line end
// at its length
__global__ void k_fq_tail(const uint8_t *text, uint32_t len, uint32_t *nl, uint64_t *n_nl) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && len && text[len - 1] != '\n') { nl[*n_nl] = len; *n_nl += 1; }
}

struct FqStats {
    uint64_t n_rec;          // complete records both files hold
    uint64_t boundary[2];    // per file: the first byte behind its record n_rec - 1
    uint64_t file_rec[2];    // per file: the complete records its text holds
    uint32_t max_bytes, max_win, err;   // max_bytes / max_win: per READ (both mates together: the classification kernel's LDS sizing)
    uint32_t max_seq;                   // the longest single sequence (the k-mer set's per-sequence limit)
};
__global__ void k_fq_nrec(FqFile f0, FqFile f1, int n_files, FqStats *st) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t n = *f0.n_nl / 4;
    if (n_files == 2 && *f1.n_nl / 4 < n) n = *f1.n_nl / 4;
    st->file_rec[0] = *f0.n_nl / 4;
    st->file_rec[1] = n_files == 2 ? *f1.n_nl / 4 : 0;
    st->n_rec = n;
    st->boundary[0] = n ? f0.nl[4 * n - 1] + 1ull : 0ull;
    st->boundary[1] = (n_files == 2 && n) ? f1.nl[4 * n - 1] + 1ull : 0ull;
    st->max_bytes = 0; st->max_win = 0; st->err = 0; st->max_seq = 0;
}

struct FqSpan { uint32_t seq, qual, len; };   // where a sequence and its quality line start in the text, and the masked length

// line `i` of a file: [begin, end) without its '\n' and without one '\r' before it (lines() strips "\r\n" too)
__device__ __forceinline__ void fq_line(const FqFile &f, uint64_t i, uint32_t &b, uint32_t &e) {
    b = i ? f.nl[i - 1] + 1u : 0u;
    e = f.nl[i];
    if (e > b && f.text[e - 1] == '\r') --e;
}

// one thread per read: the spans of its sequence(s), the id span (first file), the read's size for the classification kernel's LDS
__global__ void k_fq_records(FqFile f0, FqFile f1, int n_files, uint32_t quality, uint32_t k, uint32_t stride_d, FqStats *st, FqSpan *span,
                             uint64_t *seq_len, uint32_t *id_begin, uint64_t *id_len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = st->n_rec;
    uint32_t bytes = 0, win = 0, longest = 0;
    bool err = false;
    if (r < n) {
        for (int f = 0; f < n_files; ++f) {
            const FqFile &F = f ? f1 : f0;
            uint32_t b0, e0, b1, e1, b3, e3;
            fq_line(F, 4 * r, b0, e0);
            fq_line(F, 4 * r + 1, b1, e1);
            fq_line(F, 4 * r + 3, b3, e3);
            const uint32_t slen = e1 - b1, qlen = e3 - b3;
            // seq::qual_mask (seq.rs:36-56): -Q 0 keeps the sequence; otherwise one base per quality character — the result has the
            // quality line's length, and a sequence shorter than it is the reference's "could not get the next nt" panic
            const uint32_t out = quality ? qlen : slen;
            if (quality && slen < qlen) err = true;
            span[r * n_files + f] = FqSpan{b1, b3, out};
            seq_len[r * n_files + f] = out;
            if (f == 0) { id_begin[r] = b0; id_len[r] = (uint64_t)(e0 - b0) + 1; }   // + the terminating NUL
            bytes += out;
            longest = out > longest ? out : longest;
            win += out >= k ? (out - k) / stride_d + 1 : 0;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t ob = __shfl_xor(bytes, d, 64), ow = __shfl_xor(win, d, 64), ol = __shfl_xor(longest, d, 64);
        bytes = ob > bytes ? ob : bytes;
        win = ow > win ? ow : win;
        longest = ol > longest ? ol : longest;
    }
    if ((threadIdx.x & 63) == 0) {
        if (longest) atomicMax(&st->max_seq, longest);
        if (bytes) atomicMax(&st->max_bytes, bytes);
        if (win) atomicMax(&st->max_win, win);
    }
    if (err) atomicOr(&st->err, 1u);
}

// one wave per sequence: its bases, quality-masked, to bases[seq_off[s] ..)
__global__ __launch_bounds__(256) void k_fq_pack(FqFile f0, FqFile f1, int n_files, uint32_t quality, uint64_t n_seqs, const FqSpan *span,
                                                 const uint64_t *seq_off, uint8_t *bases) {
    const int lane = threadIdx.x & 63;
    const uint8_t maxq = (uint8_t)(quality + 33);
    for (uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n_seqs; s += (uint64_t)gridDim.x * 4) {
        const FqSpan sp = span[s];
        const uint8_t *text = (n_files == 2 && (s & 1)) ? f1.text : f0.text;
        uint8_t *out = bases + seq_off[s];
        for (uint32_t j = lane; j < sp.len; j += 64) {
            uint8_t b = text[sp.seq + j];
            if (quality && text[sp.qual + j] < maxq) b = 'N';
            out[j] = b;
        }
    }
}
// one wave per read: its id line + NUL to ids[id_off[r] ..)
__global__ __launch_bounds__(256) void k_fq_ids(FqFile f0, uint64_t n_reads, const uint32_t *id_begin, const uint64_t *id_off, uint8_t *ids) {
    const int lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_reads; r += (uint64_t)gridDim.x * 4) {
        const uint64_t o = id_off[r], n = id_off[r + 1] - o - 1;
        const uint8_t *src = f0.text + id_begin[r];
        for (uint64_t j = lane; j < n; j += 64) ids[o + j] = src[j];
        if (lane == 0) ids[o + n] = 0;
    }
}
__global__ void k_fq_read_seq0(uint64_t *read_seq0, uint64_t n_reads, uint32_t n_files) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) read_seq0[r] = r * n_files;
}

// start-up: this unit's code object (its own kernels + rocPRIM's scans and selects) loaded ahead of the first stretch
hipError_t warm_fastq() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_fq_read_seq0));
}

}  // namespace cid

struct cid_fastq {
    cid_ctx *ctx = nullptr;
    int n_files = 1;
    uint64_t n_classify_steps = 0;
    uint32_t quality = 0;
    // block-gzip pushes are inflated on a stream of their own, ahead of the classification of the stretch before: a stream of
    // DEFLATE is decoded serially (one lane), so a launch takes ~14 ms however few members it holds — time the other stream fills
    hipStream_t inflate_streams[2] = {nullptr, nullptr};   // pushes alternate: a launch's floor is the decoding time of ONE member, two launches overlap theirs
    uint64_t n_inflates = 0;
    hipStream_t text_stream = nullptr;      // host-inflated text travels beside the inflate kernels, not behind them
    struct Staged {           // one push_bgzf: its text (members' texts back to back) once `done` has fired
        uint8_t *text = nullptr;
        size_t bytes = 0;
        uint32_t *d_status = nullptr;
        size_t n_members = 0;
        void *d_in = nullptr, *d_mem = nullptr, *d_scratch = nullptr;
        void *h_mem = nullptr;    // the members' table on the host: its copy may still be on the way when push_bgzf returns (CID_FASTQ_KEEP)
        hipEvent_t done = nullptr;
        bool last = false;
    };
    struct File {
        uint8_t *text = nullptr;       // carry (the unfinished tail of the stretch before) + the text appended since
        size_t cap = 0, len = 0;
        bool last = false, push_closed = false;
        // pairs: the mate file has ended and every record of it has been paired — what this file still holds or pushes has no mate and
        // is dropped instead of piling up on the device (a truncated R2 beside a whole R1; the reference's walk ends with the shorter
        // file, read_id_mt_pe.rs:727-760, kmer.rs:596-650)
        bool surplus = false;
        std::deque<Staged> staged;
        hipEvent_t copy_pending = nullptr;   // the H2D of a CID_FASTQ_KEEP text push still to be waited for (owned by its Staged entry)
        hipEvent_t members_pending = nullptr;   // the H2D of a CID_FASTQ_KEEP push of members: waited for (and destroyed) by the next such push
        size_t members_seen = 0;
    } f[2];
    // the last classify's results (device), fetched by cid_fastq_fetch
    uint64_t n_reads = 0, id_bytes = 0;
    uint32_t *d_nk = nullptr;
    uint8_t *d_status = nullptr, *d_ids = nullptr;
    uint64_t *d_id_off = nullptr;
    // a step in two halves (cid_fastq_classify_begin / _end): everything up to the launch of the classifier, then the report's
    // compaction — between them the caller's thread is free while the classifier runs (fetch the step before, push the next stretch)
    struct Inflight {
        bool active = false, classify = false;
        uint64_t n = 0, total_ids = 0;
        uint32_t n_colors = 0;
        uint64_t boundary[2] = {0, 0};
        bool spent[2] = {false, false};   // the file is `last` and this step pairs every record it holds
        uint32_t *report = nullptr, *nk = nullptr;
        uint8_t *status = nullptr, *ids = nullptr;
        uint64_t *id_off = nullptr;
        std::vector<void *> scratch;   // what the kernels in flight read: back to the cache in _end
    } infl;
    hipStream_t fetch_stream = nullptr;   // results leave beside the classifier of the NEXT step, not behind it
    // CID_FASTQ_TIMING=1: host time per part of a step, printed when the reader is destroyed
    bool timing = false;
    double ms_part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n_steps = 0;
};

namespace {
struct PartClock {   // adds the time since its start (or the last lap) to a part of the step
    cid_fastq *fq;
    std::chrono::steady_clock::time_point t;
    explicit PartClock(cid_fastq *f) : fq(f), t(std::chrono::steady_clock::now()) {}
    void lap(int part) {
        if (!fq->timing) return;
        const auto now = std::chrono::steady_clock::now();
        fq->ms_part[part] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
};

template <typename T>
struct Buf {   // scoped scratch from the ctx's block cache
    cid_ctx *c;
    T *p = nullptr;
    explicit Buf(cid_ctx *ctx) : c(ctx) {}
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { if (p) cid::ctx_free(c, p); }
    int alloc(size_t n) { void *q = nullptr; const int rc = cid::ctx_alloc(c, (n ? n : 1) * sizeof(T), &q); p = static_cast<T *>(q); return rc; }
    T *release() { T *q = p; p = nullptr; return q; }
};

// room for `extra` more bytes of text behind what the file holds (+ slack for the line end added at the end of the input)
int text_reserve(cid_fastq *fq, int file, size_t extra) {
    cid_fastq::File &F = fq->f[file];
    const size_t want = F.len + extra + 64;
    if (want >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "more than 4 GiB of FASTQ text in one call: push smaller stretches");
    if (want <= F.cap) return CID_OK;
    cid_ctx *c = fq->ctx;
    const size_t cap = want + want / 4;
    void *nb = nullptr;
    int rc = cid::ctx_alloc(c, cap, &nb);
    if (rc) return rc;
    if (F.len) HIP_TRY(hipMemcpyAsync(nb, F.text, F.len, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // (the old block goes back to the cache, whose next owner may write at once)
    if (F.text) cid::ctx_free(c, F.text);
    F.text = (uint8_t *)nb;
    F.cap = cap;
    return CID_OK;
}

void free_staged(cid_fastq *fq, cid_fastq::Staged &sg) {   // (after its event has been waited for on the host)
    cid_ctx *c = fq->ctx;
    cid::ctx_free(c, sg.text); cid::ctx_free(c, sg.d_status); cid::ctx_free(c, sg.d_in); cid::ctx_free(c, sg.d_mem);
    if (sg.d_scratch) cid::ctx_free(c, sg.d_scratch);
    free(sg.h_mem);
    if (sg.done) (void)hipEventDestroy(sg.done);
    sg = cid_fastq::Staged();
}

void drop_results(cid_fastq *fq) {
    cid_ctx *c = fq->ctx;
    cid::ctx_free(c, fq->d_nk); fq->d_nk = nullptr;
    cid::ctx_free(c, fq->d_status); fq->d_status = nullptr;
    cid::ctx_free(c, fq->d_ids); fq->d_ids = nullptr;
    cid::ctx_free(c, fq->d_id_off); fq->d_id_off = nullptr;
    fq->n_reads = fq->id_bytes = 0;
}

void drop_inflight(cid_fastq *fq) {   // (after the stream has drained)
    cid_ctx *c = fq->ctx;
    cid_fastq::Inflight &in = fq->infl;
    for (void *p : in.scratch) cid::ctx_free(c, p);
    cid::ctx_free(c, in.report); cid::ctx_free(c, in.nk); cid::ctx_free(c, in.status); cid::ctx_free(c, in.ids); cid::ctx_free(c, in.id_off);
    in = cid_fastq::Inflight();
}

}  // namespace

extern "C" {

int cid_fastq_create(cid_ctx *c, int n_files, uint32_t quality, cid_fastq **out) {
    if (!c || !out) return fail(CID_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_files != 1 && n_files != 2) return fail(CID_ERR_INVALID, "1 file (single-end) or 2 (read pairs)");
    if (quality > 93) return fail(CID_ERR_INVALID, "quality %u: phred + 33 must stay a printable character", quality);
    cid_fastq *fq = new (std::nothrow) cid_fastq();
    if (!fq) return fail(CID_ERR_NOMEM, "fastq");
    fq->ctx = c; fq->n_files = n_files; fq->quality = quality;
    fq->timing = c->tune.fastq_timing;
    hipStream_t side[4];
    if (cid::ctx_side_streams(c, side) != hipSuccess) {
        delete fq;
        return fail(CID_ERR_HIP, "stream creation failed");
    }
    fq->inflate_streams[0] = side[0]; fq->inflate_streams[1] = side[1]; fq->text_stream = side[2]; fq->fetch_stream = side[3];
    *out = fq;
    return CID_OK;
}

void cid_fastq_destroy(cid_fastq *fq) {
    if (!fq) return;
    if (fq->timing)
        fprintf(stderr, "cid_fastq: %llu steps; host ms: waiting for the inflate %.1f, member checks + text append %.1f, line ends + record count %.1f, "
                "records + scans %.1f, pack + launch of the classifier %.1f, report compaction + drain %.1f, carry %.1f\n", (unsigned long long)fq->n_steps,
                fq->ms_part[0], fq->ms_part[1], fq->ms_part[2], fq->ms_part[3], fq->ms_part[4], fq->ms_part[5], fq->ms_part[6]);
    cid_ctx *c = fq->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (hipStream_t st : fq->inflate_streams) if (st) (void)hipStreamSynchronize(st);
    if (fq->text_stream) (void)hipStreamSynchronize(fq->text_stream);
    if (fq->fetch_stream) (void)hipStreamSynchronize(fq->fetch_stream);
    drop_results(fq);
    drop_inflight(fq);
    for (int i = 0; i < 2; ++i) {
        if (fq->f[i].members_pending) (void)hipEventDestroy(fq->f[i].members_pending);   // (the streams have drained)
        cid::ctx_free(c, fq->f[i].text);
        for (cid_fastq::Staged &sg : fq->f[i].staged) free_staged(fq, sg);
    }
    delete fq;   // (the side streams are the context's)
}

// A push of either kind waits its turn as a Staged entry: the bytes travel on the reader's own stream, the text is appended by the
// classify call that takes it — so host-inflated text and device-inflated members of one file may alternate in any order.
int cid_fastq_push_text(cid_fastq *fq, int file, const uint8_t *text, size_t n_bytes, int flags) {
    if (!fq || file < 0 || file >= fq->n_files || (n_bytes && !text)) return fail(CID_ERR_INVALID, "bad argument");
    const int last = flags & CID_FASTQ_LAST;
    cid_fastq::File &F = fq->f[file];
    if (F.copy_pending) {   // the CID_FASTQ_KEEP push before this one: its buffer is the caller's again from here on
        const hipError_t e = hipEventSynchronize(F.copy_pending);
        F.copy_pending = nullptr;
        if (e != hipSuccess) return fail(CID_ERR_HIP, "cid_fastq_push_text: %s", hipGetErrorString(e));
    }
    if (F.push_closed) return fail(CID_ERR_STATE, "file %d was closed (last) by an earlier push", file);
    if (F.surplus) { F.push_closed = last != 0; if (last) F.last = true; return CID_OK; }   // no mates left for it: nothing travels
    if (F.staged.size() >= 64) return fail(CID_ERR_STATE, "file %d: 64 pushes are waiting for classify calls", file);
    if (n_bytes >= (1ull << 32)) return fail(CID_ERR_UNSUPPORTED, "a push is limited to 4 GiB of text");
    cid_ctx *c = fq->ctx;
    HIP_TRY(hipSetDevice(c->device));
    cid_fastq::Staged sg;
    sg.bytes = n_bytes; sg.last = last != 0;
    if (n_bytes) {
        void *p = nullptr;
        const int rc = cid::ctx_alloc(c, n_bytes + 16, &p);
        if (rc) return rc;
        sg.text = (uint8_t *)p;
        hipEvent_t behind = cid::ctx_event(c, 0);
        hipError_t e = hipEventRecord(behind, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(fq->text_stream, behind, 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sg.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMemcpyAsync(sg.text, text, n_bytes, hipMemcpyHostToDevice, fq->text_stream);
        if (e == hipSuccess) e = hipEventRecord(sg.done, fq->text_stream);
        if (e == hipSuccess && (flags & CID_FASTQ_KEEP)) F.copy_pending = sg.done;   // (page-locked source: the copy runs on beside the caller)
        else if (e == hipSuccess) e = hipEventSynchronize(sg.done);                   // the caller's buffer is free again
        if (e != hipSuccess) { free_staged(fq, sg); return fail(CID_ERR_HIP, "cid_fastq_push_text: %s", hipGetErrorString(e)); }
    }
    F.staged.push_back(sg);
    F.push_closed = sg.last;
    return CID_OK;
}

int cid_fastq_push_bgzf(cid_fastq *fq, int file, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                        const uint32_t *text_len, size_t n_members, int flags) {
    const int last = flags & CID_FASTQ_LAST;
    if (!fq || file < 0 || file >= fq->n_files) return fail(CID_ERR_INVALID, "bad argument");
    if (n_members && (!members || !member_off || !member_len || !text_len)) return fail(CID_ERR_INVALID, "null argument");
    cid_fastq::File &F = fq->f[file];
    if (F.members_pending) {   // the CID_FASTQ_KEEP push of members before this one: its buffer is the caller's again from here on
        const hipError_t e = hipEventSynchronize(F.members_pending);
        (void)hipEventDestroy(F.members_pending);
        F.members_pending = nullptr;
        if (e != hipSuccess) return fail(CID_ERR_HIP, "cid_fastq_push_bgzf: %s", hipGetErrorString(e));
    }
    if (F.push_closed) return fail(CID_ERR_STATE, "file %d was closed (last) by an earlier push", file);
    if (F.surplus) { F.push_closed = last != 0; if (last) F.last = true; return CID_OK; }   // no mates left for it: nothing travels
    if (F.staged.size() >= 64) return fail(CID_ERR_STATE, "file %d: 64 pushes are waiting for classify calls", file);
    if (n_bytes >= (1ull << 32) || n_members >= (1ull << 31)) return fail(CID_ERR_UNSUPPORTED, "a push of BGZF members is limited to 4 GiB");
    cid_ctx *c = fq->ctx;
    HIP_TRY(hipSetDevice(c->device));
    cid::BgzfMember *mem = static_cast<cid::BgzfMember *>(malloc((n_members ? n_members : 1) * sizeof(cid::BgzfMember)));
    if (!mem) return fail(CID_ERR_NOMEM, "cid_fastq_push_bgzf: the members' table");
    uint64_t text_total = 0;
    for (size_t i = 0; i < n_members; ++i) {
        if ((uint64_t)member_off[i] + member_len[i] > n_bytes) { free(mem); return fail(CID_ERR_INVALID, "member %zu lies outside the push", i); }
        if (text_len[i] > 65536u) { free(mem); return fail(CID_ERR_INVALID, "member %zu: more than 64 KiB of text", i); }
        mem[i] = cid::BgzfMember{member_off[i], member_len[i], (uint32_t)text_total, text_len[i]};
        text_total += text_len[i];
    }
    if (text_total >= (1ull << 32)) { free(mem); return fail(CID_ERR_UNSUPPORTED, "a push of BGZF members is limited to 4 GiB of text"); }
    cid_fastq::Staged sg;
    sg.bytes = (size_t)text_total; sg.n_members = n_members; sg.last = last != 0;
    sg.h_mem = mem;   // (freed with the entry, after its inflate has been waited for)
    if (n_members) {
        void *p = nullptr;
        int rc;
        if ((rc = cid::ctx_alloc(c, text_total + 16, &p))) return rc;
        sg.text = (uint8_t *)p;
        if ((rc = cid::ctx_alloc(c, n_members * 4, &p))) { free_staged(fq, sg); return rc; }
        sg.d_status = (uint32_t *)p;
        if ((rc = cid::ctx_alloc(c, n_bytes + 16, &sg.d_in)) || (rc = cid::ctx_alloc(c, n_members * sizeof(cid::BgzfMember), &sg.d_mem))) { free_staged(fq, sg); return rc; }
        if (cid::ctx_alloc(c, cid::bgzf_inflate_scratch_bytes((uint32_t)n_members), &sg.d_scratch) != CID_OK) {   // refused: one lane per member (3 x slower)
            sg.d_scratch = nullptr;
            static bool said = false;
            if (!said) { said = true; fprintf(stderr, "cid_fastq_push_bgzf: no room for the wave-parallel inflate's %zu MB of match tokens (%zu members): "
                                              "inflating one member per lane; push smaller stretches\n", cid::bgzf_inflate_scratch_bytes((uint32_t)n_members) >> 20, n_members); }
        }
        // the blocks may have just come back from work queued on the ctx stream: the inflate stream starts behind it
        // (the members travel on the text stream: the inflate stream may still be busy with the push before, and the copy need not wait for it)
        hipEvent_t behind = cid::ctx_event(c, 0);
        hipError_t e = hipEventRecord(behind, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(fq->text_stream, behind, 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sg.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMemcpyAsync(sg.d_in, members, n_bytes, hipMemcpyHostToDevice, fq->text_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sg.d_mem, mem, n_members * sizeof(cid::BgzfMember), hipMemcpyHostToDevice, fq->text_stream);
        hipEvent_t copied = cid::ctx_event(c, 1);
        hipStream_t inflate_stream = c->tune.fastq_inflate_beside ? fq->inflate_streams[fq->n_inflates++ & 1] : c->stream;
        if (e == hipSuccess) e = hipEventRecord(copied, fq->text_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(inflate_stream, copied, 0);
        if (e == hipSuccess) e = cid::bgzf_inflate_launch(c, inflate_stream, (const uint8_t *)sg.d_in, (const cid::BgzfMember *)sg.d_mem, (uint32_t)n_members, sg.text,
                                                          sg.d_status, sg.d_scratch);
        if (e == hipSuccess) e = hipEventRecord(sg.done, inflate_stream);
        if (e == hipSuccess && (flags & CID_FASTQ_KEEP)) {   // (page-locked members: the copy runs on beside the caller)
            e = hipEventCreateWithFlags(&F.members_pending, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(F.members_pending, fq->text_stream);
        } else if (e == hipSuccess) e = hipEventSynchronize(copied);   // the caller's buffers are free again; the kernel runs on
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(inflate_stream);
            free_staged(fq, sg);
            return fail(CID_ERR_HIP, "cid_fastq_push_bgzf: %s", hipGetErrorString(e));
        }
    }
    F.staged.push_back(sg);
    F.push_closed = sg.last;
    return CID_OK;
}

}  // extern "C"

// One step over the text pushed so far: whole records -> masked, packed reads in HBM, which go either through read_id's kernels (ix:
// per-read counts, sparse report, ids kept for cid_fastq_fetch) or into a k-mer set (ks: `search`'s query k-mers, kmer.rs:461-510 /
// :581-655 — every read of either file, windows with an N dropped, case kept).
// First half of a step: the staged pushes' text joins the file's, lines -> records -> packed reads, and the classifier (or the k-mer
// set's extraction) is LAUNCHED; nothing here waits for it.
static int fastq_begin(cid_fastq *fq, const cid_index *ix, cid_kmerset *ks, uint32_t k, uint32_t stride_d, uint32_t start_sample, int max_pushes) {
    cid_ctx *c = fq->ctx;
    if (fq->infl.active) return fail(CID_ERR_INVALID, "cid_fastq_classify_begin: the step before has not been ended");
    int rc = ix ? cid::check_ready(c, ix) : CID_OK;
    if (rc) return rc;
    if (stride_d == 0) return fail(CID_ERR_INVALID, "stride_d must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int nf = fq->n_files;
    cid_fastq::Inflight &in = fq->infl;
    PartClock pc(fq);
    ++fq->n_steps;
    // the block-gzip pushes this call takes (per file the oldest max_pushes; <= 0: all): wait for their inflate, check every member as
    // zlib checks it (the first corrupt one is an error), append their text behind what the file holds
    for (int f = 0; f < nf; ++f) {
        cid_fastq::File &F = fq->f[f];
        for (int taken = 0; !F.staged.empty() && (max_pushes <= 0 || taken < max_pushes); ++taken) {
            cid_fastq::Staged sg = F.staged.front();
            F.staged.pop_front();
            if (sg.done) {
                if (F.copy_pending == sg.done) F.copy_pending = nullptr;
                hipError_t e = hipEventSynchronize(sg.done);
                pc.lap(0);
                if (e == hipSuccess && sg.n_members) {
                    std::vector<uint32_t> h(sg.n_members);
                    e = hipMemcpyAsync(h.data(), sg.d_status, sg.n_members * 4, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    for (size_t i = 0; e == hipSuccess && i < sg.n_members; ++i)
                        if (h[i]) {
                            const size_t at = F.members_seen + i;
                            free_staged(fq, sg);
                            return fail(CID_ERR_INVALID, "corrupt gzip member %zu of file %d: %s", at, f, cid::bgzf_status_text(h[i]));
                        }
                    F.members_seen += sg.n_members;
                }
                if (e != hipSuccess) { free_staged(fq, sg); return fail(CID_ERR_HIP, "cid_fastq_classify: %s", hipGetErrorString(e)); }
                if (F.surplus) sg.bytes = 0;   // (pushed before the mate file was known to be spent)
                if ((rc = text_reserve(fq, f, sg.bytes))) { free_staged(fq, sg); return rc; }
                if (sg.bytes) {
                    const hipError_t e2 = hipMemcpyAsync(F.text + F.len, sg.text, sg.bytes, hipMemcpyDeviceToDevice, st);
                    if (e2 != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { free_staged(fq, sg); return fail(CID_ERR_HIP, "cid_fastq_classify: text append"); }
                }
                F.len += sg.bytes;
            }
            if (sg.last) F.last = true;
            free_staged(fq, sg);
        }
    }
    pc.lap(1);
    // line ends of either text
    Buf<uint32_t> nl[2] = {Buf<uint32_t>(c), Buf<uint32_t>(c)};
    Buf<uint64_t> n_nl(c);
    Buf<cid::FqStats> stats(c);
    if ((rc = n_nl.alloc(2)) || (rc = stats.alloc(1))) return rc;
    HIP_TRY(hipMemsetAsync(n_nl.p, 0, 16, st));
    cid::FqFile F[2] = {{nullptr, nullptr, n_nl.p, 0}, {nullptr, nullptr, n_nl.p + 1, 0}};
    for (int f = 0; f < nf; ++f) {
        cid_fastq::File &src = fq->f[f];
        // how many line ends: counted first (per chunk of the text, then one scan), so that their positions take 4 bytes per LINE of
        // scratch, not 4 per byte of text
        uint64_t lines = 0;
        Buf<uint32_t> chunk_off(c);
        const uint32_t n_chunks = (uint32_t)((src.len + cid::kNlChunk - 1) / cid::kNlChunk);
        const unsigned nl_grid = (unsigned)std::min<uint64_t>((n_chunks + 4) / 4, 4096);
        if (src.len) {
            if ((rc = chunk_off.alloc((size_t)n_chunks + 1))) return rc;
            hipLaunchKernelGGL(cid::k_nl_count, dim3(nl_grid), dim3(256), 0, st, src.text, (uint32_t)src.len, n_chunks, chunk_off.p);
            Buf<uint64_t> scan_state(c);
            if ((rc = scan_state.alloc(cid::scan_state_words((size_t)n_chunks + 1)))) return rc;
            HIP_TRY(cid::scan_launch(cid::ScanInU32{chunk_off.p}, cid::ScanOutU32{chunk_off.p}, (size_t)n_chunks + 1, scan_state.p, st));
            uint32_t total = 0;
            HIP_TRY(hipMemcpyAsync(&total, chunk_off.p + n_chunks, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            lines = total;
        }
        if ((rc = nl[f].alloc(lines + 2))) return rc;
        F[f] = cid::FqFile{src.text, nl[f].p, n_nl.p + f, (uint32_t)src.len};
        if (src.len) {
            hipLaunchKernelGGL(cid::k_nl_positions, dim3(nl_grid), dim3(256), 0, st, src.text, (uint32_t)src.len, n_chunks, chunk_off.p, nl[f].p, n_nl.p + f);
            HIP_TRY(hipGetLastError());   // (chunk_off returns to the block cache: whatever takes it next runs behind this kernel on the stream)
            if (src.last) hipLaunchKernelGGL(cid::k_fq_tail, dim3(1), dim3(64), 0, st, src.text, (uint32_t)src.len, nl[f].p, n_nl.p + f);
        }
    }
    hipLaunchKernelGGL(cid::k_fq_nrec, dim3(1), dim3(64), 0, st, F[0], F[1], nf, stats.p);
    HIP_TRY(hipGetLastError());
    cid::FqStats hs;
    HIP_TRY(hipMemcpyAsync(&hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    pc.lap(2);
    const uint64_t n = hs.n_rec, n_seqs = n * (uint64_t)nf;
    uint64_t total_bases = 0, total_ids = 0;
    auto keep = [&in](auto &buf) { if (buf.p) in.scratch.push_back(buf.release()); };
    if (n) {
        if (n >= (1ull << 31)) return fail(CID_ERR_UNSUPPORTED, "more than 2^31 reads in one call: push smaller stretches");
        Buf<cid::FqSpan> span(c);
        Buf<uint64_t> seq_off(c), read_seq0(c), id_off(c);
        Buf<uint32_t> id_begin(c);
        if ((rc = span.alloc(n_seqs)) || (rc = seq_off.alloc(n_seqs + 1)) || (rc = read_seq0.alloc(n + 1)) || (rc = id_off.alloc(n + 1)) ||
            (rc = id_begin.alloc(n)))
            return rc;
        HIP_TRY(hipMemsetAsync(seq_off.p + n_seqs, 0, 8, st));
        HIP_TRY(hipMemsetAsync(id_off.p + n, 0, 8, st));
        hipLaunchKernelGGL(cid::k_fq_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, F[0], F[1], nf, fq->quality, k, stride_d, stats.p,
                           span.p, seq_off.p, id_begin.p, id_off.p);
        Buf<uint64_t> scan_a(c), scan_b(c);
        if ((rc = scan_a.alloc(cid::scan_state_words(n_seqs + 1))) || (rc = scan_b.alloc(cid::scan_state_words(n + 1)))) return rc;
        HIP_TRY(cid::scan_launch(cid::ScanInU64{seq_off.p}, cid::ScanOutU64{seq_off.p, 0ull}, n_seqs + 1, scan_a.p, st));
        HIP_TRY(cid::scan_launch(cid::ScanInU64{id_off.p}, cid::ScanOutU64{id_off.p, 0ull}, n + 1, scan_b.p, st));
        HIP_TRY(hipMemcpyAsync(&hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&total_bases, seq_off.p + n_seqs, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&total_ids, id_off.p + n, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        pc.lap(3);
        if (hs.err) return fail(CID_ERR_INVALID, "ERROR: could not get the next nt in the sequence (a quality line longer than its sequence, src/seq.rs:43-45)");
        Buf<uint8_t> bases(c), ids(c), status(c);
        Buf<uint32_t> report(c), nk(c);
        unsigned grid = (unsigned)((n_seqs + 3) / 4);
        if (grid > 16384) grid = 16384;
        if (ks) {   // the reads' k-mers into the set: no ids, no report
            if ((rc = bases.alloc(total_bases + 16))) return rc;
            hipLaunchKernelGGL(cid::k_fq_pack, dim3(grid), dim3(256), 0, st, F[0], F[1], nf, fq->quality, n_seqs, span.p, seq_off.p, bases.p);
            HIP_TRY(hipGetLastError());
            if ((rc = cid_kmerset_add_seqs_dev(ks, bases.p, seq_off.p, n_seqs, hs.max_seq, 1))) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            pc.lap(4);
        } else {
        const size_t C1 = (size_t)ix->n_colors + 1;
        if ((double)n * (double)C1 * 4.0 > 64.0 * (double)(1ull << 30))
            return fail(CID_ERR_UNSUPPORTED, "%llu reads x %u colours need more than 64 GiB of dense report rows on the device: push smaller stretches",
                        (unsigned long long)n, ix->n_colors);
        if ((rc = bases.alloc(total_bases + 16)) || (rc = ids.alloc(total_ids)) || (rc = report.alloc(n * C1)) || (rc = nk.alloc(n)) ||
            (rc = status.alloc(n)))
            return rc;
        hipLaunchKernelGGL(cid::k_fq_pack, dim3(grid), dim3(256), 0, st, F[0], F[1], nf, fq->quality, n_seqs, span.p, seq_off.p, bases.p);
        hipLaunchKernelGGL(cid::k_fq_ids, dim3(grid), dim3(256), 0, st, F[0], n, id_begin.p, id_off.p, ids.p);
        hipLaunchKernelGGL(cid::k_fq_read_seq0, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, read_seq0.p, n, (uint32_t)nf);
        HIP_TRY(hipGetLastError());
        // a6-a10 on the packed batch: records of any length (cid_readid_count_dev routes the long ones through cid_readlong.hip, whose
        // work lists are made on the device from seq_off / read_seq0 — offsets that never exist on the host)
        rc = cid_readid_count_dev(c, ix, bases.p, seq_off.p, read_seq0.p, n, stride_d, start_sample, hs.max_bytes, hs.max_win ? hs.max_win : 1,
                                  report.p, nk.p, status.p);
        if (rc) return rc;
        pc.lap(4);
        in.classify = true;
        in.n_colors = ix->n_colors;
        in.report = report.release(); in.nk = nk.release(); in.status = status.release(); in.ids = ids.release(); in.id_off = id_off.release();
        }
        // what the kernels in flight still read stays until _end has seen the stream drain
        keep(span); keep(seq_off); keep(read_seq0); keep(id_off); keep(id_begin); keep(scan_a); keep(scan_b); keep(bases); keep(ids); keep(status); keep(report); keep(nk);
    }
    keep(nl[0]); keep(nl[1]); keep(n_nl); keep(stats);
    in.active = true;
    in.n = n; in.total_ids = total_ids;
    for (int f = 0; f < nf; ++f) {
        in.boundary[f] = hs.boundary[f];
        in.spent[f] = fq->f[f].last && fq->f[f].staged.empty() && hs.file_rec[f] == hs.n_rec;
    }
    return CID_OK;
}

// Second half: the report's sparse rows (the classifier has to be through for their sizes), the results published for cid_fastq_fetch —
// those of the step before are dropped here, not in _begin — and the unfinished tail of either text carried to the front.
static int fastq_end(cid_fastq *fq, uint64_t *n_reads, uint64_t *n_entries, uint64_t *id_bytes) {
    *n_reads = *n_entries = *id_bytes = 0;
    cid_ctx *c = fq->ctx;
    cid_fastq::Inflight &in = fq->infl;
    if (!in.active) return fail(CID_ERR_INVALID, "cid_fastq_classify_end without a begin");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int nf = fq->n_files;
    int rc = CID_OK;
    PartClock pc(fq);
    const uint64_t n = in.n;
    if (in.classify) {
        if (fq->fetch_stream) HIP_TRY(hipStreamSynchronize(fq->fetch_stream));   // (a fetch of the step before never outlives its call; belt and braces)
        drop_results(fq);
        cid::ctx_free(c, c->sp_start); c->sp_start = nullptr;
        cid::ctx_free(c, c->sp_col); c->sp_col = nullptr;
        cid::ctx_free(c, c->sp_cnt); c->sp_cnt = nullptr;
        c->sp_rows = 0; c->sp_entries = 0;
        rc = cid::compact_report(c, in.report, in.n_colors + 1, n, &c->sp_start, &c->sp_col, &c->sp_cnt, &c->sp_entries);
        if (rc) { (void)hipStreamSynchronize(st); drop_inflight(fq); return rc; }
        c->sp_rows = n;
        fq->d_nk = in.nk; fq->d_status = in.status; fq->d_ids = in.ids; fq->d_id_off = in.id_off;
        in.nk = nullptr; in.status = nullptr; in.ids = nullptr; in.id_off = nullptr;
        fq->n_reads = n; fq->id_bytes = in.total_ids;
    } else {
        drop_results(fq);
        c->sp_rows = 0; c->sp_entries = 0;
    }
    HIP_TRY(hipStreamSynchronize(st));   // the scratch of the step returns to the cache
    pc.lap(5);
    uint64_t boundary[2] = {in.boundary[0], in.boundary[1]};
    const bool spent[2] = {in.spent[0], in.spent[1]};
    const uint64_t total_ids = in.total_ids;
    const bool classified = in.classify;
    drop_inflight(fq);
    // what is left of either text moves to the front: the next push continues behind it
    for (int f = 0; f < nf; ++f) {
        cid_fastq::File &src = fq->f[f];
        if (boundary[f] > src.len) boundary[f] = src.len;   // (the line end added at the end of the input sits AT the length)
        const size_t left = src.len - (size_t)boundary[f];
        if (boundary[f] && left) {
            Buf<uint8_t> tmp(c);
            if ((rc = tmp.alloc(left))) return rc;
            HIP_TRY(hipMemcpyAsync(tmp.p, src.text + boundary[f], left, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(src.text, tmp.p, left, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        src.len = left;
    }
    // at the end of EVERY input what is left — lines that do not complete a record; for pairs the longer file's extra records — is
    // dropped: the line loops never push it (read_id_mt_pe.rs:862-895, :927-975: the walk ends with the shorter file)
    bool all_last = true;
    for (int f = 0; f < nf; ++f) all_last = all_last && fq->f[f].last;
    if (all_last) for (int f = 0; f < nf; ++f) fq->f[f].len = 0;
    if (nf == 2)
        for (int f = 0; f < 2; ++f)
            if (spent[f]) { fq->f[1 - f].surplus = true; fq->f[1 - f].len = 0; }
    pc.lap(6);
    *n_reads = n;
    *n_entries = classified ? c->sp_entries : 0;
    *id_bytes = classified ? total_ids : 0;
    return CID_OK;
}

static int fastq_step(cid_fastq *fq, const cid_index *ix, cid_kmerset *ks, uint32_t k, uint32_t stride_d, uint32_t start_sample, int max_pushes,
                      uint64_t *n_reads, uint64_t *n_entries, uint64_t *id_bytes) {
    *n_reads = *n_entries = *id_bytes = 0;
    const int rc = fastq_begin(fq, ix, ks, k, stride_d, start_sample, max_pushes);
    return rc ? rc : fastq_end(fq, n_reads, n_entries, id_bytes);
}

extern "C" {

int cid_fastq_classify(cid_fastq *fq, const cid_index *ix, uint32_t stride_d, uint32_t start_sample, int max_pushes, uint64_t *n_reads,
                       uint64_t *n_entries, uint64_t *id_bytes) {
    if (!fq || !ix || !n_reads || !n_entries || !id_bytes) return fail(CID_ERR_INVALID, "null argument");
    return fastq_step(fq, ix, nullptr, ix->k, stride_d, start_sample, max_pushes, n_reads, n_entries, id_bytes);
}

int cid_fastq_classify_begin(cid_fastq *fq, const cid_index *ix, uint32_t stride_d, uint32_t start_sample, int max_pushes) {
    if (!fq || !ix) return fail(CID_ERR_INVALID, "null argument");
    if (fq->ctx->tune.fastq_refuse_at_step >= 0 && (long)fq->n_classify_steps == fq->ctx->tune.fastq_refuse_at_step) {   // (cid_ctx_tune: tests of the callers' way out)
        ++fq->n_classify_steps;
        return fail(CID_ERR_UNSUPPORTED, "cid_ctx_tune fastq_refuse_at_step: this step is refused");
    }
    ++fq->n_classify_steps;
    return fastq_begin(fq, ix, nullptr, ix->k, stride_d, start_sample, max_pushes);
}

int cid_fastq_classify_end(cid_fastq *fq, uint64_t *n_reads, uint64_t *n_entries, uint64_t *id_bytes) {
    if (!fq || !n_reads || !n_entries || !id_bytes) return fail(CID_ERR_INVALID, "null argument");
    return fastq_end(fq, n_reads, n_entries, id_bytes);
}

int cid_fastq_count_kmers(cid_fastq *fq, cid_kmerset *ks, int max_pushes, uint64_t *n_reads) {
    if (!fq || !ks || !n_reads) return fail(CID_ERR_INVALID, "null argument");
    if (cid::kmerset_ctx(ks) != fq->ctx) return fail(CID_ERR_INVALID, "the k-mer set and the reader belong to different contexts");
    uint64_t ne = 0, idb = 0;
    return fastq_step(fq, nullptr, ks, cid::kmerset_k(ks), 1, 0, max_pushes, n_reads, &ne, &idb);
}

int cid_fastq_fetch(cid_fastq *fq, uint32_t *n_kmers, uint8_t *status, uint64_t *row_start, uint32_t *colours, uint32_t *counts, uint64_t *id_off,
                    char *ids) {
    if (!fq || !row_start) return fail(CID_ERR_INVALID, "null argument");
    cid_ctx *c = fq->ctx;
    if (fq->n_reads == 0) { row_start[0] = 0; return CID_OK; }
    if (!n_kmers || !status || !id_off || !ids) return fail(CID_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    // _end has seen the ctx's stream drain: the results are complete, and they travel on a stream of their own so that a classifier
    // launched since (cid_fastq_classify_begin of the next step) is not waited for
    const uint64_t n = fq->n_reads;
    if (c->sp_rows != n) return fail(CID_ERR_INVALID, "cid_fastq_fetch: another sparse call has replaced this step's report rows");
    if (c->sp_entries && (!colours || !counts)) return fail(CID_ERR_INVALID, "null argument");
    hipStream_t fs = fq->fetch_stream;
    HIP_TRY(hipMemcpyAsync(n_kmers, fq->d_nk, n * 4, hipMemcpyDeviceToHost, fs));
    HIP_TRY(hipMemcpyAsync(status, fq->d_status, n, hipMemcpyDeviceToHost, fs));
    HIP_TRY(hipMemcpyAsync(id_off, fq->d_id_off, (n + 1) * 8, hipMemcpyDeviceToHost, fs));
    HIP_TRY(hipMemcpyAsync(ids, fq->d_ids, fq->id_bytes, hipMemcpyDeviceToHost, fs));
    HIP_TRY(hipMemcpyAsync(row_start, c->sp_start, (n + 1) * 8, hipMemcpyDeviceToHost, fs));
    if (c->sp_entries) {
        HIP_TRY(hipMemcpyAsync(colours, c->sp_col, c->sp_entries * 4, hipMemcpyDeviceToHost, fs));
        HIP_TRY(hipMemcpyAsync(counts, c->sp_cnt, c->sp_entries * 4, hipMemcpyDeviceToHost, fs));
    }
    HIP_TRY(hipStreamSynchronize(fs));
    return CID_OK;
}

}  // extern "C"
