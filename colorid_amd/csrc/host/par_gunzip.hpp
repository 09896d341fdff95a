// One DEFLATE stream decoded on several threads — the commonest fastq.gz is ONE gzip member, which the reference (flate2's MultiGzDecoder on
// its reader thread, src/read_id_mt_pe.rs:848-856, src/kmer.rs:469-476) and any ordinary inflate can only decode serially.
//
// The compressed bytes are cut into chunks.  Chunk 0 starts where the stream (or the piece before) stopped, with its history known.  Every
// other chunk looks for the first position in its range where a dynamic-Huffman block header starts and a whole block of printable text
// decodes (find_block), and decodes from there WITHOUT knowing the 32 KiB of text before it: its output is 16-bit, a value >= 256 standing
// for "the byte at this place of the window before me" (a marker; matches that copy markers copy them on).  A chunk stops at the first
// block boundary at or behind the start the next chunk found.  Then, in order: a chunk whose start is exactly where the accepted text
// before it ended is accepted (anything else — a false start, a chunk that found none — is decoded again from the accepted end, serially,
// with the history known); its window is the last 32 KiB before it, its markers are replaced, the text narrowed to bytes and summed
// (CRC-32) on the threads again.  Nothing is trusted that the serial decoder would not have produced: a start is a block boundary of the
// real stream or it is never used.
//
// ParallelInflate::run decodes ONE raw DEFLATE stream (a gzip member's body); the container stays with the caller.  Input comes through a
// callback, text leaves through a callback in order; what was read beyond the end of the stream is handed back (leftover()).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "fast_inflate.hpp"

namespace colorid {

// Whole blocks of a DEFLATE stream into 16-bit symbols, from any bit position that is a block boundary.  Not streaming: it stops at a
// block boundary (the one asked for, the end of the stream, or the last one it could finish with the input and output it had).
class MarkerInflate {
  public:
    enum Result { kStop = 0, kStreamEnd = 1, kNeedInput = 2, kOutputFull = 3, kError = -1 };
    struct Outcome {
        Result r = kError;
        uint64_t end_bit = 0;     // the block boundary decoding stopped at (kStreamEnd: the first bit behind the final block)
        size_t n_out = 0;         // symbols written up to that boundary
        const char *error = "";
    };
    static constexpr size_t kWindow = 32768;
    static constexpr size_t kInPad = 16;   // readable bytes behind the input

    // out[-back .. -1] is the text before (bytes, or markers 256 + i for window place i when it is unknown: back == kWindow then);
    // text_only: a literal that is not printable ASCII / tab / line end fails the decode (a wrong start dies quickly on FASTQ text)
    Outcome decode(const uint8_t *buf, size_t n_bytes, uint64_t start_bit, uint64_t stop_bit, uint16_t *out, size_t out_cap, size_t back, bool text_only,
                   size_t max_blocks = ~(size_t)0) {
        // the same loop compiled twice, as FastInflate's: with BMI2 where the CPU has it
        static const bool bmi2 = __builtin_cpu_supports("bmi2") != 0;
        return bmi2 ? decode_bmi2(buf, n_bytes, start_bit, stop_bit, out, out_cap, back, text_only, max_blocks)
                    : decode_plain(buf, n_bytes, start_bit, stop_bit, out, out_cap, back, text_only, max_blocks);
    }

    static bool text_byte(uint32_t b) { return (b >= 32 && b < 127) || b == '\n' || b == '\r' || b == '\t'; }
    struct TextTable { bool ok[256]; TextTable() { for (uint32_t b = 0; b < 256; ++b) ok[b] = text_byte(b); } bool operator[](uint32_t b) const { return ok[b]; } };
    static inline const TextTable kText{};

  private:
    Outcome decode_plain(const uint8_t *buf, size_t n_bytes, uint64_t start_bit, uint64_t stop_bit, uint16_t *out, size_t out_cap, size_t back, bool text_only,
                         size_t max_blocks) {
        return decode_body(buf, n_bytes, start_bit, stop_bit, out, out_cap, back, text_only, max_blocks);
    }
    __attribute__((target("bmi2"))) Outcome decode_bmi2(const uint8_t *buf, size_t n_bytes, uint64_t start_bit, uint64_t stop_bit, uint16_t *out, size_t out_cap,
                                                       size_t back, bool text_only, size_t max_blocks) {
        return decode_body(buf, n_bytes, start_bit, stop_bit, out, out_cap, back, text_only, max_blocks);
    }
    __attribute__((always_inline)) inline Outcome decode_body(const uint8_t *buf, size_t n_bytes, uint64_t start_bit, uint64_t stop_bit, uint16_t *out, size_t out_cap,
                                                              size_t back, bool text_only, size_t max_blocks) {
        Outcome oc;
        const uint8_t *in = buf + (start_bit >> 3);
        const uint8_t *const in_end = buf + n_bytes;
        uint64_t bitbuf = 0;
        uint32_t bitcnt = 0;
        auto refill = [&]() {
            uint64_t w;
            memcpy(&w, in, 8);
            bitbuf |= w << bitcnt;
            const uint32_t adv = (63u - bitcnt) >> 3;
            in += adv;
            bitcnt += adv * 8;
        };
        auto bits = [&](uint32_t n) -> uint32_t { const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1)); bitbuf >>= n; bitcnt -= n; return v; };
        auto bitpos = [&]() -> uint64_t { return (uint64_t)(in - buf) * 8 - bitcnt; };
        auto past_end = [&]() -> bool { return in - (bitcnt >> 3) > in_end; };   // whole bytes USED beyond the input (`in` itself runs up to 7 ahead)
        // (what is decoded from bytes the input does not hold yet is not an error of the stream: the block is decoded again with more input)
        auto fail = [&](const char *what) -> Outcome { if (past_end()) oc.r = kNeedInput; else { oc.r = kError; oc.error = what; } return oc; };
        if (in + 8 > in_end + kInPad) { oc.r = kNeedInput; oc.end_bit = start_bit; return oc; }
        refill();
        bits((uint32_t)(start_bit & 7));
        uint16_t *o = out;
        uint16_t *const out_safe = out + out_cap - 288;   // a longest match (258) and the overshoot of its eight-at-a-time copy fit behind it
        size_t blocks = 0;
        for (;;) {
            // ---- a block boundary
            const uint64_t here = bitpos();
            oc.end_bit = here;
            oc.n_out = (size_t)(o - out);
            if (here >= stop_bit || blocks >= max_blocks) { oc.r = kStop; return oc; }
            if ((here >> 3) >= n_bytes) { oc.r = kNeedInput; return oc; }
            ++blocks;
            if (bitcnt < 32) refill();
            const bool final = bits(1) != 0;
            const uint32_t type = bits(2);
            if (type == 0) {   // stored
                bits(bitcnt & 7u);
                in -= bitcnt >> 3;
                bitbuf = 0; bitcnt = 0;
                if (in + 4 > in_end) { oc.r = kNeedInput; return oc; }
                const uint32_t len = in[0] | ((uint32_t)in[1] << 8), nlen = in[2] | ((uint32_t)in[3] << 8);
                if ((len ^ 0xFFFFu) != nlen) return fail("invalid stored block lengths");
                in += 4;
                if (in + len > in_end) { oc.r = kNeedInput; return oc; }
                if (o + len > out_safe) { oc.r = kOutputFull; return oc; }
                for (uint32_t i = 0; i < len; ++i) o[i] = in[i];
                if (text_only) for (uint32_t i = 0; i < len; ++i) if (!text_byte(in[i])) return fail("not text");
                o += len; in += len;
                if (final) { oc.r = kStreamEnd; oc.end_bit = (uint64_t)(in - buf) * 8; oc.n_out = (size_t)(o - out); return oc; }
                if (in + 8 <= in_end + kInPad) refill();
                continue;
            }
            if (type == 3) return fail("invalid block type");
            if (type == 1) {
                uint8_t lens[288 + 32];
                for (int s = 0; s < 144; ++s) lens[s] = 8;
                for (int s = 144; s < 256; ++s) lens[s] = 9;
                for (int s = 256; s < 280; ++s) lens[s] = 7;
                for (int s = 280; s < 288; ++s) lens[s] = 8;
                for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
                if (!FastInflate::build(lens, 288, litlen_, FastInflate::kLitBits, FastInflate::kLitEntries, true) ||
                    !FastInflate::build(lens + 288, 32, dist_, FastInflate::kDistBits, FastInflate::kDistEntries, false)) return fail("internal: fixed tables");
            } else {
                const uint32_t hlit = bits(5) + 257, hdist = bits(5) + 1, hclen = bits(4) + 4;
                if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t pl[19] = {0};
                refill();
                for (uint32_t i = 0; i < hclen; ++i) { if (bitcnt < 3) refill(); pl[order[i]] = (uint8_t)bits(3); }
                if (!FastInflate::build(pl, 19, pre_, FastInflate::kPreBits, FastInflate::kPreEntries, false, true)) return fail("invalid code lengths set");
                uint8_t lens[286 + 30 + 140];
                uint32_t i = 0;
                while (i < hlit + hdist) {
                    if (past_end()) { oc.r = kNeedInput; return oc; }
                    refill();
                    const uint32_t e = pre_[bitbuf & ((1u << FastInflate::kPreBits) - 1)];
                    if ((e & 0xFFu) == 0) return fail("invalid code lengths");
                    bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                    const uint32_t sym = e >> 16;
                    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                    uint32_t rep, val = 0;
                    if (sym == 16) { if (i == 0) return fail("invalid code lengths"); val = lens[i - 1]; rep = 3 + bits(2); }
                    else if (sym == 17) rep = 3 + bits(3);
                    else rep = 11 + bits(7);
                    if (i + rep > hlit + hdist) return fail("invalid code lengths");
                    memset(lens + i, (int)val, rep);
                    i += rep;
                }
                if (lens[256] == 0) return fail("no end-of-block code");
                if (!FastInflate::build(lens, hlit, litlen_, FastInflate::kLitBits, FastInflate::kLitEntries, true)) return fail("invalid literal/lengths set");
                if (!FastInflate::build(lens + hlit, hdist, dist_, FastInflate::kDistBits, FastInflate::kDistEntries, false)) return fail("invalid distances set");
            }
            // ---- the block's symbols (the loop of FastInflate::body, sixteen bits a symbol: the next entry is looked up one step ahead, up
            // to three literals go on one refill, matches are copied eight symbols at a time)
            {
                refill();
                uint32_t e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                for (;;) {
                    if (past_end()) { oc.r = kNeedInput; return oc; }        // (the block is decoded again, from its boundary, when more input is there)
                    if (o > out_safe) { oc.r = kOutputFull; return oc; }
                    // here: >= 56 bits buffered
                    if (e & FastInflate::kLiteral) {
                        uint32_t b = e >> 16 & 0xFFu;
                        bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                        if (text_only && !kText[b]) return fail("not text");
                        *o++ = (uint16_t)b;
                        e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                        if (!(e & FastInflate::kLiteral)) goto not_literal;
                        b = e >> 16 & 0xFFu;
                        bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                        if (text_only && !kText[b]) return fail("not text");
                        *o++ = (uint16_t)b;
                        e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                        if (!(e & FastInflate::kLiteral)) goto not_literal;
                        b = e >> 16 & 0xFFu;
                        bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                        if (text_only && !kText[b]) return fail("not text");
                        *o++ = (uint16_t)b;
                        refill();
                        e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                        continue;
                    }
                not_literal:
                    if (bitcnt < 48) refill();
                    if (e & FastInflate::kSpecial) {
                        if (e & FastInflate::kSubtable) {
                            bitbuf >>= FastInflate::kLitBits; bitcnt -= FastInflate::kLitBits;
                            e = litlen_[(e >> 16 & 0x1FFFu) + (uint32_t)(bitbuf & ((1u << (e >> 8 & 0xFu)) - 1))];
                            if (e & FastInflate::kLiteral) {
                                const uint32_t b = e >> 16 & 0xFFu;
                                bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                                if (text_only && !kText[b]) return fail("not text");
                                *o++ = (uint16_t)b;
                                refill();
                                e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                                continue;
                            }
                            if (e & FastInflate::kSpecial) {   // end of block (sub-tables do not nest) or an unused code
                                if ((e & 0xFFu) == 0) return fail("invalid literal/length code");
                                bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                                break;
                            }
                        } else {
                            if ((e & 0xFFu) == 0) return fail("invalid literal/length code");
                            bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                            break;
                        }
                    }
                    {
                        const uint32_t xb = e >> 8 & 0x1Fu, tb = e & 0xFFu;
                        const uint32_t len = (e >> 16 & 0x1FFu) + ((uint32_t)(bitbuf >> (tb - xb)) & ((1u << xb) - 1));
                        bitbuf >>= tb; bitcnt -= tb;
                        uint32_t d = dist_[bitbuf & ((1u << FastInflate::kDistBits) - 1)];
                        if (d & FastInflate::kDistSub) {
                            bitbuf >>= FastInflate::kDistBits; bitcnt -= FastInflate::kDistBits;
                            d = dist_[(d >> 12 & 0x1FFFu) + (uint32_t)(bitbuf & ((1u << (d >> 8 & 0xFu)) - 1))];
                        }
                        if ((d & 0xFFu) == 0) return fail("invalid distance code");
                        const uint32_t dxb = d >> 8 & 0xFu, dtb = d & 0xFFu;
                        const uint32_t dist = (d >> 12 & 0x7FFFu) + ((uint32_t)(bitbuf >> (dtb - dxb)) & ((1u << dxb) - 1));
                        bitbuf >>= dtb; bitcnt -= dtb;
                        if ((size_t)dist > back + (size_t)(o - out)) return fail("invalid distance too far back");
                        const uint16_t *src = o - dist;
                        uint16_t *const end = o + len;
                        refill();                                        // the next symbol's entry is on its way while the match is copied
                        e = litlen_[bitbuf & ((1u << FastInflate::kLitBits) - 1)];
                        if (dist >= 8) {   // eight symbols (sixteen bytes) at a time; the slack behind out_safe takes the overshoot
                            do { memcpy(o, src, 16); o += 8; src += 8; } while (o < end);
                        } else {
                            do { *o++ = *src++; } while (o < end);
                        }
                        o = end;
                    }
                }
            }
            if (past_end()) { oc.r = kNeedInput; return oc; }
            if (final) {
                oc.r = kStreamEnd;
                oc.end_bit = bitpos();
                oc.n_out = (size_t)(o - out);
                return oc;
            }
        }
    }

    uint32_t litlen_[FastInflate::kLitEntries], dist_[FastInflate::kDistEntries], pre_[FastInflate::kPreEntries];
};

// the first bit position in [from_bit, to_bit) where a non-final dynamic-Huffman block header starts and the block decodes as text; ~0 if none
template <typename Scratch>
inline uint64_t find_block(const uint8_t *buf, size_t n_bytes, uint64_t from_bit, uint64_t to_bit, MarkerInflate &mi, Scratch &scratch) {
    constexpr size_t kTrial = MarkerInflate::kWindow + (1u << 22);
    scratch.need(kTrial);
    // the unknown window before the trial position holds markers, as in the real chunk decode: a match reaching back into it copies
    // defined values (the trial's verdict does not depend on them; an indeterminate read would still be one)
    for (size_t i = 0; i < MarkerInflate::kWindow; ++i) scratch.data()[i] = (uint16_t)(256 + i);
    for (uint64_t p = from_bit; p < to_bit; ++p) {
        const size_t byte = (size_t)(p >> 3);
        if (byte + 16 > n_bytes) return ~(uint64_t)0;
        uint64_t w0, w1;
        memcpy(&w0, buf + byte, 8);
        memcpy(&w1, buf + byte + 8, 8);
        const uint32_t sh = (uint32_t)(p & 7);
        const uint64_t v = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
        if ((v & 7u) != 4u) continue;   // BFINAL = 0, BTYPE = 10 (read low bit first: 0, then 0 1)
        const uint32_t hlit = (uint32_t)(v >> 3 & 31u) + 257, hdist = (uint32_t)(v >> 8 & 31u) + 1, hclen = (uint32_t)(v >> 13 & 15u) + 4;
        if (hlit > 286 || hdist > 30) continue;
        // the code-length code must be complete (zlib refuses anything else): sum of 2^(7 - len) over its codes == 128
        uint32_t kraft = 0, used = 0;
        const uint64_t tail = sh ? (w1 >> sh) : w1;   // bits 64.. of the position (17 + 3 * 19 = 74 are needed at most)
        for (uint32_t i = 0; i < hclen; ++i) {
            const uint32_t at = 17 + 3 * i;
            uint32_t l;
            if (at + 3 <= 64) l = (uint32_t)(v >> at & 7u);
            else if (at >= 64) l = (uint32_t)(tail >> (at - 64) & 7u);
            else l = (uint32_t)(((v >> at) | (tail << (64 - at))) & 7u);
            if (l) { kraft += 128u >> l; ++used; }
        }
        if (kraft != 128u || used < 2) continue;
        const MarkerInflate::Outcome oc = mi.decode(buf, n_bytes, p, p + 1, scratch.data() + MarkerInflate::kWindow, kTrial - MarkerInflate::kWindow,
                                                    MarkerInflate::kWindow, true, 1);
        if (oc.r == MarkerInflate::kStop && oc.n_out >= 64) return p;
    }
    return ~(uint64_t)0;
}

template <typename T>
struct RawBuf {   // grown, never filled: a std::vector would write zeros over every byte the decoder is about to write
    std::unique_ptr<T[]> p;
    size_t cap = 0;
    void need(size_t n) { if (n > cap) { p.reset(new T[n]); cap = n; } }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
};

class ParallelInflate {
  public:
    using Reader = std::function<size_t(uint8_t *dst, size_t cap)>;              // more compressed bytes; 0 = no more
    using Sink = std::function<bool(const uint8_t *text, size_t n)>;            // decoded text, in order; false: stop (run returns false, error() "stopped")
    using ParallelFor = std::function<void(size_t n, const std::function<void(size_t)> &)>;
    using Crc32 = std::function<uint32_t(uint32_t crc, const uint8_t *p, size_t n)>;

    // chunk_bytes of compressed input per task, n_chunks tasks per round
    ParallelInflate(size_t chunk_bytes, size_t n_chunks) : chunk_(chunk_bytes < 65536 ? 65536 : chunk_bytes), n_chunks_(n_chunks < 2 ? 2 : n_chunks) {}

    const char *error() const { return error_.c_str(); }
    uint32_t crc() const { return crc_; }
    uint64_t total() const { return total_; }
    // compressed bytes read but not part of the stream (the container's trailer, the members behind it)
    const std::vector<uint8_t> &leftover() const { return left_; }
    struct Stats { uint64_t chunks = 0, accepted = 0, serial = 0, rounds = 0; } stats;

    // `first`: compressed bytes the caller has already read (the stream starts at first[0], bit 0).  false: error() says why — nothing
    // decoded is wrong before that point, but the caller cannot continue this stream.
    bool run(const uint8_t *first, size_t n_first, const Reader &reader, const Sink &sink, const ParallelFor &pfor, const Crc32 &crc32,
             const std::function<uint32_t(uint32_t, uint32_t, uint64_t)> &crc_combine) {
        std::vector<uint8_t> in(n_first);
        if (n_first) memcpy(in.data(), first, n_first);
        bool eof = false;
        uint64_t start_bit = 0;                    // of the next block boundary, relative to in[0]
        std::vector<uint8_t> window;               // the last <= 32 KiB of the text so far
        crc_ = 0; total_ = 0; left_.clear(); error_.clear();
        const size_t round_bytes = chunk_ * n_chunks_;
        std::future<bool> delivery;   // (its destructor waits: no exit of this function leaves the thread behind)
        std::future<std::vector<uint8_t>> ahead;
        auto read_more = [&]() {   // a round's worth of compressed bytes behind `in`: what was read ahead, else from the reader now
            if (ahead.valid()) {
                const std::vector<uint8_t> v = ahead.get();
                if (v.empty()) eof = true;
                in.insert(in.end(), v.begin(), v.end());
                return;
            }
            const size_t at = in.size();
            in.resize(at + round_bytes);
            const size_t n = reader(in.data() + at, round_bytes);
            in.resize(at + n);
            if (n == 0) eof = true;
        };
        std::vector<Seg> segs(n_chunks_);
        std::vector<MarkerInflate> dec(n_chunks_);
        std::vector<RawBuf<uint16_t>> scratch(n_chunks_);
        for (;;) {
            // ---- fill: a round's worth of compressed bytes behind what is left of the round before
            while (!eof && in.size() < round_bytes + chunk_) read_more();
            // (the next round's bytes are read while this one is decoded: the file is read by one thread at 3-4 GB/s, a sixth of a round)
            if (!eof && !ahead.valid())
                ahead = std::async(std::launch::async, [&reader, round_bytes]() {
                    std::vector<uint8_t> v(round_bytes);
                    v.resize(reader(v.data(), round_bytes));
                    return v;
                });
            const size_t n_in = in.size();
            in.resize(n_in + 64, 0);   // (padding the decoders may read)
            ++stats.rounds;
            // ---- starts: chunk 0 is where we are; the others look for a block of text in their range
            const size_t first_byte = (size_t)(start_bit >> 3);
            const size_t nt = std::min(n_chunks_, std::max<size_t>(1, (n_in - first_byte + chunk_ - 1) / chunk_));
            for (size_t j = 0; j < nt; ++j) { segs[j].start = ~(uint64_t)0; segs[j].oc = MarkerInflate::Outcome(); segs[j].n = 0; }
            segs[0].start = start_bit;
            pfor(nt, [&](size_t j) {
                if (j == 0) return;
                const uint64_t lo = (uint64_t)(first_byte + j * chunk_) * 8, hi = std::min<uint64_t>((uint64_t)(first_byte + (j + 1) * chunk_) * 8, (uint64_t)n_in * 8);
                if (lo < hi) segs[j].start = find_block(in.data(), n_in, lo, hi, dec[j], scratch[j]);
            });
            // ---- decode: every chunk with a start runs to the first block boundary at or behind the next start
            pfor(nt, [&](size_t j) {
                Seg &sg = segs[j];
                if (sg.start == ~(uint64_t)0) return;
                // (the last one stops behind the round's chunks, not at the end of what has been read ahead: that is the next round's)
                uint64_t stop = std::min<uint64_t>((uint64_t)(first_byte + nt * chunk_) * 8, (uint64_t)n_in * 8 + 8);
                for (size_t k = j + 1; k < nt; ++k) if (segs[k].start != ~(uint64_t)0) { stop = segs[k].start; break; }
                if (stop <= sg.start) stop = sg.start + 1;
                const size_t span = (size_t)((std::min<uint64_t>(stop, (uint64_t)n_in * 8) - std::min<uint64_t>(sg.start, (uint64_t)n_in * 8)) >> 3) + chunk_ / 4 + 1;
                const size_t cap = span * 8 + (2u << 20);
                sg.out.need(MarkerInflate::kWindow + cap);
                uint16_t *o = sg.out.data() + MarkerInflate::kWindow;
                size_t back;
                if (j == 0) {   // the history is known
                    back = window.size();
                    for (size_t i = 0; i < back; ++i) o[-(std::ptrdiff_t)back + (std::ptrdiff_t)i] = window[i];
                } else {
                    back = MarkerInflate::kWindow;
                    for (size_t i = 0; i < MarkerInflate::kWindow; ++i) sg.out.data()[i] = (uint16_t)(256 + i);
                }
                sg.oc = dec[j].decode(in.data(), n_in, sg.start, stop, o, cap, back, j != 0);
                sg.n = sg.oc.n_out;
            });
            stats.chunks += nt;
            // ---- in order: accept, or decode again from where the accepted text ends
            uint64_t at = start_bit;
            bool stream_end = false;
            std::vector<Piece> pieces;     // the round's text, in order: (segment, its window)
            std::vector<std::vector<uint8_t>> windows;
            windows.push_back(window);
            auto take = [&](Seg &sg, bool exact) {
                pieces.push_back(Piece{&sg, windows.size() - 1, exact});
                // the window behind this piece: the last 32 KiB of (window before + this piece's text), markers replaced
                const std::vector<uint8_t> &wb = windows.back();
                std::vector<uint8_t> wn;
                const uint16_t *o = sg.out.data() + MarkerInflate::kWindow;
                const size_t keep_old = sg.n >= MarkerInflate::kWindow ? 0 : std::min(wb.size(), MarkerInflate::kWindow - sg.n);
                wn.assign(wb.end() - (std::ptrdiff_t)keep_old, wb.end());
                const size_t from = sg.n > MarkerInflate::kWindow ? sg.n - MarkerInflate::kWindow : 0;
                for (size_t i = from; i < sg.n; ++i) wn.push_back(resolve(o[i], wb));
                windows.push_back(std::move(wn));
            };
            std::deque<Seg> extra;    // serial re-decodes (a deque: the round's pieces point at them, and there may be any number)
            size_t j = 0;
            bool progress = false;
            while (j < nt && !stream_end) {
                Seg &sg = segs[j];
                // (input or buffer ran out: the text up to the last whole block counts)
                const bool partial = (sg.oc.r == MarkerInflate::kNeedInput || sg.oc.r == MarkerInflate::kOutputFull) && sg.n > 0;
                const bool usable = sg.start == at && (sg.oc.r == MarkerInflate::kStop || sg.oc.r == MarkerInflate::kStreamEnd || partial);
                if (usable) {
                    take(sg, j == 0);
                    if (j) ++stats.accepted;
                    at = sg.oc.end_bit;
                    progress = progress || sg.n > 0 || sg.oc.r == MarkerInflate::kStreamEnd;
                    if (sg.oc.r == MarkerInflate::kStreamEnd) { stream_end = true; break; }
                    if (sg.oc.r == MarkerInflate::kNeedInput) break;   // the rest of the input waits for the next round
                    ++j;
                    while (j < nt && segs[j].start != at) {
                        if (segs[j].start != ~(uint64_t)0 && segs[j].start > at) break;   // a later start: decode up to it below
                        ++j;                                                               // no start, or one we have passed
                    }
                    if (j >= nt) break;                                                    // what is left waits for the next round
                    if (segs[j].start == at) continue;
                }
                if (j == 0 && !usable && sg.oc.r == MarkerInflate::kError) { error_ = sg.oc.error; return false; }
                // (chunk 0 with a block whose text does not fit its buffer — long runs of one letter — goes to the serial branch, which grows it)
                if (j == 0 && !usable && sg.oc.r != MarkerInflate::kOutputFull) break;   // kNeedInput without a whole block: more input (or a bigger round) is needed
                // decode serially from `at` to the next start behind it (or as far as the input goes)
                uint64_t stop = ~(uint64_t)0;
                size_t k = j;
                for (; k < nt; ++k) if (segs[k].start != ~(uint64_t)0 && segs[k].start > at) { stop = segs[k].start; break; }
                extra.emplace_back();
                Seg &ex = extra.back();
                const size_t span = (size_t)(((stop == ~(uint64_t)0 ? (uint64_t)n_in * 8 : stop) - at) >> 3) + 1;
                size_t cap = span * 8 + (2u << 20);
                const std::vector<uint8_t> &wb = windows.back();
                ex.start = at;
                for (;;) {   // (a single block of long matches can hold megabytes of text: the buffer grows until one fits)
                    ex.out.need(MarkerInflate::kWindow + cap);
                    uint16_t *o = ex.out.data() + MarkerInflate::kWindow;
                    for (size_t i = 0; i < wb.size(); ++i) o[-(std::ptrdiff_t)wb.size() + (std::ptrdiff_t)i] = wb[i];
                    ex.oc = dec[0].decode(in.data(), n_in, at, stop, o, cap, wb.size(), false);
                    ex.n = ex.oc.n_out;
                    if (ex.oc.r != MarkerInflate::kOutputFull || ex.n > 0 || cap >= ((size_t)1 << 30)) break;
                    cap *= 4;
                }
                ++stats.serial;
                if (ex.oc.r == MarkerInflate::kError) { error_ = ex.oc.error; return false; }
                if (ex.oc.r == MarkerInflate::kOutputFull && ex.n == 0) { error_ = "internal: a block's text does not fit its buffer"; return false; }
                if (ex.oc.r == MarkerInflate::kNeedInput && ex.n == 0) break;
                take(ex, true);
                at = ex.oc.end_bit;
                progress = progress || ex.n > 0 || ex.oc.r == MarkerInflate::kStreamEnd;
                if (ex.oc.r == MarkerInflate::kStreamEnd) { stream_end = true; break; }
                if (ex.oc.r == MarkerInflate::kNeedInput) break;
                j = k;   // (segs[k].start >= at: equal -> accepted next; passed -> decoded over again)
                while (j < nt && segs[j].start != at && !(segs[j].start != ~(uint64_t)0 && segs[j].start > at)) ++j;
            }
            // ---- the round's text: markers replaced, narrowed, summed — on the threads — then out, in order
            // (the text of the round before may still be on its way to the sink: its buffers are written next)
            if (delivery.valid() && !delivery.get()) { error_ = "stopped"; return false; }
            if (text_.size() < pieces.size()) text_.resize(pieces.size());
            std::vector<uint32_t> sums(pieces.size());
            pfor(pieces.size(), [&](size_t i) {
                const Seg &sg = *pieces[i].seg;
                const std::vector<uint8_t> &wb = windows[pieces[i].window];
                const uint16_t *o = sg.out.data() + MarkerInflate::kWindow;
                text_[i].need(sg.n + 1);
                uint8_t *t = text_[i].data();
                if (pieces[i].exact) for (size_t x = 0; x < sg.n; ++x) t[x] = (uint8_t)o[x];
                else for (size_t x = 0; x < sg.n; ++x) t[x] = resolve(o[x], wb);
                sums[i] = crc32(0, t, sg.n);
            });
            std::vector<size_t> lens(pieces.size());
            for (size_t i = 0; i < pieces.size(); ++i) {
                lens[i] = pieces[i].seg->n;
                crc_ = crc_combine(crc_, sums[i], lens[i]);
                total_ += lens[i];
            }
            // the round's text leaves on a thread of its own (the sink may wait for whoever takes it) while the next round is decoded
            delivery = std::async(std::launch::async, [this, lens, &sink]() -> bool {
                for (size_t i = 0; i < lens.size(); ++i)
                    if (lens[i] && !sink(text_[i].data(), lens[i])) return false;
                return true;
            });
            window = windows.back();
            in.resize(n_in);
            if (stream_end) {
                if (!delivery.get()) { error_ = "stopped"; return false; }
                const size_t used = (size_t)((at + 7) >> 3);
                left_.assign(in.begin() + (std::ptrdiff_t)std::min(used, n_in), in.end());
                if (ahead.valid()) { const std::vector<uint8_t> v = ahead.get(); left_.insert(left_.end(), v.begin(), v.end()); }
                return true;
            }
            if (!progress) {
                if (eof) { error_ = "compressed data ends early"; return false; }
                if (in.size() - (size_t)(at >> 3) > (64u << 20)) { error_ = "internal: a block larger than 64 MiB"; return false; }
                // a block that does not fit the round: read more behind it
                const size_t drop0 = (size_t)(at >> 3);
                in.erase(in.begin(), in.begin() + (std::ptrdiff_t)drop0);
                start_bit = at & 7;
                read_more();
                continue;
            }
            // ---- carry: the bytes from the boundary on
            const size_t drop = (size_t)(at >> 3);
            in.erase(in.begin(), in.begin() + (std::ptrdiff_t)drop);
            start_bit = at & 7;
            if (eof && in.size() * 8 <= start_bit) { error_ = "compressed data ends early"; return false; }
        }
    }

  private:
    struct Seg {
        uint64_t start = ~(uint64_t)0;
        MarkerInflate::Outcome oc;
        size_t n = 0;
        RawBuf<uint16_t> out;        // kWindow places of history, then the text
    };
    struct Piece { const Seg *seg; size_t window; bool exact; };
    static uint8_t resolve(uint16_t v, const std::vector<uint8_t> &window_before) {
        if (v < 256) return (uint8_t)v;
        // marker 256 + i: place i of the 32 KiB before the piece, place kWindow - 1 being the byte just before it
        const size_t i = (size_t)v - 256, n = window_before.size();
        const size_t from_end = MarkerInflate::kWindow - i;   // 1 = the last byte of the window
        return from_end <= n ? window_before[n - from_end] : 0;   // (a place before the start of the stream: a valid stream never asks)
    }
    size_t chunk_, n_chunks_;
    std::vector<RawBuf<uint8_t>> text_;
    uint32_t crc_ = 0;
    uint64_t total_ = 0;
    std::vector<uint8_t> left_;
    std::string error_;
};

}  // namespace colorid
