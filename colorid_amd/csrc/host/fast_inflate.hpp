// A streaming raw-DEFLATE (RFC 1951) decoder for the one input the CLI cannot spread over threads: single-stream gzip, the commonest
// fastq.gz (the reference decodes it with flate2's MultiGzDecoder on its one reader thread, src/read_id_mt_pe.rs:848-856,
// src/kmer.rs:469-476).  zlib's inflate decodes FASTQ text at ~1 GB/s on this host; this loop is built for nothing else than being
// faster at that: a 64-bit bit buffer refilled without a branch, an 11-bit literal/length table whose entries carry everything a step
// needs (literal byte, or length base + extra-bit count, or sub-table), up to three literals per refill, matches copied eight bytes at
// a time.  It checks what zlib checks (block types, code completeness, distances inside the text produced so far, end-of-block code
// present); the gzip container and its CRC-32 / ISIZE stay with the caller (LineReader::Impl::run_gzip).
//
// Streaming contract: run() works in whole steps (one block header, one symbol, a piece of a stored block).  It starts a step only when
// kMargin input bytes are in sight (or the caller said no more will come), so a step never runs dry in the middle, and only when a
// maximal match fits the output (kOutSlack).  The 16 bytes behind in_end must be readable (any content).  Matches reach back into what
// this stream produced before `out`: the caller keeps at least the last 32 KiB of it directly in front of `out`.
#pragma once
#include <cstdint>
#include <cstring>

namespace colorid {

class FastInflate {
  public:
    enum Result { kNeedInput = 0, kOutputFull = 1, kStreamEnd = 2, kError = -1 };
    static constexpr size_t kMargin = 1024;    // > the longest step's input (a dynamic block header: < 600 bytes)
    static constexpr size_t kOutSlack = 258 + 24;

    void reset() { bitbuf_ = 0; bitcnt_ = 0; mode_ = kHeader; final_ = false; stored_left_ = 0; produced_ = 0; error_ = ""; }
    uint64_t produced() const { return produced_; }
    const char *error() const { return error_; }

    Result run(const uint8_t *&in_io, const uint8_t *in_end, bool last_input, uint8_t *&out_io, uint8_t *out_end) {
        // the same loop compiled twice: with BMI2 (shifts and masks without flag dependencies: +7 %) where the CPU has it
        static const bool bmi2 = __builtin_cpu_supports("bmi2") != 0;
        return bmi2 ? run_bmi2(in_io, in_end, last_input, out_io, out_end) : run_plain(in_io, in_end, last_input, out_io, out_end);
    }

  private:
    Result run_plain(const uint8_t *&in_io, const uint8_t *in_end, bool last_input, uint8_t *&out_io, uint8_t *out_end) {
        return body(in_io, in_end, last_input, out_io, out_end);
    }
    __attribute__((target("bmi2"))) Result run_bmi2(const uint8_t *&in_io, const uint8_t *in_end, bool last_input, uint8_t *&out_io, uint8_t *out_end) {
        return body(in_io, in_end, last_input, out_io, out_end);
    }
    __attribute__((always_inline)) inline Result body(const uint8_t *&in_io, const uint8_t *in_end, bool last_input, uint8_t *&out_io, uint8_t *out_end) {
        const uint8_t *in = in_io;
        uint8_t *out = out_io;
        uint8_t *const out0 = out;
        uint64_t bitbuf = bitbuf_;
        uint32_t bitcnt = bitcnt_;
        Result res = kError;
        // (bits above bitcnt in bitbuf are copies of bytes at `in` and beyond: OR-ing the same bytes in again changes nothing)
        auto refill = [&]() {
            uint64_t w;
            memcpy(&w, in, 8);
            bitbuf |= w << bitcnt;
            const uint32_t adv = (63u - bitcnt) >> 3;
            in += adv;
            bitcnt += adv * 8;
        };
        auto bits = [&](uint32_t n) -> uint32_t { const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1)); bitbuf >>= n; bitcnt -= n; return v; };
        for (;;) {
            if (mode_ == kDone) { res = kStreamEnd; break; }
            // (`in` runs ahead of what has been used by the whole bytes still in the bit buffer)
            const bool in_ok = last_input ? in - (bitcnt >> 3) <= in_end : (size_t)(in_end - in) >= kMargin;
            if (!in_ok) {
                if (last_input) { error_ = "compressed data ends early"; res = kError; } else res = kNeedInput;
                break;
            }
            if ((size_t)(out_end - out) < kOutSlack) { res = kOutputFull; break; }
            if (mode_ == kHeader) {
                refill();
                final_ = bits(1) != 0;
                const uint32_t type = bits(2);
                if (type == 0) {   // stored: to the byte boundary, LEN NLEN; the whole bytes still in the buffer go back to the input
                    bits(bitcnt & 7u);
                    in -= bitcnt >> 3;
                    bitbuf = 0; bitcnt = 0;
                    if (last_input && in_end - in < 4) { error_ = "compressed data ends early"; break; }
                    const uint32_t len = in[0] | ((uint32_t)in[1] << 8), nlen = in[2] | ((uint32_t)in[3] << 8);
                    in += 4;
                    if ((len ^ 0xFFFFu) != nlen) { error_ = "invalid stored block lengths"; break; }
                    stored_left_ = len;
                    mode_ = kStored;
                } else if (type == 1) {
                    uint8_t lens[288 + 32];
                    for (int s = 0; s < 144; ++s) lens[s] = 8;
                    for (int s = 144; s < 256; ++s) lens[s] = 9;
                    for (int s = 256; s < 280; ++s) lens[s] = 7;
                    for (int s = 280; s < 288; ++s) lens[s] = 8;
                    for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
                    if (!build(lens, 288, litlen_, kLitBits, kLitEntries, true) || !build(lens + 288, 32, dist_, kDistBits, kDistEntries, false)) {
                        error_ = "internal: fixed tables"; break;
                    }
                    mode_ = kHuff;
                } else if (type == 2) {
                    const uint32_t hlit = bits(5) + 257, hdist = bits(5) + 1, hclen = bits(4) + 4;
                    if (hlit > 286 || hdist > 30) { error_ = "too many length or distance symbols"; break; }
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t pl[19] = {0};
                    refill();
                    for (uint32_t i = 0; i < hclen; ++i) { if (bitcnt < 3) refill(); pl[order[i]] = (uint8_t)bits(3); }
                    if (!build(pl, 19, pre_, kPreBits, kPreEntries, false, true)) { error_ = "invalid code lengths set"; break; }
                    uint8_t lens[286 + 30 + 140];
                    uint32_t i = 0;
                    bool bad = false;
                    while (i < hlit + hdist) {
                        refill();
                        const uint32_t e = pre_[bitbuf & ((1u << kPreBits) - 1)];
                        if ((e & 0xFFu) == 0) { bad = true; break; }
                        bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                        const uint32_t sym = e >> 16;
                        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                        uint32_t rep, val = 0;
                        if (sym == 16) { if (i == 0) { bad = true; break; } val = lens[i - 1]; rep = 3 + bits(2); }
                        else if (sym == 17) rep = 3 + bits(3);
                        else rep = 11 + bits(7);
                        if (i + rep > hlit + hdist) { bad = true; break; }
                        memset(lens + i, (int)val, rep);
                        i += rep;
                    }
                    if (bad) { error_ = "invalid code lengths"; break; }
                    if (lens[256] == 0) { error_ = "no end-of-block code"; break; }
                    if (!build(lens, hlit, litlen_, kLitBits, kLitEntries, true)) { error_ = "invalid literal/lengths set"; break; }
                    if (!build(lens + hlit, hdist, dist_, kDistBits, kDistEntries, false)) { error_ = "invalid distances set"; break; }
                    mode_ = kHuff;
                } else { error_ = "invalid block type"; break; }
                continue;
            }
            if (mode_ == kStored) {
                size_t n = stored_left_;
                if ((size_t)(in_end - in) < n) n = (size_t)(in_end - in);
                if ((size_t)(out_end - out) < n) n = (size_t)(out_end - out);
                memcpy(out, in, n);
                in += n; out += n; stored_left_ -= (uint32_t)n;
                if (stored_left_ == 0) mode_ = final_ ? kDone : kHeader;
                else if (in == in_end) {   // the rest of the block is in input the caller has yet to bring
                    if (last_input) { error_ = "compressed data ends early"; break; }
                    res = kNeedInput;
                    break;
                } else { res = kOutputFull; break; }
                continue;
            }
            // ---- kHuff: symbols until the block ends, the input margin is reached or the output is nearly full
            const uint8_t *const in_safe = last_input ? in_end + 8 : in_end - kMargin;   // (the 16 bytes behind in_end are readable)
            uint8_t *const out_safe = out_end - kOutSlack;
            // where the stream's first byte lies if it is still within a window's reach (else: any distance a code can express is fine)
            const uint8_t *const first_byte = produced_ + (uint64_t)(out - out0) >= 32768u ? out - 32768 : out - (produced_ + (uint64_t)(out - out0));
            bool block_end = false, failed = false;
            refill();
            uint32_t e = litlen_[bitbuf & ((1u << kLitBits) - 1)];   // (the entry of the next symbol is always looked up one step ahead)
            while (in <= in_safe && out <= out_safe) {
                // here: >= 56 bits buffered — a literal/length code (15) + its extra bits (5) + a distance code (15) + its extra bits (13) = 48
                if (e & kLiteral) {   // up to three literals on this refill (11 bits each at most from the first-level table)
                    bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                    *out++ = (uint8_t)(e >> 16);
                    e = litlen_[bitbuf & ((1u << kLitBits) - 1)];
                    if (!(e & kLiteral)) goto not_literal;
                    bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                    *out++ = (uint8_t)(e >> 16);
                    e = litlen_[bitbuf & ((1u << kLitBits) - 1)];
                    if (!(e & kLiteral)) goto not_literal;
                    bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                    *out++ = (uint8_t)(e >> 16);
                    refill();
                    e = litlen_[bitbuf & ((1u << kLitBits) - 1)];
                    continue;
                }
            not_literal:
                if (bitcnt < 48) refill();   // (after up to two literals of <= 11 bits: 56 - 22 = 34 left)
                if (e & kSpecial) {
                    if (e & kSubtable) {
                        bitbuf >>= kLitBits; bitcnt -= kLitBits;
                        e = litlen_[(e >> 16 & 0x1FFFu) + (uint32_t)(bitbuf & ((1u << (e >> 8 & 0xFu)) - 1))];
                        if (e & kLiteral) {
                            bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                            *out++ = (uint8_t)(e >> 16);
                            refill();
                            e = litlen_[bitbuf & ((1u << kLitBits) - 1)];
                            continue;
                        }
                        if (e & kSpecial) {   // end of block (sub-tables do not nest) or an unused code
                            if ((e & 0xFFu) == 0) { error_ = "invalid literal/length code"; failed = true; break; }
                            bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                            block_end = true;
                            break;
                        }
                    } else {
                        if ((e & 0xFFu) == 0) { error_ = "invalid literal/length code"; failed = true; break; }
                        bitbuf >>= (e & 0xFFu); bitcnt -= (e & 0xFFu);
                        block_end = true;
                        break;
                    }
                }
                {   // a length: base in bits 16..24, extra-bit count in bits 8..12; bits 0..7 count the code AND its extra bits, so the bit
                    // buffer — the one chain every step hangs on — moves once per code, and the extra bits are picked off beside it
                    const uint32_t xb = e >> 8 & 0x1Fu, tb = e & 0xFFu;
                    const uint32_t len = (e >> 16 & 0x1FFu) + ((uint32_t)(bitbuf >> (tb - xb)) & ((1u << xb) - 1));
                    bitbuf >>= tb; bitcnt -= tb;
                    uint32_t d = dist_[bitbuf & ((1u << kDistBits) - 1)];
                    if (d & kDistSub) {
                        bitbuf >>= kDistBits; bitcnt -= kDistBits;
                        d = dist_[(d >> 12 & 0x1FFFu) + (uint32_t)(bitbuf & ((1u << (d >> 8 & 0xFu)) - 1))];
                    }
                    if ((d & 0xFFu) == 0) { error_ = "invalid distance code"; failed = true; break; }
                    const uint32_t dxb = d >> 8 & 0xFu, dtb = d & 0xFFu;
                    const uint32_t dist = (d >> 12 & 0x7FFFu) + ((uint32_t)(bitbuf >> (dtb - dxb)) & ((1u << dxb) - 1));
                    bitbuf >>= dtb; bitcnt -= dtb;
                    const uint8_t *src = out - dist;
                    if (src < first_byte) { error_ = "invalid distance too far back"; failed = true; break; }
                    uint8_t *const end = out + len;
                    refill();                                        // the next symbol's entry is on its way while the match is copied
                    e = litlen_[bitbuf & ((1u << kLitBits) - 1)];
                    if (dist >= 8) {   // sixteen bytes at once cover most matches of FASTQ text (the second eight may read what the first wrote)
                        uint64_t w;
                        memcpy(&w, src, 8); memcpy(out, &w, 8);
                        memcpy(&w, src + 8, 8); memcpy(out + 8, &w, 8);
                        if (len > 16) {
                            src += 16; out += 16;
                            do { memcpy(&w, src, 8); memcpy(out, &w, 8); src += 8; out += 8; } while (out < end);
                        }
                    } else if (dist == 1) {
                        memset(out, *src, len);
                    } else {
                        do { *out++ = *src++; } while (out < end);
                    }
                    out = end;
                }
            }
            if (failed) break;
            if (block_end) mode_ = final_ ? kDone : kHeader;
            if (mode_ == kDone) {   // the container's trailer starts at the next byte boundary: whole unread bytes go back to the input
                bitcnt -= bitcnt & 7u;
                in -= bitcnt >> 3;
                bitbuf = 0; bitcnt = 0;
            }
        }
        // whole bytes still in the bit buffer go back to the input: between calls it holds fewer than eight bits, so the bytes a later step
        // hands back (a stored block's start, the end of the stream) are always bytes of the CURRENT call's input
        {
            const uint32_t whole = bitcnt >> 3;
            in -= whole;
            bitcnt -= whole * 8;
            bitbuf &= (1ull << bitcnt) - 1;
        }
        produced_ += (uint64_t)(out - out0);
        bitbuf_ = bitbuf; bitcnt_ = bitcnt;
        in_io = in; out_io = out;
        return res;
    }

    enum Mode { kHeader, kStored, kHuff, kDone };
  public:   // the tables' layout and their builder are shared with the block-parallel decoder (par_gunzip.hpp)
    static constexpr uint32_t kLitBits = 11, kDistBits = 8, kPreBits = 7;
    static constexpr uint32_t kLitEntries = 2048 + 4096, kDistEntries = 256 + 2048, kPreEntries = 128;
    // entry: bits 0..7 = bits this look-up uses up — the code's, and for lengths and distances their extra bits too (0: no such code); then one of
    //   kLiteral            bits 16..23 the byte
    //   kSpecial            end of block, or with kSubtable: bits 16..28 first entry of the sub-table, bits 8..11 its index bits
    //   (neither)           litlen: bits 16..24 length base, bits 8..12 extra bits; pre (code-length code): bits 16.. the symbol
    // distance entries: bits 12..26 base, bits 8..11 extra bits; kDistSub: bits 12..24 first entry of the sub-table, bits 8..11 its index bits
    static constexpr uint32_t kLiteral = 1u << 31, kSpecial = 1u << 30, kSubtable = 1u << 29, kDistSub = 1u << 31;

    static uint32_t rev(uint32_t code, uint32_t len) {
        uint32_t r = 0;
        for (uint32_t i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
        return r;
    }
    // the part of an entry that does not depend on the code's length; 0 with `ok` false: a symbol that must not occur in a stream
    static uint32_t entry_for(uint32_t sym, bool litlen, bool pre, bool &ok) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lextra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                           4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        ok = true;
        if (pre) return sym << 16;
        if (litlen) {
            if (sym < 256) return kLiteral | (sym << 16);
            if (sym == 256) return kSpecial;
            if (sym > 285) { ok = false; return 0; }   // 286 / 287: in the fixed code's tree only, never in a stream
            return ((uint32_t)lbase[sym - 257] << 16) | ((uint32_t)lextra[sym - 257] << 8);
        }
        if (sym > 29) { ok = false; return 0; }
        return ((uint32_t)dbase[sym] << 12) | ((uint32_t)dextra[sym] << 8);
    }
    // canonical Huffman code -> two-level table (index = the next bits of the stream, first bit lowest).  zlib's rules: over-subscribed
    // sets are refused; an incomplete set is accepted only as ONE code of length 1 (never for the code-length code); an empty distance
    // set is fine.  Entries no code reaches read "0 bits": the decoder reports them as invalid codes.
    static bool build(const uint8_t *lens, uint32_t n, uint32_t *tab, uint32_t tbits, uint32_t cap, bool litlen, bool pre = false) {
        const uint32_t invalid = litlen ? kSpecial : 0u;
        uint32_t count[16] = {0};
        for (uint32_t s = 0; s < n; ++s) count[lens[s]]++;
        for (uint32_t i = 0; i < (1u << tbits); ++i) tab[i] = invalid;
        if (count[0] == n) return !pre && !litlen;
        int left = 1;
        for (uint32_t l = 1; l < 16; ++l) {
            left <<= 1;
            left -= (int)count[l];
            if (left < 0) return false;
        }
        if (left > 0 && (pre || !(n - count[0] == 1 && count[1] == 1))) return false;
        uint32_t next_code[16];
        uint32_t code = 0;
        for (uint32_t l = 1; l < 16; ++l) { code = (code + count[l - 1] * (l > 1 ? 1u : 0u)) << 1; next_code[l] = code; }
        // the longest code behind every first-level index (long codes only)
        uint8_t sub_bits[1u << 11] = {0};
        uint16_t rcode[320];
        for (uint32_t s = 0; s < n; ++s) {
            const uint32_t l = lens[s];
            if (!l) continue;
            const uint32_t r = rev(next_code[l]++, l);
            rcode[s] = (uint16_t)r;
            if (l > tbits) {
                uint8_t &sb = sub_bits[r & ((1u << tbits) - 1)];
                if (l - tbits > sb) sb = (uint8_t)(l - tbits);
            }
        }
        uint32_t next_free = 1u << tbits;
        for (uint32_t s = 0; s < n; ++s) {
            const uint32_t l = lens[s];
            if (!l) continue;
            bool ok;
            const uint32_t body = entry_for(s, litlen, pre, ok);
            const uint32_t r = rcode[s];
            // (length and distance entries count their extra bits with the code's: see the decoding loop)
            const uint32_t extra = pre || (body & (kLiteral | kSpecial)) ? 0u : (litlen ? body >> 8 & 0x1Fu : body >> 8 & 0xFu);
            if (l <= tbits) {
                const uint32_t e = ok ? (body | (l + extra)) : invalid;
                for (uint32_t i = r; i < (1u << tbits); i += 1u << l) tab[i] = e;
                continue;
            }
            const uint32_t prefix = r & ((1u << tbits) - 1), sb = sub_bits[prefix];
            const bool is_dist = !litlen && !pre;
            const uint32_t sub_flag = is_dist ? kDistSub : kSubtable, sub_shift = is_dist ? 12u : 16u;
            if (!(tab[prefix] & sub_flag) || (tab[prefix] & 0xFFu) != tbits) {   // the first long code of this prefix makes its sub-table
                if (next_free + (1u << sb) > cap) return false;
                tab[prefix] = (is_dist ? kDistSub : (kSpecial | kSubtable)) | (next_free << sub_shift) | (sb << 8) | tbits;
                for (uint32_t i = 0; i < (1u << sb); ++i) tab[next_free + i] = invalid;
                next_free += 1u << sb;
            }
            const uint32_t start = tab[prefix] >> sub_shift & 0x1FFFu;
            const uint32_t e = ok ? (body | (l - tbits + extra)) : invalid;
            for (uint32_t i = r >> tbits; i < (1u << sb); i += 1u << (l - tbits)) tab[start + i] = e;
        }
        return true;
    }

  private:
    uint64_t bitbuf_ = 0;
    uint32_t bitcnt_ = 0;
    Mode mode_ = kHeader;
    bool final_ = false;
    uint32_t stored_left_ = 0;
    uint64_t produced_ = 0;
    const char *error_ = "";
    uint32_t litlen_[kLitEntries], dist_[kDistEntries], pre_[kPreEntries];
};

}  // namespace colorid
