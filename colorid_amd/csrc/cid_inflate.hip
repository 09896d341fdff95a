// Block-gzip (BGZF) members inflated on the GPU: the input side of `read_id` / `search` on fastq.gz files written by bgzip / htslib /
// Illumina's converters (SURVEY.md §8f.3 "gz decode"; the reference inflates with flate2's MultiGzDecoder on one thread,
// src/read_id_mt_pe.rs:848-856, src/kmer.rs:469-476).  A BGZF file is a series of independent gzip members of at most 64 KiB of text,
// each carrying its compressed size in a "BC" extra field — so a batch of members is a batch of independent DEFLATE streams.
//
// DEFLATE is serial inside a stream, so one lane decodes a member: its chain of dependent steps is what a launch cannot be shorter than.
// Everything that chain touches at every symbol — the Huffman tables, the bit reader's input ring — lives in LDS (7.5 KiB per member),
// and nothing in it waits for HBM: a literal is a store nobody waits for, and a match is NOT copied by the decoder (reading the source
// would wait for every earlier store — a round trip to L2 per match) but left as a token (position, length, distance) for the whole
// wave, which copies a run's matches 64 at a time behind one fence per round (copy_matches; round 3: 64 KiB of FASTQ text took a lane
// 13.7 ms with the copies in the chain, 7.9 ms without, 4.5 ms when the wave carries one member).  The wave also moves the data in:
// compressed bytes stream into the ring in 1 KiB wave-wide loads between the decoding lanes' runs; and it checks the result: a member's
// CRC-32 is computed by all 64 lanes over 1 KiB slices of the text (slicing by 4) and folded with the "append 1024 zero bytes" operator.
// ~20 members fit a CU, so the 4 800 members of a million reads are all in flight at once (the first version kept each member's 64 KiB
// image in LDS: 2 members per CU, 104 ms per million reads).  A wave carries TWO members on its first two lanes when the launch can fill
// the chip: two decoders that share an instruction stream wherever their steps coincide finish a million reads' members in 9.3 ms
// against 14.3 ms one per wave (4 and 8 per wave: 13.7 / 30.6 ms — the divergence eats the rest); a small launch takes one per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <vector>

#include "../../include/colorid_hip.h"
#include "cid_objects.hpp"

namespace cid {


constexpr int kLitBits = 10, kDistBits = 8;
constexpr uint32_t kRing = 2048;
// per-member LDS: input ring | literal/length table | dist table | canonical-decode arrays | code lengths | match tokens
constexpr uint32_t kTokens = 256;   // matches a run may leave for the wave to copy
constexpr uint32_t kLdsRing = 0, kLdsLit = kLdsRing + kRing, kLdsDist = kLdsLit + 2u * (1u << kLitBits),
                   kLdsCnt = kLdsDist + 2u * (1u << kDistBits), kLdsLens = kLdsCnt + 2u * (16 + 288 + 16 + 32), kLdsTok = kLdsLens + 320,
                   kLdsBytes = kLdsTok + 8u * kTokens;

struct CrcShift { uint32_t m[32]; };   // column j: the CRC register 1 << j after 1024 zero bytes

enum : uint32_t { ST_OK = 0, ST_BAD_HEADER = 1, ST_BAD_BLOCK = 2, ST_BAD_CODE = 3, ST_OVERRUN_IN = 4, ST_OVERRUN_OUT = 5, ST_BAD_LEN = 6, ST_BAD_CRC = 7 };

// lane 0's view of the compressed stream.  The ring is read a 32-bit word at a time and one word AHEAD of the bit buffer, so the LDS
// round trip of a refill overlaps the decoding of the bits already buffered: after refill() the buffer holds 33 .. 64 bits, more than
// any one step takes (a length code with its extra bits: 20; a distance code with its extra bits: 28).
struct BitReader {
    const uint32_t *ring;   // kRing / 4 words
    uint32_t wpos;          // the next word of the stream to load (the one before it sits in `ahead`)
    uint64_t buf;
    uint32_t cnt;
    uint32_t ahead;
    __device__ __forceinline__ void start() { wpos = 1; buf = 0; cnt = 0; ahead = ring[0]; }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            buf |= (uint64_t)ahead << cnt;
            cnt += 32;
            ahead = ring[wpos & (kRing / 4 - 1)];
            ++wpos;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }   // n <= 16
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    // bytes of the stream that are used up: everything before the first byte with a bit still in the buffer (or in `ahead`)
    __device__ __forceinline__ uint32_t consumed_bytes() const { return (wpos - 1) * 4 - cnt / 8; }
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32 - n); }

// canonical Huffman tables from code lengths: a 2^bits direct table (entry = sym << 4 | len; 0 = a longer code) and the count /
// symbol arrays of the bit-by-bit decoder for codes longer than `bits`.  false: over-subscribed or incomplete — zlib's inflate_table
// rule: an incomplete set is accepted only when it is ONE code of length 1 (max == 1), and never for the code-length code
// (`allow_single` false, its type CODES)
__device__ bool build_tables(const uint8_t *lens, uint32_t n_sym, uint16_t *tab, uint32_t bits, uint16_t *cnt, uint16_t *sym, bool allow_single = true) {
    for (uint32_t i = 0; i < 16; ++i) cnt[i] = 0;
    for (uint32_t s = 0; s < n_sym; ++s) cnt[lens[s]]++;
    for (uint32_t i = 0; i < (1u << bits); ++i) tab[i] = 0;
    if (cnt[0] == n_sym) return true;   // no codes at all (a distance tree may be empty)
    int left = 1;
    for (uint32_t l = 1; l < 16; ++l) {
        left <<= 1;
        left -= cnt[l];
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (uint32_t l = 1; l < 15; ++l) offs[l + 1] = offs[l] + cnt[l];
    uint32_t next_code[16];
    uint32_t code = 0;   // RFC 1951 3.2.2: code = (code + bl_count[bits-1]) << 1, with bl_count[0] = 0
    for (uint32_t l = 1; l < 16; ++l) { code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1; next_code[l] = code; }
    for (uint32_t s = 0; s < n_sym; ++s) {
        const uint32_t l = lens[s];
        if (!l) continue;
        sym[offs[l]++] = (uint16_t)s;
        const uint32_t c = next_code[l]++;
        if (l <= bits) {
            const uint32_t r = rev_bits(c, l);
            for (uint32_t i = r; i < (1u << bits); i += 1u << l) tab[i] = (uint16_t)((s << 4) | l);
        }
    }
    return left == 0 || (allow_single && (n_sym - cnt[0]) == 1 && cnt[1] == 1);
}

// one symbol: direct table, else bit by bit over the canonical code (puff-style); -1 = invalid code
template <typename Reader>
__device__ __forceinline__ int decode_sym(Reader &br, const uint16_t *tab, uint32_t bits, const uint16_t *cnt, const uint16_t *sym) {
    const uint32_t e = tab[br.peek(bits)];
    if (e) { br.drop(e & 15u); return (int)(e >> 4); }
    int code = 0, first = 0, index = 0;
    uint64_t b = br.buf;
    for (uint32_t l = 1; l < 16; ++l) {
        code |= (int)(b & 1u);
        b >>= 1;
        const int c = cnt[l];
        if (code - c < first) { br.drop(l); return sym[index + (code - first)]; }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// lane 0's decoder state between its runs (the wave refills the input ring in between)
struct Decoder {
    uint32_t phase;     // 0 block header next, 1 symbols of a Huffman block, 2 bytes of a stored block, 3 done
    uint32_t last;      // BFINAL of the current block
    uint32_t stored_left;
    uint32_t out_pos;
    uint32_t status;
    uint32_t n_tok;     // matches of this run waiting in the token list
};

// the LDS of one member's decoder
struct LaneLds {
    uint8_t *ring;
    uint16_t *lit, *dist, *lcnt, *lsym, *dcnt, *dsym;
    uint8_t *lens;
    uint2 *tok;   // kTokens matches: x = position in the text | length << 16, y = distance
    __device__ explicit LaneLds(uint8_t *base)
        : ring(base + kLdsRing), lit(reinterpret_cast<uint16_t *>(base + kLdsLit)), dist(reinterpret_cast<uint16_t *>(base + kLdsDist)),
          lcnt(reinterpret_cast<uint16_t *>(base + kLdsCnt)), lsym(lcnt + 16), dcnt(lsym + 288), dsym(dcnt + 16), lens(base + kLdsLens),
          tok(reinterpret_cast<uint2 *>(base + kLdsTok)) {}
};

// one run of a member's decoder: steps are taken while they START within 400 bytes of the run's first one (the ring holds >= 1024 bytes
// ahead, or the stream's end; the longest step — a dynamic block header — takes < 600)
__device__ void decode_run(BitReader &br, Decoder &d, const LaneLds &L, uint8_t *img, uint32_t out_len, uint32_t data_len) {
    uint16_t *const s_lit = L.lit, *const s_dist = L.dist, *const s_lcnt = L.lcnt, *const s_lsym = L.lsym, *const s_dcnt = L.dcnt, *const s_dsym = L.dsym;
    uint8_t *const s_lens = L.lens;
    // a step starts below this mark; the longest one (a dynamic block header) takes < 600 bytes, and the ring holds >= 1024 ahead (or the stream's end)
    const uint32_t run_end = br.consumed_bytes() + 400;
    while (d.phase != 3 && d.status == ST_OK && br.consumed_bytes() < run_end && d.n_tok < kTokens) {
        br.refill();
        if (br.consumed_bytes() > data_len) { d.status = ST_OVERRUN_IN; break; }
        if (d.phase == 0) {   // block header
            d.last = br.take(1);
            const uint32_t type = br.take(2);
            if (type == 0) {   // stored: skip to the byte boundary, LEN NLEN
                br.drop(br.cnt & 7u);
                br.refill();
                const uint32_t len = br.take(16), nlen = br.take(16);
                if ((len ^ 0xFFFFu) != nlen) { d.status = ST_BAD_BLOCK; break; }
                d.stored_left = len;
                d.phase = 2;
            } else if (type == 1) {   // fixed Huffman codes (RFC 1951 3.2.6)
                for (uint32_t s = 0; s < 144; ++s) s_lens[s] = 8;
                for (uint32_t s = 144; s < 256; ++s) s_lens[s] = 9;
                for (uint32_t s = 256; s < 280; ++s) s_lens[s] = 7;
                for (uint32_t s = 280; s < 288; ++s) s_lens[s] = 8;
                build_tables(s_lens, 288, s_lit, kLitBits, s_lcnt, s_lsym);
                for (uint32_t s = 0; s < 30; ++s) s_lens[s] = 5;
                build_tables(s_lens, 30, s_dist, kDistBits, s_dcnt, s_dsym);
                d.phase = 1;
            } else if (type == 2) {   // dynamic: HLIT HDIST HCLEN, the code-length code, then the two trees' lengths
                const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
                if (hlit > 286 || hdist > 30) { d.status = ST_BAD_BLOCK; break; }
                for (uint32_t i = 0; i < 19; ++i) s_lens[i] = 0;
                for (uint32_t i = 0; i < hclen; ++i) { br.refill(); s_lens[c_clen_order[i]] = (uint8_t)br.take(3); }
                // the code-length code decodes through the dist table's storage (7-bit direct table)
                if (!build_tables(s_lens, 19, s_dist, 7, s_dcnt, s_dsym, false)) { d.status = ST_BAD_BLOCK; break; }
                uint32_t i = 0;
                bool bad = false;
                while (i < hlit + hdist) {
                    br.refill();
                    const int sym = decode_sym(br, s_dist, 7, s_dcnt, s_dsym);
                    if (sym < 0) { bad = true; break; }
                    if (sym < 16) { s_lens[i++] = (uint8_t)sym; continue; }
                    uint32_t rep, val = 0;
                    if (sym == 16) { if (i == 0) { bad = true; break; } val = s_lens[i - 1]; rep = 3 + br.take(2); }
                    else if (sym == 17) rep = 3 + br.take(3);
                    else rep = 11 + br.take(7);
                    if (i + rep > hlit + hdist) { bad = true; break; }
                    while (rep--) s_lens[i++] = (uint8_t)val;
                }
                if (bad || s_lens[256] == 0) { d.status = ST_BAD_BLOCK; break; }
                // distance lengths sit behind the literal/length ones: build that tree first (its storage was the scratch above)
                uint8_t dl[30];
                for (uint32_t s = 0; s < hdist; ++s) dl[s] = s_lens[hlit + s];
                if (!build_tables(s_lens, hlit, s_lit, kLitBits, s_lcnt, s_lsym)) { d.status = ST_BAD_BLOCK; break; }
                for (uint32_t s = 0; s < hdist; ++s) s_lens[s] = dl[s];
                if (!build_tables(s_lens, hdist, s_dist, kDistBits, s_dcnt, s_dsym)) { d.status = ST_BAD_BLOCK; break; }
                d.phase = 1;
            } else { d.status = ST_BAD_BLOCK; break; }
        } else if (d.phase == 1) {   // literal / length-distance symbols
            // The common steps — a literal, or a match whose two codes are short — in one tight loop: table word, store or token, drop.
            // It stops WITHOUT having consumed anything at whatever else comes (a long code, the end of the block, an invalid symbol, the
            // output's end, the run's budget of input words or tokens), and the general step below takes that symbol or reports it.
            {
                const uint32_t w_end = (run_end + 3) / 4 + 1;
                uint32_t op = d.out_pos, nt = d.n_tok;
                for (;;) {
                    br.refill();
                    const uint32_t e = s_lit[br.peek(kLitBits)];
                    if (e == 0 || br.wpos >= w_end) break;
                    const uint32_t sym = e >> 4;
                    if (sym < 256u) {
                        if (op >= out_len) break;
                        img[op++] = (uint8_t)sym;
                        br.drop(e & 15u);
                        continue;
                    }
                    if (sym - 257u > 28u || nt >= kTokens) break;
                    BitReader b2 = br;   // (committed only when the whole match was taken here)
                    b2.drop(e & 15u);
                    const uint32_t li = sym - 257u;
                    const uint32_t lx = li < 8u || li == 28u ? 0u : (li >> 2) - 1u;
                    const uint32_t len = (li < 8u ? 3u + li : li == 28u ? 258u : ((4u + (li & 3u)) << lx) + 3u) + b2.take(lx);
                    b2.refill();
                    const uint32_t de = s_dist[b2.peek(kDistBits)];
                    const uint32_t ds = de >> 4;
                    if (de == 0 || ds > 29u) break;
                    b2.drop(de & 15u);
                    const uint32_t dx = ds < 4u ? 0u : (ds >> 1) - 1u;
                    const uint32_t dist = (ds < 4u ? 1u + ds : ((2u + (ds & 1u)) << dx) + 1u) + b2.take(dx);
                    if (dist > op || op + len > out_len) break;
                    L.tok[nt++] = make_uint2(op | (len << 16), dist);
                    op += len;
                    br = b2;
                }
                d.out_pos = op;
                d.n_tok = nt;
                if (br.consumed_bytes() > data_len) { d.status = ST_OVERRUN_IN; break; }
                if (nt >= kTokens) continue;   // (the outer loop ends the run)
            }
            const int sym = decode_sym(br, s_lit, kLitBits, s_lcnt, s_lsym);
            if (sym < 0) { d.status = ST_BAD_CODE; break; }
            if (sym < 256) {
                if (d.out_pos >= out_len) { d.status = ST_OVERRUN_OUT; break; }
                img[d.out_pos++] = (uint8_t)sym;
            } else if (sym == 256) {
                d.phase = d.last ? 3u : 0u;
            } else {
                if (sym > 285) { d.status = ST_BAD_CODE; break; }
                // base and extra bits by arithmetic (RFC 1951 3.2.5's tables are regular): a table in constant memory indexed per lane is a
                // vector load, two round trips to the cache per match inside the dependent chain
                const uint32_t li = (uint32_t)sym - 257u;
                const uint32_t lx = li < 8u || li == 28u ? 0u : (li >> 2) - 1u;
                const uint32_t lb = li < 8u ? 3u + li : li == 28u ? 258u : ((4u + (li & 3u)) << lx) + 3u;
                const uint32_t len = lb + br.take(lx);
                br.refill();
                const int ds = decode_sym(br, s_dist, kDistBits, s_dcnt, s_dsym);
                if (ds < 0 || ds > 29) { d.status = ST_BAD_CODE; break; }
                const uint32_t dx = ds < 4 ? 0u : ((uint32_t)ds >> 1) - 1u;
                const uint32_t dist = (ds < 4 ? 1u + (uint32_t)ds : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u) + br.take(dx);
                if (dist > d.out_pos) { d.status = ST_BAD_CODE; break; }
                if (d.out_pos + len > out_len) { d.status = ST_OVERRUN_OUT; break; }
                // the copy is the wave's (copy_matches): a match read here would wait for every store before it — a round trip to L2 per
                // match inside the one dependent chain the member has
                L.tok[d.n_tok++] = make_uint2(d.out_pos | (len << 16), dist);
                d.out_pos += len;
            }
        } else {   // stored bytes
            uint32_t n = d.stored_left < 256u ? d.stored_left : 256u;
            if (d.out_pos + n > out_len) { d.status = ST_OVERRUN_OUT; break; }
            for (uint32_t i = 0; i < n; ++i) { br.refill(); img[d.out_pos++] = (uint8_t)br.take(8); }
            d.stored_left -= n;
            if (d.stored_left == 0) d.phase = d.last ? 3u : 0u;
        }
    }
    if (d.status == ST_OK && d.phase == 3 && br.consumed_bytes() > data_len) d.status = ST_OVERRUN_IN;
}

// The matches a run left behind, copied by the whole wave, 64 at a time, one lane each.  A match may go when every byte it reads is final:
// literals are (the decoder stored them; the fence at the top of a round covers them), and so is everything below the first match still
// waiting, because the matches before that one went in earlier rounds.  So per round: the first waiting match, and every later one that
// reads only below it.  Block-gzip text rarely needs a third round (a line that repeats the line before it); a run of one byte — each
// match reading the one before — degrades to a round per match and is still 258 bytes a round.
__device__ void copy_matches(const uint2 *tok, uint32_t n, uint8_t *img, int lane) {
    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
        const uint32_t t = b0 + (uint32_t)lane;
        bool waiting = t < n;
        const uint2 tk = waiting ? tok[t] : make_uint2(0, 1);
        const uint32_t o = tk.x & 0xFFFFu, len = tk.x >> 16, dist = tk.y;
        const uint32_t read_end = o - dist + (len < dist ? len : dist);   // one past the last byte it reads
        for (;;) {
            const uint64_t wm = __ballot(waiting);
            if (!wm) break;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // stores so far (literals, earlier rounds) have landed
            const int first = __builtin_ctzll(wm);
            const uint32_t o_first = (uint32_t)__builtin_amdgcn_readlane((int)o, first);
            if (waiting && (lane == first || read_end <= o_first)) {
                uint8_t *dst = img + o;
                const uint8_t *f = dst - dist;
                auto put_tail = [](uint8_t *q, uint64_t v, uint32_t r) {   // the low r (< 8) bytes of v
                    if (r & 4u) { const uint32_t w = (uint32_t)v; __builtin_memcpy(q, &w, 4); q += 4; v >>= 32; }
                    if (r & 2u) { const uint16_t h = (uint16_t)v; __builtin_memcpy(q, &h, 2); q += 2; v >>= 16; }
                    if (r & 1u) *q = (uint8_t)v;
                };
                if (dist >= 8) {   // eight bytes per load and store (any alignment); a piece reads nothing the same piece writes
                    uint32_t i = 0;
                    for (; i + 8 <= len; i += 8) {
                        uint64_t v;
                        __builtin_memcpy(&v, f + i, 8);
                        __builtin_memcpy(dst + i, &v, 8);
                    }
                    if (i < len) {
                        uint64_t v;
                        __builtin_memcpy(&v, f + i, 8);   // (reads up to 7 bytes past the match's source: still below dst + len, inside the image)
                        put_tail(dst + i, v, len - i);
                    }
                } else {           // a short period (runs, dinucleotide repeats, quality plateaus): the pattern repeats in a register
                    uint64_t pat = 0;
                    for (uint32_t j = 0; j < dist; ++j) pat |= (uint64_t)f[j] << (8 * j);
                    if ((8u % dist) == 0) {   // periods 1, 2, 4: every eight bytes are the same eight
                        uint64_t rep = pat;
                        for (uint32_t w = dist; w < 8; w <<= 1) rep |= rep << (8 * w);
                        uint32_t i = 0;
                        for (; i + 8 <= len; i += 8) __builtin_memcpy(dst + i, &rep, 8);
                        put_tail(dst + i, rep, len - i);
                    } else {
                        const uint32_t top = 8 * (dist - 1);
                        for (uint32_t i = 0; i < len; ++i) {
                            const uint32_t b = (uint32_t)pat & 0xFFu;
                            dst[i] = (uint8_t)b;
                            pat = (pat >> 8) | ((uint64_t)b << top);
                        }
                    }
                }
                waiting = false;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

#ifdef CID_INFLATE_STAMPS   // tools/inflate_probe.hip: cycles of wave 0 of every block per part of the kernel
__device__ unsigned long long g_inflate_stamps[8];
#define CID_STAMP(slot, t0) do { if (lane == 0) atomicAdd(&g_inflate_stamps[slot], (unsigned long long)(__builtin_readcyclecounter() - (t0))); } while (0)
#define CID_NOW() __builtin_readcyclecounter()
#else
#define CID_STAMP(slot, t0) do { (void)(t0); } while (0)
#define CID_NOW() 0ull
#endif

// LPW members per wave, decoded by its first LPW lanes side by side (the same instruction stream wherever their steps coincide: the
// kernel is bound by instruction issue, so sharing it is worth more than the divergence costs); the wave-wide parts — ring refills,
// CRC-32 — take the members one after the other.
template <int LPW>
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t *in, const BgzfMember *members, uint32_t n_members, uint8_t *out, uint32_t *status,
                                                     CrcShift shift, const uint32_t *retry /* NULL, or [0] count [1 ..] the members to take */) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x;
    const LaneLds L(smem + (size_t)(lane < LPW ? lane : 0) * kLdsBytes);
    if (retry) n_members = retry[0];   // (what k_bgzf_inflate_wave left)

    for (uint32_t base = blockIdx.x * LPW; base < n_members; base += gridDim.x * LPW) {
        const bool mine = lane < LPW && base + (uint32_t)lane < n_members;
        const uint32_t mi = !mine ? 0u : retry ? retry[1 + base + (uint32_t)lane] : base + (uint32_t)lane;
        const BgzfMember mem = mine ? members[mi] : BgzfMember{0, 0, 0, 0};
        const uint8_t *src = in + mem.in_off;
        uint8_t *img = out + mem.out_off;                 // the text is written in place: literals are stores nobody waits for, a match
                                                          // reads what this lane wrote earlier (L2-resident), and the other waves of the SIMD hide that latency
        // gzip header (RFC 1952): 1f 8b 08 FLG(4 = FEXTRA) mtime(4) xfl os | XLEN | extra ... ; trailer CRC32 ISIZE
        uint32_t st = ST_OK, data0 = 0, data_len = 0, want_crc = 0;
        if (mine) {
            if (mem.in_len < 28 || mem.out_len > 65536u) st = ST_BAD_HEADER;
            else {
                if (src[0] != 0x1f || src[1] != 0x8b || src[2] != 8 || src[3] != 4) st = ST_BAD_HEADER;
                const uint32_t xlen = src[10] | ((uint32_t)src[11] << 8);
                data0 = 12 + xlen;
                if (data0 + 8 > mem.in_len) st = ST_BAD_HEADER;
                else {
                    data_len = mem.in_len - 8 - data0;
                    const uint8_t *t = src + mem.in_len - 8;
                    want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                    const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                    if (isize != mem.out_len) st = ST_BAD_LEN;
                }
            }
        }
        const uint8_t *data = src + data0;
        bool running = mine && st == ST_OK;
        BitReader br{reinterpret_cast<const uint32_t *>(L.ring), 1, 0, 0, 0};
        bool started = false;
        Decoder d{0, 0, 0, 0, ST_OK, 0};
        uint32_t fill = 0;   // bytes of this lane's stream in its ring
        __builtin_amdgcn_wave_barrier();
        for (;;) {   // wave-uniform loop: the rings are refilled by the whole wave, member by member; then the lanes decode a run each
            const unsigned long long t_fill = CID_NOW();
            const uint32_t used_l = br.consumed_bytes();
            const uint32_t dlo = (uint32_t)reinterpret_cast<uintptr_t>(data), dhi = (uint32_t)(reinterpret_cast<uintptr_t>(data) >> 32);
#pragma unroll
            for (int m = 0; m < LPW; ++m) {
                if (!__builtin_amdgcn_readlane((int)running, m)) continue;
                const uint32_t used = (uint32_t)__builtin_amdgcn_readlane((int)used_l, m), dl = (uint32_t)__builtin_amdgcn_readlane((int)data_len, m);
                uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)fill, m);
                const uint8_t *dm = reinterpret_cast<const uint8_t *>((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)dlo, m) |
                                                                      ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)dhi, m) << 32));
                uint8_t *ring_m = smem + (size_t)m * kLdsBytes + kLdsRing;
                while (fl < dl && fl + 1024 <= used + kRing) {   // 1 KiB per step: 64 lanes x 16 bytes
                    const uint32_t o = fl + 16u * (uint32_t)lane;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (o + 16 <= dl) {
                        const uint8_t *p = dm + o;
                        if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) {
                            const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
                            v = make_uint4(q[0], q[1], q[2], q[3]);
                        } else {
                            uint32_t w[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                w[k] = p[4 * k] | ((uint32_t)p[4 * k + 1] << 8) | ((uint32_t)p[4 * k + 2] << 16) | ((uint32_t)p[4 * k + 3] << 24);
                            v = make_uint4(w[0], w[1], w[2], w[3]);
                        }
                    } else if (o < dl) {
                        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                        for (uint32_t b = 0; b < 16; ++b) if (o + b < dl) w[b >> 2] |= (uint32_t)dm[o + b] << (8u * (b & 3u));
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    *reinterpret_cast<uint4 *>(ring_m + (o & (kRing - 1))) = v;
                    fl += 1024;
                }
                if (fl > dl) fl = dl;
                if (lane == m) fill = fl;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CID_STAMP(0, t_fill);
            const unsigned long long t_dec = CID_NOW();
            if (running) {
                if (!started) { br.start(); started = true; }
                decode_run(br, d, L, img, mem.out_len, data_len);
                if (d.phase == 3 || d.status != ST_OK) running = false;
            }
            CID_STAMP(1, t_dec);
            const unsigned long long t_copy = CID_NOW();
            // the runs' matches, member by member, by all 64 lanes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const uint32_t ilo = (uint32_t)reinterpret_cast<uintptr_t>(img), ihi = (uint32_t)(reinterpret_cast<uintptr_t>(img) >> 32);
#pragma unroll
                for (int m = 0; m < LPW; ++m) {
                    const uint32_t nt = (uint32_t)__builtin_amdgcn_readlane((int)d.n_tok, m);
                    if (!nt) continue;
                    uint8_t *im = reinterpret_cast<uint8_t *>((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)ilo, m) |
                                                              ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)ihi, m) << 32));
                    copy_matches(reinterpret_cast<const uint2 *>(smem + (size_t)m * kLdsBytes + kLdsTok), nt, im, lane);
                }
                d.n_tok = 0;
            }
            CID_STAMP(2, t_copy);
            if (!__any(running)) break;
        }
        const unsigned long long t_crc = CID_NOW();
        if (mine && st == ST_OK) st = d.status;
        if (mine && st == ST_OK && d.out_pos != mem.out_len) st = ST_BAD_LEN;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // CRC-32 (RFC 1952 8) of every member that decoded, by all 64 lanes: slices aligned to the END of the text, so that every slice
        // but the first is exactly 1024 bytes; lane l takes slice l; the table (256 words) is built in the first ring's storage
        // (four tables, "slicing by 4": a 32-bit word of text per step, its four look-ups independent of each other — the chain through
        // the CRC register is one LDS round trip per four bytes instead of one per byte)
        uint32_t *tab = reinterpret_cast<uint32_t *>(smem + kLdsRing);
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            tab[i] = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = tab[i];
            for (int t = 1; t < 4; ++t) { c = tab[c & 0xFFu] ^ (c >> 8); tab[256 * t + i] = c; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t ilo = (uint32_t)reinterpret_cast<uintptr_t>(img), ihi = (uint32_t)(reinterpret_cast<uintptr_t>(img) >> 32);
        const uint32_t ok_l = mine && st == ST_OK;
#pragma unroll
        for (int m = 0; m < LPW; ++m) {
            if (!__builtin_amdgcn_readlane((int)ok_l, m)) continue;
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)mem.out_len, m);
            const uint8_t *im = reinterpret_cast<const uint8_t *>((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)ilo, m) |
                                                                  ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)ihi, m) << 32));
            const uint32_t n_slices = (len + 1023u) / 1024u;                // <= 64
            const uint32_t first_len = len - (n_slices ? (n_slices - 1u) * 1024u : 0u);
            uint32_t c = 0;
            if ((uint32_t)lane < n_slices) {
                const uint32_t b0 = lane == 0 ? 0u : first_len + ((uint32_t)lane - 1u) * 1024u;
                const uint32_t b1 = lane == 0 ? first_len : b0 + 1024u;
                c = lane == 0 ? 0xFFFFFFFFu : 0u;
                uint32_t i = b0;
                for (; i < b1 && ((b1 - i) & 3u); ++i) c = tab[(c ^ im[i]) & 0xFFu] ^ (c >> 8);   // (only the first slice is not a multiple of 4)
                for (; i < b1; i += 4) {
                    uint32_t w;
                    __builtin_memcpy(&w, im + i, 4);
                    c ^= w;
                    c = tab[768 + (c & 0xFFu)] ^ tab[512 + ((c >> 8) & 0xFFu)] ^ tab[256 + ((c >> 16) & 0xFFu)] ^ tab[c >> 24];
                }
            }
            uint32_t reg = 0xFFFFFFFFu;   // (an empty member: CRC 0)
            for (uint32_t s = 0; s < n_slices; ++s) {
                const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)s);
                if (s == 0) reg = cs;
                else {
                    uint32_t r = 0;
                    for (uint32_t j = 0; j < 32; ++j) r ^= shift.m[j] & (0u - ((reg >> j) & 1u));
                    reg = r ^ cs;
                }
            }
            if (lane == m && (reg ^ 0xFFFFFFFFu) != want_crc) st = ST_BAD_CRC;
        }
        if (mine) status[mi] = st;
        CID_STAMP(3, t_crc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------ one member per WAVE
// k_bgzf_inflate above keeps 62 lanes of every wave waiting for one DEFLATE chain.  Huffman codes fall into step: a decoder started at
// an arbitrary bit of a block of FASTQ text sits on a true symbol boundary after 8 symbols at the median, 51 at the 99th percentile
// (tools/exp_inflate_sync.py, profiles/r05_inflate_sync_probe.json).  So the 64 lanes cut a block's bits into 64 chunks and decode them
// at once, from guessed starts:
//   a pass      lane i decodes from its start u_i up to the end of its chunk and notes where it lands in the next chunk (e_i), how many
//               bytes and matches that makes, and whether it met the end of the block;
//   the check   lane 0's start is true; lane i + 1's was true if it EQUALS e_i of a lane that was true.  The first lane that fails gets
//               e_i as its start (every other lane its neighbour's latest landing point) and the pass is repeated — two passes when
//               every guess fell into step inside its own chunk, which is the rule;
//   the writing prefix sums of the bytes and matches give every chunk its place: the lanes decode once more, literals go to the text,
//               matches to a token list in HBM (sorted by position), and copy_matches resolves them wave-wide as before.
// Anything out of the ordinary — a stored block, a member whose compressed bits do not fit 2^19, no agreement after kWaveMaxPasses, any
// invalid code on the true path — hands the member to the one-lane kernel (status kRetry, a retry list), which also owns the error codes.
constexpr uint32_t kRetry = 100;           // status of a member left for the one-lane kernel
constexpr uint32_t kWaveTokens = 21848;    // matches of one member at most: three bytes each of 65 536
constexpr uint32_t kWaveMaxPasses = 12;

struct GBits {   // a lane's view of the compressed stream in HBM: 32-bit words, two of them read ahead of the bit buffer
    const uint32_t *w;
    uint32_t n_words;   // words that exist (beyond them: zeros)
    uint32_t wpos, cnt, a0, a1, pos;   // pos: the next bit, counted from w
    uint64_t buf;
    __device__ __forceinline__ uint32_t ld(uint32_t i) const { return i < n_words ? w[i] : 0u; }
    __device__ __forceinline__ void start(uint32_t bit) {
        wpos = bit >> 5;
        const uint32_t sh = bit & 31u;
        buf = (uint64_t)(ld(wpos) >> sh);
        cnt = 32u - sh;
        a0 = ld(wpos + 1); a1 = ld(wpos + 2);
        wpos += 3;
        pos = bit;
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32) {
            buf |= (uint64_t)a0 << cnt;
            cnt += 32;
            a0 = a1;
            a1 = ld(wpos++);
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; pos += n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

struct ChunkEnd { uint32_t end_pos, n_out, n_tok, flags; };   // flags: 1 = the end of the block was met (end_pos: the bit after it), 2 = not a valid code

// The wave kernel's tables: 12 / 10 bits wide instead of 10 / 8.  64 lanes decode different symbols in lockstep, so a code longer than the
// direct table sends the WHOLE wave through the bit-by-bit loop whenever any lane meets one — with 10 / 8 bits that was most iterations
// (2 200 cycles per symbol step); and the tables are filled by the wave, an index per lane, instead of by one lane's nest of loops
// (367 k cycles per member for the headers alone).
constexpr int kWLitBits = 11, kWDistBits = 9;
constexpr uint32_t kWLit = 0, kWDist = kWLit + 2u * (1u << kWLitBits), kWCnt = kWDist + 2u * (1u << kWDistBits), kWLens = kWCnt + 2u * (16 + 288 + 16 + 32),
                   kWClen = kWLens + 320, kWRun = kWClen + 2u * (128 + 16 + 20), kWaveLdsBytes = kWRun + 4u * 32;
struct WaveLds {
    uint16_t *lit, *dist, *lcnt, *lsym, *dcnt, *dsym, *ctab, *ccnt, *csym;
    uint8_t *lens;
    uint32_t *run;   // 32 words of scratch for the table builder
    __device__ explicit WaveLds(uint8_t *base)
        : lit(reinterpret_cast<uint16_t *>(base + kWLit)), dist(reinterpret_cast<uint16_t *>(base + kWDist)), lcnt(reinterpret_cast<uint16_t *>(base + kWCnt)),
          lsym(lcnt + 16), dcnt(lsym + 288), dsym(dcnt + 16), ctab(reinterpret_cast<uint16_t *>(base + kWClen)), ccnt(ctab + 128), csym(ccnt + 16),
          lens(base + kWLens), run(reinterpret_cast<uint32_t *>(base + kWRun)) {}
};

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// build_tables by all 64 lanes (same tables, same verdict): the lengths are counted with LDS atomics, the symbols are ranked inside their length
// by ballots, and every index of the direct table finds its own code by walking the canonical code (what decode_sym's slow path does per symbol)
__device__ __noinline__ bool wave_build_tables(const uint8_t *lens, uint32_t n_sym, uint16_t *tab, uint32_t bits, uint16_t *cnt, uint16_t *sym, uint32_t *run, int lane,
                                  bool allow_single = true) {
    if (lane < 32) run[lane] = 0;
    wave_sync_lds();
    for (uint32_t s = lane; s < n_sym; s += 64) atomicAdd(&run[lens[s]], 1u);
    wave_sync_lds();
    uint32_t c[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) c[l] = run[l];
    wave_sync_lds();
    if (lane < 16) cnt[lane] = (uint16_t)c[lane];
    if (c[0] == n_sym) {   // no codes at all (a distance tree may be empty)
        for (uint32_t i = lane; i < (1u << bits); i += 64) tab[i] = 0;
        wave_sync_lds();
        return true;
    }
    int left = 1;
    bool over = false;
#pragma unroll
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= (int)c[l]; over = over || left < 0; }
    if (over) return false;
    uint32_t offs[16];
    offs[0] = 0; offs[1] = 0;
#pragma unroll
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + c[l];
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n_sym; base += 64) {   // symbols in order: a symbol's place is its rank among the symbols of its length
        const uint32_t s = base + (uint32_t)lane;
        const uint32_t l = s < n_sym ? lens[s] : 0u;
#pragma unroll
        for (int len = 1; len < 16; ++len) {
            const uint64_t m = __ballot(l == (uint32_t)len);
            if (l == (uint32_t)len) sym[offs[len] + (uint32_t)__popcll(m & lt)] = (uint16_t)s;
            offs[len] += (uint32_t)__popcll(m);
        }
    }
    wave_sync_lds();
    for (uint32_t i = lane; i < (1u << bits); i += 64) {
        int code = 0, first = 0, index = 0;
        uint32_t e = 0;
#pragma unroll 1
        for (uint32_t l = 1; l <= bits; ++l) {
            code |= (int)((i >> (l - 1)) & 1u);
            const int n = (int)cnt[l];   // (bits <= 12: LDS, written above — a register array indexed by l would live in scratch memory)
            if (code - n < first) { e = ((uint32_t)sym[index + (code - first)] << 4) | l; break; }
            index += n;
            first += n;
            first <<= 1;
            code <<= 1;
        }
        tab[i] = (uint16_t)e;
    }
    wave_sync_lds();
    return left == 0 || (allow_single && (n_sym - c[0]) == 1 && c[1] == 1);
}

// one lane's chunk: symbols from br.pos while they START below `limit`.  WRITE: literals to img[out_off ..], matches to tok[tok_off ..]
template <bool WRITE>
__device__ ChunkEnd decode_chunk(GBits &br, uint32_t limit, const WaveLds &L, uint8_t *img, uint32_t out_off, uint32_t out_len, uint2 *tok, uint32_t tok_off) {
    ChunkEnd r{0, 0, 0, 0};
    // WRITE: a run of literals leaves eight bytes per store (any alignment), what is left of it when a match or the chunk's end comes in 4 / 2 / 1
    // (one byte per store made the writing pass the most expensive one under load: 64 lanes x 1 byte to 64 lines per instruction)
    uint64_t acc = 0;
    uint32_t n_acc = 0, at_acc = out_off;
    auto flush = [&]() {
        uint8_t *q = img + at_acc;
        uint64_t v = acc;
        if (n_acc & 4u) { const uint32_t w = (uint32_t)v; __builtin_memcpy(q, &w, 4); q += 4; v >>= 32; }
        if (n_acc & 2u) { const uint16_t h = (uint16_t)v; __builtin_memcpy(q, &h, 2); q += 2; v >>= 16; }
        if (n_acc & 1u) *q = (uint8_t)v;
        acc = 0; n_acc = 0;
    };
    while (br.pos < limit) {
        br.refill();
        const int sym = decode_sym(br, L.lit, kWLitBits, L.lcnt, L.lsym);
        if (sym < 0) { r.flags = 2; break; }
        if (sym < 256) {
            if (WRITE) {
                if (out_off + r.n_out >= out_len) { r.flags = 2; break; }
                acc |= (uint64_t)(uint32_t)sym << (8u * n_acc);
                if (++n_acc == 8u) { __builtin_memcpy(img + at_acc, &acc, 8); at_acc += 8; acc = 0; n_acc = 0; }
            }
            ++r.n_out;
        } else if (sym == 256) {
            r.flags = 1;
            break;
        } else {
            if (sym > 285) { r.flags = 2; break; }
            const uint32_t li = (uint32_t)sym - 257u;
            const uint32_t lx = li < 8u || li == 28u ? 0u : (li >> 2) - 1u;
            const uint32_t len = (li < 8u ? 3u + li : li == 28u ? 258u : ((4u + (li & 3u)) << lx) + 3u) + br.take(lx);
            br.refill();
            const int ds = decode_sym(br, L.dist, kWDistBits, L.dcnt, L.dsym);
            if (ds < 0 || ds > 29) { r.flags = 2; break; }
            const uint32_t dx = ds < 4 ? 0u : ((uint32_t)ds >> 1) - 1u;
            const uint32_t dist = (ds < 4 ? 1u + (uint32_t)ds : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u) + br.take(dx);
            if (WRITE) {
                const uint32_t at = out_off + r.n_out;
                if (dist > at || at + len > out_len) { r.flags = 2; break; }
                tok[tok_off + r.n_tok] = make_uint2(at | (len << 16), dist);
                if (n_acc) flush();
                at_acc = at + len;   // the next literal lands behind the match
            }
            r.n_out += len;
            ++r.n_tok;
        }
        if (r.n_out > (1u << 20)) { r.flags = 2; break; }   // (a guessed start decoding nonsense: nothing of it is used)
    }
    if (WRITE && n_acc) flush();
    r.end_pos = br.pos;
    return r;
}

// lane 0: the header of the block at `bit` up to its code lengths (L.lens: the literal/length tree's hlit lengths, then the distance tree's hdist);
// returns the first bit of the symbols, or 0 = not this kernel's case (a stored block, an invalid header)
__device__ __noinline__ uint32_t wave_block_header(const uint32_t *w, uint32_t n_words, uint32_t bit, const WaveLds &L, uint32_t *last, uint32_t *hlit_out, uint32_t *hdist_out) {
    GBits br{w, n_words, 0, 0, 0, 0, 0, 0};
    br.start(bit);
    br.refill();
    *last = br.take(1);
    const uint32_t type = br.take(2);
    uint8_t *const s_lens = L.lens;
    if (type == 1) {   // fixed codes (RFC 1951 3.2.6)
        for (uint32_t s = 0; s < 144; ++s) s_lens[s] = 8;
        for (uint32_t s = 144; s < 256; ++s) s_lens[s] = 9;
        for (uint32_t s = 256; s < 280; ++s) s_lens[s] = 7;
        for (uint32_t s = 280; s < 288; ++s) s_lens[s] = 8;
        for (uint32_t s = 0; s < 30; ++s) s_lens[288 + s] = 5;
        *hlit_out = 288; *hdist_out = 30;
        return br.pos;
    }
    if (type != 2) return 0;
    const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return 0;
    uint8_t cl[19];
    for (uint32_t i = 0; i < 19; ++i) cl[i] = 0;
    for (uint32_t i = 0; i < hclen; ++i) { br.refill(); cl[c_clen_order[i]] = (uint8_t)br.take(3); }
    if (!build_tables(cl, 19, L.ctab, 7, L.ccnt, L.csym, false)) return 0;
    uint32_t i = 0;
    while (i < hlit + hdist) {
        br.refill();
        const int sym = decode_sym(br, L.ctab, 7, L.ccnt, L.csym);
        if (sym < 0) return 0;
        if (sym < 16) { s_lens[i++] = (uint8_t)sym; continue; }
        uint32_t rep, val = 0;
        if (sym == 16) { if (i == 0) return 0; val = s_lens[i - 1]; rep = 3 + br.take(2); }
        else if (sym == 17) rep = 3 + br.take(3);
        else rep = 11 + br.take(7);
        if (i + rep > hlit + hdist) return 0;
        while (rep--) s_lens[i++] = (uint8_t)val;
    }
    if (s_lens[256] == 0) return 0;
    *hlit_out = hlit; *hdist_out = hdist;
    return br.pos;
}


__global__ __launch_bounds__(64) void k_bgzf_inflate_wave(const uint8_t *in, const BgzfMember *members, uint32_t n_members, uint8_t *out, uint32_t *status,
                                                          CrcShift shift, uint2 *tok_all, uint32_t *retry /* [0] count, [1 ..] members */) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x;
    const WaveLds L(smem);
    for (uint32_t mi = blockIdx.x; mi < n_members; mi += gridDim.x) {
        wave_sync_lds();
        const BgzfMember mem = members[mi];
        const uint8_t *src = in + mem.in_off;
        uint8_t *img = out + mem.out_off;
        uint2 *tok = tok_all + (size_t)mi * kWaveTokens;
        // gzip header (RFC 1952), as in k_bgzf_inflate; anything unexpected is the one-lane kernel's to report
        bool mine_ok = mem.in_len >= 28 && mem.out_len <= 65536u && src[0] == 0x1f && src[1] == 0x8b && src[2] == 8 && src[3] == 4;
        uint32_t data0 = 0, data_len = 0, want_crc = 0;
        if (mine_ok) {
            const uint32_t xlen = src[10] | ((uint32_t)src[11] << 8);
            data0 = 12 + xlen;
            if (data0 + 8 > mem.in_len) mine_ok = false;
            else {
                data_len = mem.in_len - 8 - data0;
                const uint8_t *t = src + mem.in_len - 8;
                want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (isize != mem.out_len) mine_ok = false;
            }
        }
        if (data_len >= (1u << 16)) mine_ok = false;
        const uintptr_t dptr = reinterpret_cast<uintptr_t>(src + data0);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(dptr & ~(uintptr_t)3);
        const uint32_t bit0 = 8u * (uint32_t)(dptr & 3u), end_bit = bit0 + 8u * data_len;
        const uint32_t n_words = (end_bit + 31u) / 32u + 2u;   // (the trailer's 8 bytes follow the data: still inside the member)
        uint32_t bit = bit0, out_pos = 0, n_tok = 0;
        bool ok = mine_ok, final = false;
        while (ok && !final) {
            // ---- the block's header and tables: lane 0
            const unsigned long long t_hdr = CID_NOW();
            uint32_t sym0 = 0, last = 0, hlit = 0, hdist = 0;
            if (lane == 0) sym0 = wave_block_header(w, n_words, bit, L, &last, &hlit, &hdist);
            sym0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sym0);
            last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
            hlit = (uint32_t)__builtin_amdgcn_readfirstlane((int)hlit);
            hdist = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdist);
            wave_sync_lds();
            if (sym0 == 0 || sym0 >= end_bit) { ok = false; break; }
            if (!wave_build_tables(L.lens, hlit, L.lit, kWLitBits, L.lcnt, L.lsym, L.run, lane) ||
                !wave_build_tables(L.lens + hlit, hdist, L.dist, kWDistBits, L.dcnt, L.dsym, L.run, lane)) { ok = false; break; }
            CID_STAMP(4, t_hdr);
            const unsigned long long t_pass = CID_NOW();
            // ---- the passes
            const uint32_t chunk = (end_bit - sym0 + 63u) / 64u < 256u ? 256u : (end_bit - sym0 + 63u) / 64u;
            const uint32_t limit = lane == 63 ? end_bit : (sym0 + ((uint32_t)lane + 1u) * chunk < end_bit ? sym0 + ((uint32_t)lane + 1u) * chunk : end_bit);
            uint32_t u = sym0 + (uint32_t)lane * chunk;
            if (u > end_bit) u = end_bit;
            ChunkEnd r{0, 0, 0, 0};
            int E = -1;   // the lane that met the end of the block on the true path
            for (uint32_t pass = 0; pass < kWaveMaxPasses; ++pass) {
                GBits br{w, n_words, 0, 0, 0, 0, 0, 0};
                br.start(u);
                r = u < limit ? decode_chunk<false>(br, limit, L, nullptr, 0, 0, nullptr, 0) : ChunkEnd{u, 0, 0, 0};
                const uint32_t e_prev = __shfl_up(r.end_pos, 1, 64);
                const uint32_t f_prev = __shfl_up(r.flags, 1, 64);
                const bool link = lane == 0 || (u == e_prev && f_prev == 0);
                const uint64_t lm = __ballot(link), fm = __ballot(r.flags != 0);
                const int V = ~lm ? __builtin_ctzll(~lm) : 64;          // lanes 0 .. V-1 started on the true path
                const int F = fm ? __builtin_ctzll(fm) : 64;            // the first lane that stopped
                if (F < V) { E = F; break; }
                if (V == 64) break;                                     // the stream ends without an end of block
                if (lane >= V) u = e_prev < end_bit ? e_prev : end_bit;
            }
            if (E < 0) { ok = false; break; }
            CID_STAMP(5, t_pass);
            const unsigned long long t_wr = CID_NOW();
            const uint32_t flagsE = (uint32_t)__builtin_amdgcn_readlane((int)r.flags, E);
            if (flagsE != 1u) { ok = false; break; }
            // ---- where every chunk's bytes and matches go
            const bool part = lane <= E;
            uint32_t o_inc = part ? r.n_out : 0u, t_inc = part ? r.n_tok : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t a = __shfl_up(o_inc, o, 64), b = __shfl_up(t_inc, o, 64);
                if (lane >= o) { o_inc += a; t_inc += b; }
            }
            const uint32_t tot_out = (uint32_t)__builtin_amdgcn_readlane((int)o_inc, 63), tot_tok = (uint32_t)__builtin_amdgcn_readlane((int)t_inc, 63);
            if (out_pos + tot_out > mem.out_len || n_tok + tot_tok > kWaveTokens) { ok = false; break; }
            // ---- the writing pass
            uint32_t bad = 0;
            if (part && u < limit) {
                GBits br{w, n_words, 0, 0, 0, 0, 0, 0};
                br.start(u);
                const ChunkEnd wr = decode_chunk<true>(br, limit, L, img, out_pos + o_inc - r.n_out, mem.out_len, tok, n_tok + t_inc - r.n_tok);
                bad = (wr.flags & 2u) | (wr.end_pos != r.end_pos ? 2u : 0u);
            }
            if (__any(bad != 0)) { ok = false; break; }
            CID_STAMP(6, t_wr);
            out_pos += tot_out;
            n_tok += tot_tok;
            bit = (uint32_t)__builtin_amdgcn_readlane((int)r.end_pos, E);
            final = last != 0;
            wave_sync_lds();
        }
        if (ok && (out_pos != mem.out_len || (bit + 7u) / 8u > bit0 / 8u + data_len)) ok = false;
        if (!ok) {   // the one-lane kernel takes it (and names what is wrong with it, if anything is)
            if (lane == 0) { status[mi] = kRetry; retry[1 + atomicAdd(&retry[0], 1u)] = mi; }
            continue;
        }
        // ---- the matches, by the whole wave (the literals and the tokens are in HBM: visible to the wave behind a fence)
        const unsigned long long t_cp = CID_NOW();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        copy_matches(tok, n_tok, img, lane);
        CID_STAMP(7, t_cp);
        // ---- CRC-32, as in k_bgzf_inflate (slices of 1 KiB aligned to the text's end, slicing by 4, folded with the shift operator)
        uint32_t *tab = reinterpret_cast<uint32_t *>(smem + kWLit);   // (4 KiB of the literal/length table's 8: the decoding is over)
        wave_sync_lds();
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            tab[i] = c;
        }
        wave_sync_lds();
        for (uint32_t i = lane; i < 256; i += 64) {
            uint32_t c = tab[i];
            for (int t = 1; t < 4; ++t) { c = tab[c & 0xFFu] ^ (c >> 8); tab[256 * t + i] = c; }
        }
        wave_sync_lds();
        const uint32_t len = mem.out_len;
        const uint32_t n_slices = (len + 1023u) / 1024u;
        const uint32_t first_len = len - (n_slices ? (n_slices - 1u) * 1024u : 0u);
        uint32_t c = 0;
        if ((uint32_t)lane < n_slices) {
            const uint32_t b0 = lane == 0 ? 0u : first_len + ((uint32_t)lane - 1u) * 1024u;
            const uint32_t b1 = lane == 0 ? first_len : b0 + 1024u;
            c = lane == 0 ? 0xFFFFFFFFu : 0u;
            uint32_t i = b0;
            for (; i < b1 && ((b1 - i) & 3u); ++i) c = tab[(c ^ img[i]) & 0xFFu] ^ (c >> 8);
            for (; i < b1; i += 4) {
                uint32_t wd;
                __builtin_memcpy(&wd, img + i, 4);
                c ^= wd;
                c = tab[768 + (c & 0xFFu)] ^ tab[512 + ((c >> 8) & 0xFFu)] ^ tab[256 + ((c >> 16) & 0xFFu)] ^ tab[c >> 24];
            }
        }
        uint32_t reg = 0xFFFFFFFFu;
        for (uint32_t sl = 0; sl < n_slices; ++sl) {
            const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)sl);
            if (sl == 0) reg = cs;
            else {
                uint32_t rr = 0;
                for (uint32_t j = 0; j < 32; ++j) rr ^= shift.m[j] & (0u - ((reg >> j) & 1u));
                reg = rr ^ cs;
            }
        }
        if (lane == 0) {
            if ((reg ^ 0xFFFFFFFFu) != want_crc) { status[mi] = kRetry; retry[1 + atomicAdd(&retry[0], 1u)] = mi; }   // (the one-lane kernel confirms or names it)
            else status[mi] = ST_OK;
        }
    }
}

static CrcShift make_crc_shift() {
    uint32_t tab[256];
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        tab[i] = c;
    }
    CrcShift s;
    for (uint32_t j = 0; j < 32; ++j) {
        uint32_t c = 1u << j;
        for (int i = 0; i < 1024; ++i) c = tab[c & 0xFFu] ^ (c >> 8);
        s.m[j] = c;
    }
    return s;
}

constexpr int kInflateLanes = 2;
hipError_t warm_inflate() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_bgzf_inflate<kInflateLanes>));
}

// the kernel on device-resident members: text to d_out + member.out_off, one status word per member; asynchronous on the ctx stream
size_t bgzf_inflate_scratch_bytes(uint32_t n_members) { return (size_t)n_members * kWaveTokens * sizeof(uint2) + ((size_t)n_members + 4) * 4; }

hipError_t bgzf_inflate_launch(cid_ctx *c, hipStream_t stream, const uint8_t *d_in, const BgzfMember *d_mem, uint32_t n_members, uint8_t *d_out,
                               uint32_t *d_st, void *d_scratch) {
    if (n_members == 0) return hipSuccess;
    static const CrcShift shift = make_crc_shift();
    // One member per WAVE first (k_bgzf_inflate_wave: 64 lanes on the chunks of a block); what it leaves — stored blocks, corrupt members, the
    // rare block whose chunks do not fall into step — goes to the one-lane kernel through the retry list.  CID_INFLATE_WAVE=0, or a caller
    // without scratch: the one-lane kernel for everything.
    const bool wave_env = c->tune.inflate_wave;
    uint32_t *d_retry = nullptr;
    if (wave_env && d_scratch) {
        uint2 *d_tok = reinterpret_cast<uint2 *>(d_scratch);
        d_retry = reinterpret_cast<uint32_t *>(d_tok + (size_t)n_members * kWaveTokens);
        hipError_t e = hipMemsetAsync(d_retry, 0, 4, stream);
        if (e != hipSuccess) return e;
        unsigned grid = n_members;
        const unsigned cap = (unsigned)c->n_cu * 20u;   // (7.4 KiB of LDS per wave: twenty per CU)
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(k_bgzf_inflate_wave, dim3(grid), dim3(64), kWaveLdsBytes, stream, d_in, d_mem, n_members, d_out, d_st, shift, d_tok, d_retry);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        unsigned rgrid = n_members < 256u ? n_members : 256u;
        hipLaunchKernelGGL(k_bgzf_inflate<1>, dim3(rgrid), dim3(64), kLdsBytes, stream, d_in, d_mem, n_members, d_out, d_st, shift, (const uint32_t *)d_retry);
        return hipGetLastError();
    }
    // members per wave (CID_INFLATE_LANES: 1, 2, 4 or 8).  Two decoders share a wave's instruction stream where their steps coincide, which
    // doubles what a full chip decodes per unit time, but each runs at 0.6 of the speed it has alone: a launch that leaves the chip mostly
    // idle anyway (<= 1 280 members = five waves per CU) takes one per wave (256 members: 4.5 against 7.9 ms, 1 024: 5.5 against 8.4,
    // the 4 794 of a million reads: 14.3 against 9.3 — tools/exp_inflate_lanes.sh)
    const int lanes_env = c->tune.inflate_lanes;
    const int lanes = lanes_env ? lanes_env : n_members <= 1280u ? 1 : kInflateLanes;
    const unsigned lpw = lanes == 1 ? 1u : lanes == 4 ? 4u : lanes == 8 ? 8u : 2u;
    unsigned grid = (unsigned)((n_members + lpw - 1) / lpw);
    const unsigned cap = (unsigned)c->n_cu * 32u * 4u;   // a few rounds per block at most
    if (grid > cap) grid = cap;
    const size_t lds = (size_t)lpw * kLdsBytes;
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, stream, d_in, d_mem, n_members, d_out, d_st, shift, (const uint32_t *)nullptr); };
    if (lpw == 1) launch(k_bgzf_inflate<1>);
    else if (lpw == 4) launch(k_bgzf_inflate<4>);
    else if (lpw == 8) launch(k_bgzf_inflate<8>);
    else launch(k_bgzf_inflate<2>);
    return hipGetLastError();
}
const char *bgzf_status_text(uint32_t st) {
    static const char *const why[] = {"", "not a BGZF member header", "invalid DEFLATE block", "invalid Huffman code", "compressed data ends early",
                                      "more text than the member's ISIZE", "text length differs from ISIZE", "CRC-32 mismatch"};
    return why[st < 8 ? st : 0];
}

}  // namespace cid

using cid::fail;
using namespace cid::slots;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return cid::fail(CID_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// start: the members go to the device (through the ctx's pinned arena when they fit, so that the caller's buffer is free again when
// this returns), the kernel and the copies of the text and of the members' status back into the arena are queued on the ctx stream;
// finish: waits, checks every member, copies the text out.  Two contexts taking turns keep two batches in flight.
extern "C" int cid_bgzf_inflate_start(cid_ctx *c, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                                      const uint32_t *text_off, const uint32_t *text_len, size_t n_members, size_t text_bytes) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    if (c->inflate.open) return fail(CID_ERR_STATE, "cid_bgzf_inflate_start: the previous batch has not been finished");
    c->inflate.n_members = n_members; c->inflate.text_bytes = text_bytes; c->inflate.staged = false;
    if (n_members == 0) { c->inflate.open = true; return CID_OK; }
    if (!members || !member_off || !member_len || !text_off || !text_len) return fail(CID_ERR_INVALID, "null argument");
    if (n_bytes >= (1ull << 32) || text_bytes >= (1ull << 32) || n_members >= (1ull << 31)) return fail(CID_ERR_UNSUPPORTED,
        "a batch of BGZF members is limited to 4 GiB");
    std::vector<cid::BgzfMember> mem(n_members);
    for (size_t i = 0; i < n_members; ++i) {
        if ((uint64_t)member_off[i] + member_len[i] > n_bytes) return fail(CID_ERR_INVALID, "member %zu lies outside the batch", i);
        if (text_len[i] > 65536u || (uint64_t)text_off[i] + text_len[i] > text_bytes) return fail(CID_ERR_INVALID,
            "member %zu: text range outside the output", i);
        mem[i] = cid::BgzfMember{member_off[i], member_len[i], text_off[i], text_len[i]};
    }
    HIP_TRY(hipSetDevice(c->device));
    void *d_in, *d_mem, *d_out, *d_st;
    int rc = cid::slot_reserve(c, S_KMERS, n_bytes + 16, &d_in); if (rc) return rc;
    rc = cid::slot_reserve(c, S_MISC, n_members * sizeof(cid::BgzfMember), &d_mem); if (rc) return rc;
    rc = cid::slot_reserve(c, S_BASES, text_bytes + 16, &d_out); if (rc) return rc;
    rc = cid::slot_reserve(c, S_FREQ, n_members * 4, &d_st); if (rc) return rc;
    void *d_scratch = nullptr;   // (refused: the one-lane kernel takes the batch)
    if (cid::slot_reserve(c, S_ROWIDS, cid::bgzf_inflate_scratch_bytes((uint32_t)n_members), &d_scratch) != CID_OK) d_scratch = nullptr;
    // arena: members | member table | text | status
    const size_t b_mem = (n_bytes + 63) & ~(size_t)63, b_text = b_mem + ((n_members * sizeof(cid::BgzfMember) + 63) & ~(size_t)63),
                 b_st = b_text + ((text_bytes + 63) & ~(size_t)63), b_end = b_st + n_members * 4;
    uint8_t *pin = cid::pin_reserve(c, b_end + 64, 512u << 20);
    if (pin) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(pin, members, n_bytes);
        memcpy(pin + b_mem, mem.data(), n_members * sizeof(cid::BgzfMember));
        HIP_TRY(hipMemcpyAsync(d_in, pin, n_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_mem, pin + b_mem, n_members * sizeof(cid::BgzfMember), hipMemcpyHostToDevice, c->stream));
    } else {   // (no arena: the caller's buffers are read before this returns)
        HIP_TRY(hipMemcpyAsync(d_in, members, n_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_mem, mem.data(), n_members * sizeof(cid::BgzfMember), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(cid::bgzf_inflate_launch(c, c->stream, (const uint8_t *)d_in, (const cid::BgzfMember *)d_mem, (uint32_t)n_members, (uint8_t *)d_out, (uint32_t *)d_st, d_scratch));
    c->inflate.d_out = d_out; c->inflate.d_st = d_st;
    if (pin) {
        HIP_TRY(hipMemcpyAsync(pin + b_st, d_st, n_members * 4, hipMemcpyDeviceToHost, c->stream));
        if (text_bytes) HIP_TRY(hipMemcpyAsync(pin + b_text, d_out, text_bytes, hipMemcpyDeviceToHost, c->stream));
        c->inflate.staged = true; c->inflate.pin_text = b_text; c->inflate.pin_status = b_st;
    }
    c->inflate.open = true;
    return CID_OK;
}

extern "C" int cid_bgzf_inflate_finish(cid_ctx *c, uint8_t *text, size_t text_bytes, size_t *bad_member) {
    if (!c) return fail(CID_ERR_INVALID, "null ctx");
    if (bad_member) *bad_member = (size_t)-1;
    if (!c->inflate.open) return fail(CID_ERR_STATE, "cid_bgzf_inflate_finish without a start");
    c->inflate.open = false;
    const size_t n_members = c->inflate.n_members;
    if (n_members == 0) return CID_OK;
    if (text_bytes != c->inflate.text_bytes || (text_bytes && !text)) return fail(CID_ERR_INVALID, "text buffer differs from the one announced at the start");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<uint32_t> st(n_members);
    if (c->inflate.staged) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(st.data(), c->pin + c->inflate.pin_status, n_members * 4);
    } else {
        HIP_TRY(hipMemcpyAsync(st.data(), c->inflate.d_st, n_members * 4, hipMemcpyDeviceToHost, c->stream));
        if (text_bytes) HIP_TRY(hipMemcpyAsync(text, c->inflate.d_out, text_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (size_t i = 0; i < n_members; ++i)
        if (st[i] != cid::ST_OK) {
            if (bad_member) *bad_member = i;
            return fail(CID_ERR_INVALID, "corrupt gzip member %zu: %s", i, cid::bgzf_status_text(st[i]));
        }
    if (c->inflate.staged && text_bytes) memcpy(text, c->pin + c->inflate.pin_text, text_bytes);
    return CID_OK;
}

extern "C" int cid_bgzf_inflate(cid_ctx *c, const uint8_t *members, size_t n_bytes, const uint32_t *member_off, const uint32_t *member_len,
                                const uint32_t *text_off, const uint32_t *text_len, size_t n_members, uint8_t *text, size_t text_bytes,
                                size_t *bad_member) {
    if (bad_member) *bad_member = (size_t)-1;
    if (n_members && text_bytes && !text) return fail(CID_ERR_INVALID, "null argument");
    const int rc = cid_bgzf_inflate_start(c, members, n_bytes, member_off, member_len, text_off, text_len, n_members, text_bytes);
    if (rc) { if (c) c->inflate.open = false; return rc; }
    return cid_bgzf_inflate_finish(c, text, text_bytes, bad_member);
}
