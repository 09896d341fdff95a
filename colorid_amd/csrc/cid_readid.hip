// read_id kernels for MI355X (gfx950): one wave per read(-pair) — a6/a7/a9/a10 of SURVEY.md §8
// (read_id_mt_pe.rs:66-165, 282-363; kmer.rs:221-243, 363-394).  Index layout and gather scheme: see cid_search.hip.
#include "cid_gather.hpp"

namespace cid {

// Colour stripes: zacc / zin point at this read's slice of ReadIdParams::zero_acc / zero_in (NULL = not that pass)
struct StripeRead { uint32_t *zacc; const uint32_t *zin; };

// read_id over wide rows: the chunk's distinct k-mers one at a time, in order; counts go straight to the (pre-zeroed)
// report row.  s_words / s_R: rs u64 words each per wave (the AND word of the current k-mer, the colours of the first S).
// STRIPED: the row is one colour stripe's — zero pass: record which seeds' rows are all-zero here, count nothing; count pass:
// "absent" comes from the masks accumulated over all stripes, colours land at colour_base, the no-hits column (nohits_col)
// is written by one stripe only.
template <bool STRIPED = false>
__device__ __forceinline__ void readid_search_chunk_wide(const uint64_t *mat, uint32_t rs, uint32_t w64, uint32_t n, uint32_t S,
                                                         const uint32_t *ridx, uint64_t *s_words, uint64_t *s_R, uint32_t *row_out,
                                                         uint64_t dmask, uint32_t nd, bool &stopped, int lane, uint32_t nohits_col,
                                                         StripeRead sr = StripeRead{nullptr, nullptr}, uint32_t colour_base = 0,
                                                         uint32_t write_nohits = 1) {
    if (stopped || !dmask) return;
    const uint32_t steps = rs / 128u;
    const uint32_t seeds = n >= 32 ? ~0u : ((1u << n) - 1u);
    uint32_t q = nd;
    for (uint64_t dm = dmask; dm; dm &= dm - 1, ++q) {
        const int kk = __builtin_ctzll(dm);
        uint32_t zml = ~0u;
        for (uint32_t j = 0; j < steps; ++j) {
            const uint32_t col_word = 128u * j + 2u * lane;
            if (col_word >= w64) continue;
            uint32_t zm;
            const V16 a = gather_and<false, true>(mat, rs, ridx, kk, col_word, n, zm);
            s_words[col_word] = a.x;
            s_words[col_word + 1] = a.y;
            zml &= zm;
        }
        uint32_t absent = wave_and_u32(zml) & seeds;
        if constexpr (STRIPED) {
            if (sr.zacc) {   // zero pass
                if (lane == 0) sr.zacc[q] &= absent;
                continue;
            }
            if (sr.zin) absent = sr.zin[q];   // all-zero in every stripe
        }
        if (absent) {  // absent row: *report.entry(no_hits_num) += 1; break
            if (lane == 0 && write_nohits) atomicAdd(&row_out[nohits_col], 1u);
            stopped = true;
            return;
        }
        for (uint32_t j = 0; j < steps; ++j) {
            const uint32_t col_word = 128u * j + 2u * lane;
            if (col_word >= w64) continue;
            uint64_t x = s_words[col_word], y = s_words[col_word + 1];
            if (S > 0) {
                if (q < S) { s_R[col_word] |= x; s_R[col_word + 1] |= y; }
                else { x &= s_R[col_word]; y &= s_R[col_word + 1]; }
            }
            while (x) { atomicAdd(&row_out[colour_base + col_word * 64u + (uint32_t)__builtin_ctzll(x)], 1u); x &= x - 1; }
            while (y) { atomicAdd(&row_out[colour_base + col_word * 64u + 64u + (uint32_t)__builtin_ctzll(y)], 1u); y &= y - 1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a6/a7/a9/a10: per-read classification counts (src/read_id_mt_pe.rs:300-331).  One wave per read(-pair):
//   windows with stride d -> seq::has_no_n filter -> canonical choice on raw bytes (src/kmer.rs:221-243)
//   -> per-read set in first-occurrence order -> search_index_classic (:66-102) or search_index (:104-165)
//   with the reference's "absent row => count it once under no_hits_num and stop" rule applied in k-mer order.
// Two key paths, chosen per read (wave-uniform):
//   packed : the read holds no lower-case base and k <= 32 — bases are packed 2 bits each in LDS, a window's code
//            is three dword reads, the canonical choice is one integer compare, the set is an LDS hash table on the
//            64-bit code (exact), and the hash inputs are re-expanded to ASCII in registers;
//   bytes  : anything else (lower-case bases are hashed as they are; k > 32) — byte strings in
//            LDS, 32-bit tag match confirmed on the bytes.

// Sub-passes of a read's search whose row loads are in flight together.  Measured on configs[2] (1 M x 150 bp): 2 costs 24 more
// VGPRs than 1 and is no faster; 1 lets k_readid fit 96 VGPRs = 5 waves per SIMD, which is (6.13 vs 6.43-6.74 ms).
constexpr int kReadRunUnroll = 1;
// Counter planes per lane (drained every 2^planes - 1 additions; a lane adds one word per sub-pass).  k_readid comes in two
// register budgets: 3 planes in 96 VGPRs (5 waves per SIMD), or 2 planes in 80 VGPRs (6 waves per SIMD) when six workgroups'
// LDS fit a CU — configs[2], 1 M x 150 bp single-end: 5.8 ms vs 6.1 ms; paired reads need more LDS and stay on the first.
constexpr int kReadPlanes = 3;
constexpr int kReadPlanesDense = 2;


// The in-order search (read_id_mt_pe.rs:66-102 classic / :104-165 sampled) over a dense run of distinct k-mers: k-mer j (0 <= j < count, order index q_base + j) has its row
// numbers at ridx[s*stride + j].  U sub-passes (U * 64/LPR k-mers) have all their row loads issued before the first is
// consumed: a read's search is a chain of dependent gather rounds, and what bounds the kernel is how many of them there are.
template <int LOG_LPR, bool NARROW, int U, int PLANES = kReadPlanes, bool STRIPED = false>
__device__ __forceinline__ void readid_search_run(const uint64_t *mat, uint32_t rs, uint32_t n, uint32_t C, uint32_t S, const uint32_t *ridx,
                                                  uint32_t stride, uint32_t count, uint32_t q_base, uint32_t *hist, bool &stopped,
                                                  VCount<PLANES, NARROW> &vc, V16 &R, int lane, StripeRead sr = StripeRead{nullptr, nullptr}) {
    constexpr int LPR = 1 << LOG_LPR;
    constexpr int KPW = kWave / LPR;
    if (stopped || !count) return;
    const uint32_t col = lane & (LPR - 1);
    const uint32_t col_word = NARROW ? 0u : 2u * col;   // slices past the last colour word are zero padding of the row
    const uint32_t seeds_mask = n >= 32 ? ~0u : ((1u << n) - 1u);
#pragma unroll 1
    for (uint32_t j0 = 0; j0 < count; j0 += U * KPW) {
        uint32_t j[U];
        bool live[U];
        V16 a[U];
        uint32_t zm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            j[u] = j0 + u * KPW + (lane >> LOG_LPR);
            live[u] = j[u] < count;
            a[u] = V16{~0ull, ~0ull};
            zm[u] = 0;
        }
        gather_run<U, NARROW>(mat, rs, ridx, stride, j, live, col_word, n, a, zm);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u * KPW >= count) break;   // wave-uniform
            V16 w = a[u];
            if constexpr (NARROW) w.y = 0;
            uint32_t all_zero = zm[u];
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) all_zero &= __shfl_xor(all_zero, o, kWave);
            bool lv = live[u];
            if constexpr (STRIPED) {
                if (sr.zacc) {   // zero pass of a colour stripe: record, count nothing, never stop
                    if (lv && (lane & (LPR - 1)) == 0) sr.zacc[q_base + j[u]] &= (all_zero & seeds_mask);
                    continue;
                }
                if (sr.zin) all_zero = lv ? sr.zin[q_base + j[u]] : 0u;   // absent = all-zero in every stripe
            }
            const bool miss = lv && (all_zero & seeds_mask);
            const uint64_t bm = __ballot(miss);
            // keep only the k-mers before the first absent row (lane order == k-mer order in a sub-pass)
            if (bm) lv = lv && (lane >> LOG_LPR) < (__builtin_ctzll(bm) >> LOG_LPR);
            if (!lv) { w.x = 0; w.y = 0; }
            if (S > 0) {
                const uint32_t q = q_base + j[u];
                if (q_base + j0 + u * KPW < S) {   // wave-uniform: this sub-pass holds some of the first S k-mers
                    V16 ra = q < S ? w : V16{0, 0};
#pragma unroll
                    for (int o = LPR; o < kWave; o <<= 1) {
                        ra.x |= __shfl_xor(ra.x, o, kWave);
                        ra.y |= __shfl_xor(ra.y, o, kWave);
                    }
                    R.x |= ra.x; R.y |= ra.y;
                }
                if (q >= S) { w.x &= R.x; w.y &= R.y; }
            }
            vc.add(w);
            if (vc.full()) vc.drain(hist, col_word);
            if (bm) {
                stopped = true;
                if (lane == 0) hist[C] += 1;  // *report.entry(no_hits_num) += 1; break
                return;
            }
        }
    }
}

// The per-wave LDS regions of k_readid / k_readid_bytes are sized by the caller's longest read (bases_cap = that + 16 rounded up,
// win_cap windows).  The device-pointer entry points cannot trust those maxima, so this kernel runs first: one thread per read;
// a read beyond either maximum is marked skip = 1 (the LDS kernels then leave it alone), status 3, n_kmers 0, and its report row
// is zeroed (report_width == 0: no row to zero).  Keeping the test out of the LDS kernels keeps their register budget.
__global__ __launch_bounds__(256) void k_readid_check_caps(ReadIdParams p, uint8_t *skip) {
    const uint64_t read = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (read >= p.n_reads) return;
    const uint64_t s0 = p.read_seq0[read], s1 = p.read_seq0[read + 1];
    const uint64_t tb = s1 > s0 ? p.seq_off[s1] - p.seq_off[s0] : 0;
    uint64_t win = 0;
    for (uint64_t s = s0; s < s1; ++s) {
        const uint64_t len = p.seq_off[s + 1] - p.seq_off[s];
        if (len >= p.k) win += (len - p.k) / p.stride_d + 1;
    }
    const bool over = tb + 16u > p.bases_cap || win > p.win_cap;
    skip[read] = over ? 1 : 0;
    if (over) {
        p.status[read] = 3;
        p.n_kmers[read] = 0;
        if (p.report_width) for (uint32_t c = 0; c < p.report_width; ++c) p.report[read * (uint64_t)p.report_width + c] = 0;
    }
}
hipError_t launch_readid_check_caps(const ReadIdParams &p, uint8_t *skip, hipStream_t stream) {
    if (p.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_readid_check_caps, dim3((unsigned)((p.n_reads + 255) / 256)), dim3(256), 0, stream, p, skip);
    return hipGetLastError();
}

// Output of one read: drain the counters, copy the histogram to the report row, clear it for the next read.
template <bool NARROW, bool WIDE, int PLANES = kReadPlanes, bool STRIPED = false>
__device__ __forceinline__ void readid_finish_read(VCount<PLANES, NARROW> &vc, uint32_t *hist, uint32_t col_word, uint32_t *row_out,
                                                   uint32_t C, int lane, const ReadIdParams &p) {
    if constexpr (!WIDE) {
        vc.drain(hist, col_word);
        wave_lds_fence();
        if constexpr (STRIPED) {
            if (p.zero_acc) {          // zero pass: nothing to report
                for (uint32_t c = lane; c <= C; c += kWave) hist[c] = 0;
            } else {                   // count pass of a stripe: its colours inside the wide row; the no-hits entry from one stripe only
                for (uint32_t c = lane; c < C; c += kWave) { row_out[p.colour_base + c] = hist[c]; hist[c] = 0; }
                if (lane == 0) { if (p.write_nohits) row_out[p.report_width - 1] = hist[C]; hist[C] = 0; }
            }
        } else {
            for (uint32_t c = lane; c <= C; c += kWave) { row_out[c] = hist[c]; hist[c] = 0; }
        }
    }
}

// ---- k_readid: reads without lower-case bases, k <= 32.  Per-wave LDS, not WIDE: rall (win_cap*n) | table keys +
// indices | 2-bit bases | bad-base bits — the per-colour histogram of the search phase lives IN the table's region (the table is
// dead once the read's set is complete; the region is max(table, histogram) bytes), which is what lets paired 150-bp reads run
// 5 waves per SIMD instead of 4.  WIDE: ridx (64*n) | hist | table | ... (the search interleaves with the set building).
// A read with a lower-case base (its case must be kept) is appended to p.redo_list for k_readid_bytes.
// PACKED (p.idx_bits > 0; 2k + idx_bits <= 63): a table slot is ONE u64, canonical code << idx_bits | smallest window index — 8
// instead of 12 bytes per slot, which is what lets paired 150-bp reads at k = 21 keep six waves per SIMD.
// SLOT4 (p.slot4): a table slot is ONE u32 — the smallest POSITION (in bases from the aligned start of the read: it grows with the window
// index) of a window holding the slot's k-mer; the code itself is not stored but read back from the packed bases when a probe meets an
// occupied slot.  4 instead of 12 bytes per slot for the k-mers whose codes leave no room for an index beside them (k = 28..32): read pairs
// at those k keep six waves per SIMD.
template <int LOG_LPR, bool NARROW, bool WIDE, bool MINI, bool DENSE, bool STRIPED = false, bool PACKED = false, bool SLOT4 = false>
__global__ __launch_bounds__(kBlock, DENSE ? 6 : 5) void k_readid(ReadIdParams p) {
    static_assert(!(PACKED && SLOT4) && !(SLOT4 && (MINI || WIDE || STRIPED)), "one slot layout; position slots: whole k-mers, narrow rows, whole index");
    constexpr int PLANES = DENSE ? kReadPlanesDense : kReadPlanes;
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t RS = NARROW ? 1u : 2u * LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = (int)wave_uniform((uint32_t)threadIdx.x >> 6);   // (the per-read scalars below then live in SGPRs)
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;
    const uint32_t klen = MINI ? p.m_size : k;   // length of the hashed key

    uint8_t *wb = smem + (size_t)wave * p.wave_bytes;
    uint4 *s_stage = reinterpret_cast<uint4 *>(wb);                            // stage_bytes: the NEXT read's bases, raw (LDS-DMA lands here)
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wb + p.stage_bytes);         // WIDE: 64*n, this chunk's rows
    const uint32_t rcap = p.win_cap;
    constexpr bool SEPARATE = WIDE || !CID_READID_ALIAS;                       // histogram in a region of its own
    uint32_t *rall = ridx + (WIDE ? kWave * n : 0u) + (SEPARATE ? p.hist_pad : 0u);   // not WIDE: win_cap*n, rows of the read's distinct k-mers in order
    unsigned long long *t_key = reinterpret_cast<unsigned long long *>(rall + (WIDE ? 0u : rcap * n));   // table_slots
    uint32_t *t_idx = reinterpret_cast<uint32_t *>(t_key + p.table_slots);     // table_slots
    uint32_t *hist = SEPARATE ? ridx + (WIDE ? kWave * n : 0u) : reinterpret_cast<uint32_t *>(t_key);   // hist_pad; else: shares the table's region
    constexpr uint32_t SLOT = SLOT4 ? 4u : PACKED ? 8u : 12u;
    uint32_t *t_pos = reinterpret_cast<uint32_t *>(t_key);                     // SLOT4: table_slots positions
    const uint32_t region = SEPARATE ? SLOT * p.table_slots : (SLOT * p.table_slots > 4u * p.hist_pad ? SLOT * p.table_slots : 4u * p.hist_pad);
    uint32_t *s_pack = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(t_key) + region);   // bases_cap/16 + 4 dwords, 16 bases each
    uint32_t *s_bad = s_pack + (p.bases_cap / 16 + 4);                         // bases_cap/32 + 4 dwords, 1 bit per base

    if constexpr (SEPARATE)
        for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;

    const uint32_t col_word = NARROW ? 0u : 2u * (lane & (LPR - 1));
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t tmask = p.table_slots - 1;

    const uint64_t r_begin = (uint64_t)blockIdx.x * p.reads_per_block;
    const uint64_t r_end = r_begin + p.reads_per_block < p.n_reads ? r_begin + p.reads_per_block : p.n_reads;
    // Where a read starts and how long it is takes two dependent loads (read_seq0 -> seq_off) before its bases can be asked for: lane l
    // fetches them for the wave's l-th read, 64 reads at a time, and each read picks its own up with v_readlane.  The bases of the
    // read AFTER the current one are asked for (LDS-DMA into s_stage) before the current read's search, so that they arrive behind
    // its row gathers; a read then starts from LDS.
    uint64_t L_g0 = 0;                        // lane l: seq_off[first sequence] of read r_begin + wave + (batch * 64 + l) * waves
    uint32_t L_first = 0, L_tb = 0, L_mates = 0;   // its first sequence's length, all its bases, its sequences | skip << 31
    bool staged = false;                      // s_stage holds this read's first 64 pieces
    uint32_t it = kWave;
    for (uint64_t read = r_begin + wave; read < r_end; read += waves, ++it) {
        if (it == (uint32_t)kWave) {
            it = 0;
            const uint64_t r = read + (uint64_t)lane * waves;
            L_g0 = 0; L_first = L_tb = L_mates = 0;
            if (r < r_end) {
                const uint64_t a = p.read_seq0[r], b = p.read_seq0[r + 1];
                L_g0 = p.seq_off[a];
                L_first = b > a ? (uint32_t)(p.seq_off[a + 1] - L_g0) : 0u;
                L_tb = (uint32_t)(p.seq_off[b] - L_g0);
                L_mates = (uint32_t)(b - a < 0x7FFFFFFFull ? b - a : 0x7FFFFFFFull) | ((p.skip && p.skip[r]) ? 0x80000000u : 0u);
            }
        }
        const bool from_stage = staged;
        staged = false;
        const uint32_t mates_skip = (uint32_t)__builtin_amdgcn_readlane((int)L_mates, (int)it);
        if (mates_skip >> 31) continue;
        wave_lds_fence();
        const uint32_t n_mates = mates_skip;
        const uint64_t g0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(L_g0 >> 32), (int)it) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)L_g0, (int)it);
        const uint32_t first_len = (uint32_t)__builtin_amdgcn_readlane((int)L_first, (int)it);
        constexpr bool striped = STRIPED;   // striped passes: the caller zeroes the (wide) report once, rows are only added to
        uint32_t *row_out = striped ? p.report + read * (uint64_t)p.report_width : p.report + read * (uint64_t)(C + 1);
        StripeRead sr{nullptr, nullptr};
        if constexpr (STRIPED) sr = StripeRead{p.zero_acc ? p.zero_acc + p.zero_start[read] : nullptr, p.zero_in ? p.zero_in + p.zero_start[read] : nullptr};
        if (n_mates == 0 || first_len < k) {  // too_short: only the first mate is tested (read_id_mt_pe.rs:305)
            if constexpr (!WIDE)  // (wide rows: the host zeroes the whole report before the launch)
                if (!striped) for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) { p.n_kmers[read] = 0; p.status[read] = 1; }
            continue;
        }
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane((int)L_tb, (int)it);
        // The read's bases come in as aligned 16-byte pieces, one per lane (150 bases: ten or eleven lanes, ONE load), and go to LDS
        // as 2-bit codes (16 per dword) + one "not a base" bit each; positions count from the aligned start (`shift` bytes before the
        // read).  Nothing else of the read is kept: hash inputs are re-expanded from the codes.
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p.bases + g0);
        const uint32_t shift = (uint32_t)(a0 & 15u), span = shift + tb;
        const uint4 *src = reinterpret_cast<const uint4 *>(a0 - shift);
        bool lower = false;
        for (uint32_t j0 = 0; j0 * 16 < span + 64; j0 += kWave) {
            const uint32_t j = j0 + lane;
            uint32_t code = 0, bad = 0xFFFFu;
            if (j * 16 < span) {
                uint4 q;
                if (from_stage && j0 == 0) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the LDS-DMA of the previous iteration has landed
                    q = s_stage[lane];
                } else {
                    q = src[j];
                }
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                uint32_t low = 0;
                bad = 0;
#pragma unroll
                for (int dw = 0; dw < 4; ++dw) {
                    uint32_t c8, b4, l4;
                    pack_four_bases(w[dw], c8, b4, l4);
                    code |= c8 << (8 * dw);
                    bad |= b4 << (4 * dw);
                    low |= l4 << (4 * dw);
                }
                const uint32_t lo = shift > j * 16 ? (shift - j * 16 < 16u ? shift - j * 16 : 16u) : 0u;   // this lane's bytes [lo, hi) are the read's
                const uint32_t hi = span - j * 16 < 16u ? span - j * 16 : 16u;
                bad = (bad | ~(((1u << hi) - 1u) & ~((1u << lo) - 1u))) & 0xFFFFu;
                lower = lower || (low & ~bad) != 0;   // a lower-case base: its case must be kept
            }
            const uint32_t bad_hi = __shfl_down(bad, 1, kWave);
            if (j * 16 < span + 64) {
                s_pack[j] = code;
                if (!(lane & 1)) s_bad[j >> 1] = bad | (bad_hi << 16);
            }
        }
        if (__any(lower)) {   // the byte-string kernel takes this read
            if (lane == 0) p.redo_list[atomicAdd(p.redo_count, 1u)] = (uint32_t)read;
            continue;
        }
        // the next read's bases: on their way (LDS-DMA, no registers) while this read is listed and searched
        if (p.stage_bytes && it + 1 < (uint32_t)kWave && read + waves < r_end) {
            const uint32_t nms = (uint32_t)__builtin_amdgcn_readlane((int)L_mates, (int)(it + 1));
            const uint64_t ng0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(L_g0 >> 32), (int)(it + 1)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)L_g0, (int)(it + 1));
            const uintptr_t na = reinterpret_cast<uintptr_t>(p.bases + ng0);
            const uint32_t nspan = (uint32_t)(na & 15u) + (uint32_t)__builtin_amdgcn_readlane((int)L_tb, (int)(it + 1));
            if (nms - 1u < 0x7FFFFFFFu && nspan <= p.stage_bytes) {   // a read with sequences, not skipped, that fits the stage
                wave_lds_fence();   // this read's pieces have been taken out of s_stage
                if ((uint32_t)lane * 16u < nspan)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void *>((na & ~(uintptr_t)15) + 16u * (uint32_t)lane),
                                                     (__attribute__((address_space(3))) void *)s_stage, 16, 0, 0);
                staged = true;
            }
        }
        if constexpr (SLOT4) {
            for (uint32_t t = lane; t < p.table_slots; t += kWave) t_pos[t] = ~0u;
        } else {
            for (uint32_t t = lane; t < p.table_slots; t += kWave) { t_key[t] = ~0ull; if constexpr (!PACKED) t_idx[t] = ~0u; }
        }
        wave_lds_fence();

        uint32_t nd = 0;       // distinct k-mers so far == the reference's `counter`
        bool stopped = false;  // an absent row was met: nothing after it is searched
        VCount<PLANES, NARROW> vc;
        vc.clear();
        V16 R{0, 0};           // colours seen in the first S k-mers (this lane's slice)
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        // The read's windows as ONE sequence over its mates (mate 1's first: the first-occurrence order), cut into chunks of 64:
        // a chunk may straddle the mate boundary, so 2 x 130 windows are 5 chunks, not 6 (each chunk is a wave-wide pass).
        // A mate shorter than k contributes nothing.
        const uint32_t nw0 = (first_len - k) / p.stride_d + 1;
        uint32_t off1 = 0, nw1 = 0;
        if (n_mates >= 2) {
            off1 = first_len;
            const uint32_t len1 = n_mates == 2 ? tb - first_len : (uint32_t)(wave_uniform(p.seq_off[wave_uniform(p.read_seq0[read]) + 2]) - g0) - first_len;
            nw1 = len1 >= k ? (len1 - k) / p.stride_d + 1 : 0u;
        }
        uint32_t wtot = nw0 + nw1;
        const uint64_t s0 = n_mates > 2 ? wave_uniform(p.read_seq0[read]) : 0ull, s1 = s0 + n_mates;   // (only reads of three or more sequences look at these)
        for (uint64_t s = s0 + 2; s < s1; ++s) {   // reads of more than two mates (never from the CLI): counted here, located below
            const uint32_t len = (uint32_t)wave_uniform(p.seq_off[s + 1] - p.seq_off[s]);
            if (len >= k) wtot += (len - k) / p.stride_d + 1;
        }
        {
            for (uint32_t c0 = 0; c0 < wtot; c0 += kWave) {
                const uint32_t wi = c0 + lane;                 // index in the read's window sequence == first-occurrence order index
                uint32_t pos = wi < nw0 ? wi * p.stride_d : off1 + (wi - nw0) * p.stride_d;
                if (n_mates > 2 && c0 + kWave > nw0 + nw1) {   // (wave-uniform) this chunk reaches into a third or later mate
                    uint32_t base = nw0 + nw1;
                    for (uint64_t s = s0 + 2; s < s1; ++s) {
                        const uint32_t len = (uint32_t)(p.seq_off[s + 1] - p.seq_off[s]);
                        if (len < k) continue;
                        const uint32_t nws = (len - k) / p.stride_d + 1;
                        if (wi >= base && wi < base + nws) pos = (uint32_t)(p.seq_off[s] - g0) + (wi - base) * p.stride_d;
                        base += nws;
                    }
                }
                pos += shift;
                constexpr uint32_t wbase = 0;
                const uint32_t nw = wtot;
                if constexpr (WIDE) wave_lds_fence();  // the previous chunk's gathers are done with ridx
                bool valid = wi < nw;
                uint64_t lsb = 0;
                if (valid) {
                    valid = bits_at(s_bad, pos, k) == 0;                   // seq::has_no_n over the window
                    lsb = bits_at(s_pack, 2 * pos, 2 * k);
                }
                uint64_t msb = 0;
                uint64_t canon = canonical_code(lsb, k, &msb);
                if constexpr (MINI) {  // .mxi: the set holds the k-mers' minimizers (kmer.rs:363-394)
                    msb = minimizer_code(msb, k, klen);
                    canon = rev_fields(msb, klen);
                }
                // exact set with first-occurrence order: slot key = canonical code, slot value = smallest window index
                uint32_t slot = (uint32_t)((msb * 0x9E3779B97F4A7C15ull) >> 40) & tmask;
                if constexpr (SLOT4) {
                    if (valid) {
                        while (true) {
                            const uint32_t old = atomicCAS(&t_pos[slot], ~0u, pos);
                            if (old == ~0u) break;
                            uint64_t other = 0;   // the k-mer of the window that sits there
                            canonical_code(bits_at(s_pack, 2 * old, 2 * k), k, &other);
                            if (other == msb) { atomicMin(&t_pos[slot], pos); break; }   // same k-mer: keep the earlier window
                            slot = (slot + 1) & tmask;
                        }
                    }
                } else if constexpr (PACKED) {
                    if (valid) {
                        const unsigned long long mine = ((unsigned long long)msb << p.idx_bits) | (unsigned long long)wi;
                        while (true) {
                            const unsigned long long old = atomicCAS(&t_key[slot], ~0ull, mine);
                            if (old == ~0ull) break;
                            if ((old >> p.idx_bits) == msb) { atomicMin(&t_key[slot], mine); break; }   // same code: keep the earlier window
                            slot = (slot + 1) & tmask;
                        }
                    }
                } else if (valid) {
                    while (true) {
                        const unsigned long long old = atomicCAS(&t_key[slot], ~0ull, (unsigned long long)msb);
                        if (old == ~0ull || old == msb) break;
                        slot = (slot + 1) & tmask;
                    }
                    atomicMin(&t_idx[slot], wbase + wi);
                }
                wave_lds_fence();
                const bool distinct = valid && (SLOT4 ? t_pos[slot] == pos : PACKED ? (uint32_t)(t_key[slot] & ((1ull << p.idx_bits) - 1ull)) == wi : t_idx[slot] == wbase + wi);
                const uint64_t dmask = __ballot(distinct);
                if (distinct) {   // WIDE: this chunk's slot; else the read's list at the k-mer's order index
                    uint32_t *dst = WIDE ? ridx + lane : rall + nd + (uint32_t)__popcll(dmask & lt_mask);
                    const uint32_t st = WIDE ? (uint32_t)kWave : rcap;
                    // (the 6-waves-per-SIMD build is compiled for the published hash only: launch_readid sends other variants to the 5-wave build)
                    xxh3_seeds_from(CodeReader{canon}, klen, n, DENSE ? HashSel::published() : HashSel::of(p.mod),
                                    [&](uint32_t sd, uint64_t h) { dst[sd * st] = (uint32_t)mod_m(h, p.mod); });
                }
                if constexpr (WIDE) {
                    wave_lds_fence();
                    uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                    readid_search_chunk_wide<STRIPED>(p.mat, p.rs, p.w64, n, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane,
                                                      STRIPED ? p.report_width - 1 : C, sr, STRIPED ? p.colour_base : 0u, STRIPED ? p.write_nohits : 1u);
                }
                nd += (uint32_t)__popcll(dmask);
            }
        }
        if constexpr (!WIDE) {
            // the set is complete: the table's region becomes the histogram; search the nd k-mers in order
            wave_lds_fence();
            if constexpr (!SEPARATE) {
                for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
                wave_lds_fence();
            }
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll, PLANES, STRIPED>(p.mat, RS, n, C, S, rall, rcap, nd, 0u, hist, stopped, vc, R, lane, sr);
        }
        readid_finish_read<NARROW, WIDE, PLANES, STRIPED>(vc, hist, col_word, row_out, C, lane, p);
        if (lane == 0) { p.n_kmers[read] = nd; p.status[read] = 0; }
    }
}

// ---- k_readid_bytes: the same per-read work on byte strings — reads with lower-case bases (p.redo_list, filled by
// k_readid) or every read when k > 32 (p.redo_list == NULL).  Per-wave LDS: bases | ridx (64*n) | hist | rall | tags |
// window infos | k-mer image | minimizer image + distinct minimizer strings (.mxi).  A 32-bit tag match is confirmed on the bytes.
template <int LOG_LPR, bool NARROW, bool WIDE, bool STRIPED = false>
__global__ __launch_bounds__(kBlock, 2) void k_readid_bytes(ReadIdParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t RS = NARROW ? 1u : 2u * LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;

    uint8_t *wb = smem + (size_t)wave * p.wave_bytes;
    uint8_t *s_bases = wb;                                                     // bases_cap
    uint32_t *ridx = reinterpret_cast<uint32_t *>(wb + p.bases_cap);           // 64*n: this chunk's rows
    uint32_t *hist = ridx + kWave * n;                                         // hist_pad
    uint32_t *rall = hist + p.hist_pad;                                        // not WIDE: win_cap*n
    const uint32_t rcap = p.win_cap;
    uint32_t *s_tag = rall + (WIDE ? 0u : rcap * n);                           // win_cap
    uint32_t *s_info = s_tag + p.win_cap;                                      // win_cap
    uint32_t *img = s_info + p.win_cap;                                        // kmer_img_bytes(k)
    uint8_t *img8 = reinterpret_cast<uint8_t *>(img);
    uint32_t *mimg = img + kmer_img_bytes(k) / 4;                              // .mxi only: kmer_img_bytes(m_size) minimizer image
    uint8_t *mimg8 = reinterpret_cast<uint8_t *>(mimg);
    uint8_t *s_mstr = mimg8 + kmer_img_bytes(p.m_size ? p.m_size : 1);         // .mxi only: win_cap x m_size distinct minimizers

    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
    const uint32_t col_word = NARROW ? 0u : 2u * (lane & (LPR - 1));
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    const uint64_t n_items = p.redo_list ? (uint64_t)*p.redo_count : p.n_reads;
    for (uint64_t item = (uint64_t)blockIdx.x * waves + wave; item < n_items; item += (uint64_t)gridDim.x * waves) {
        const uint64_t read = p.redo_list ? (uint64_t)p.redo_list[item] : item;
        if (p.skip && p.skip[read]) continue;
        wave_lds_fence();
        const uint64_t s0 = p.read_seq0[read], s1 = p.read_seq0[read + 1];
        const uint64_t g0 = p.seq_off[s0];
        const uint32_t first_len = s1 > s0 ? (uint32_t)(p.seq_off[s0 + 1] - g0) : 0u;
        constexpr bool striped = STRIPED;   // striped passes: the caller zeroes the (wide) report once, rows are only added to
        uint32_t *row_out = striped ? p.report + read * (uint64_t)p.report_width : p.report + read * (uint64_t)(C + 1);
        StripeRead sr{nullptr, nullptr};
        if constexpr (STRIPED) sr = StripeRead{p.zero_acc ? p.zero_acc + p.zero_start[read] : nullptr, p.zero_in ? p.zero_in + p.zero_start[read] : nullptr};
        if (s1 == s0 || first_len < k) {  // too_short: only the first mate is tested (read_id_mt_pe.rs:305)
            if constexpr (!WIDE)
                if (!striped) for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) { p.n_kmers[read] = 0; p.status[read] = 1; }
            continue;
        }
        const uint32_t tb = (uint32_t)(p.seq_off[s1] - g0);
        for (uint32_t i = lane; i < tb; i += kWave) s_bases[i] = p.bases[g0 + i];
        wave_lds_fence();

        uint32_t nd = 0;
        bool stopped = false;
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        for (uint64_t s = s0; s < s1; ++s) {
            const uint32_t off = (uint32_t)(p.seq_off[s] - g0);
            const uint32_t len = (uint32_t)(p.seq_off[s + 1] - p.seq_off[s]);
            if (len < k) continue;
            const uint32_t nw = (len - k) / p.stride_d + 1;
            for (uint32_t c0 = 0; c0 < nw; c0 += kWave) {
                const uint32_t wi = c0 + lane;
                const uint32_t pos = off + wi * p.stride_d;
                wave_lds_fence();  // the previous chunk is done with ridx / img
                bool valid = wi < nw;
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) valid = valid && good_base(s_bases[pos + t]);
                uint32_t rc = 1;  // palindromes take the reverse-complement branch (same string)
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) {
                        const uint32_t f = s_bases[pos + t], r = comp_base(s_bases[pos + k - 1 - t]);
                        if (f != r) { rc = f < r ? 0u : 1u; break; }
                    }
                const uint32_t info = pos | (rc << 31);
                if (valid)
                    for (uint32_t t = 0; t < k; ++t) img8[(uint32_t)lane * k + t] = (uint8_t)canon_byte(s_bases, info, k, t);
                wave_lds_fence();
                uint32_t tag = 0;
                bool dup = false;
                const uint64_t vmask = __ballot(valid);
                uint64_t dmask;
                if (p.m_size) {  // .mxi: the key is the (upper-cased) minimizer of the canonical k-mer; strings kept in s_mstr
                    const uint32_t m = p.m_size;
                    if (valid) {
                        const uint8_t *seq = img8 + (uint32_t)lane * k;
                        const uint32_t cand = find_minimizer_bytes(seq, k, m);
                        for (uint32_t t = 0; t < m; ++t) mimg8[(uint32_t)lane * m + t] = upper_base(mini_byte(seq, cand, m, t));
                    }
                    wave_lds_fence();
                    if (valid)
                        xxh3_seeds(mimg, (uint32_t)lane * m, m, n, HashSel::of(p.mod), [&](uint32_t sd, uint64_t h) {
                            if (sd == 0) tag = (uint32_t)h ^ (uint32_t)(h >> 32);
                            ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod);
                        });
                    for (uint32_t q = 0; q < nd; ++q) {
                        if (valid && !dup && s_tag[q] == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < m && same; ++t) same = s_mstr[q * m + t] == mimg8[(uint32_t)lane * m + t];
                            dup = same;
                        }
                    }
                    for (int j = 0; j < kWave - 1; ++j) {
                        if (!((vmask >> j) & 1ull)) continue;
                        const uint32_t tj = __builtin_amdgcn_readlane(tag, j);
                        if (valid && !dup && j < lane && tj == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < m && same; ++t) same = mimg8[(uint32_t)j * m + t] == mimg8[(uint32_t)lane * m + t];
                            dup = same;
                        }
                    }
                    dmask = __ballot(valid && !dup);
                    if (valid && !dup) {
                        const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                        s_tag[q] = tag;
                        for (uint32_t t = 0; t < m; ++t) s_mstr[q * m + t] = mimg8[(uint32_t)lane * m + t];
                    }
                } else {
                    if (valid)
                        xxh3_seeds(img, (uint32_t)lane * k, k, n, HashSel::of(p.mod), [&](uint32_t sd, uint64_t h) {
                            if (sd == 0) tag = (uint32_t)h ^ (uint32_t)(h >> 32);
                            ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod);
                        });
                    for (uint32_t q = 0; q < nd; ++q) {  // against the distinct k-mers of earlier chunks
                        if (valid && !dup && s_tag[q] == tag) {
                            const uint32_t oi = s_info[q];
                            bool same = true;
                            for (uint32_t t = 0; t < k && same; ++t) same = canon_byte(s_bases, oi, k, t) == img8[(uint32_t)lane * k + t];
                            dup = same;
                        }
                    }
                    for (int j = 0; j < kWave - 1; ++j) {  // against lower lanes of this chunk
                        if (!((vmask >> j) & 1ull)) continue;
                        const uint32_t tj = __builtin_amdgcn_readlane(tag, j);
                        const uint32_t ij = __builtin_amdgcn_readlane(info, j);
                        if (valid && !dup && j < lane && tj == tag) {
                            bool same = true;
                            for (uint32_t t = 0; t < k && same; ++t) same = canon_byte(s_bases, ij, k, t) == img8[(uint32_t)lane * k + t];
                            dup = same;
                        }
                    }
                    dmask = __ballot(valid && !dup);
                    if (valid && !dup) {
                        const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                        s_tag[q] = tag;
                        s_info[q] = info;
                    }
                }
                if constexpr (WIDE) {
                    wave_lds_fence();
                    uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                    readid_search_chunk_wide<STRIPED>(p.mat, p.rs, p.w64, n, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane,
                                                      STRIPED ? p.report_width - 1 : C, sr, STRIPED ? p.colour_base : 0u, STRIPED ? p.write_nohits : 1u);
                } else if (valid && !dup) {
                    const uint32_t q = nd + (uint32_t)__popcll(dmask & lt_mask);
                    for (uint32_t sd = 0; sd < n; ++sd) rall[sd * rcap + q] = ridx[sd * kWave + lane];
                }
                nd += (uint32_t)__popcll(dmask);
            }
        }
        if constexpr (!WIDE) {
            wave_lds_fence();
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll, kReadPlanes, STRIPED>(p.mat, RS, n, C, S, rall, rcap, nd, 0u, hist, stopped, vc, R, lane, sr);
        }
        readid_finish_read<NARROW, WIDE, kReadPlanes, STRIPED>(vc, hist, col_word, row_out, C, lane, p);
        if (lane == 0) { p.n_kmers[read] = nd; p.status[read] = 0; }
    }
}

struct BaseReader {  // a key that lives in HBM as a stretch of the read (forward or reverse complement), see k_general_keys
    const uint8_t *b;
    uint32_t len, rc, upper;
    __device__ __forceinline__ uint32_t rd8(uint32_t o) const {
        uint32_t c = rc ? comp_base(b[len - 1 - o]) : (uint32_t)b[o];
        if (upper) c = upper_base(c);
        return c;
    }
    __device__ __forceinline__ uint32_t rd32(uint32_t o) const { return rd8(o) | (rd8(o + 1) << 8) | (rd8(o + 2) << 16) | (rd8(o + 3) << 24); }
    __device__ __forceinline__ uint64_t rd64(uint32_t o) const { return (uint64_t)rd32(o) | ((uint64_t)rd32(o + 4) << 32); }
};

template <int LOG_LPR, bool NARROW, bool WIDE = false, bool STRIPED = false>
__global__ __launch_bounds__(kBlock, 5) void k_readid_list(ReadIdListParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;
    uint32_t *ridx = reinterpret_cast<uint32_t *>(smem + (size_t)wave * p.wave_bytes);
    uint32_t *hist = ridx + kWave * n;
    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
    const uint32_t col_word = NARROW ? 0u : 2u * (lane & ((1 << LOG_LPR) - 1));
    for (uint64_t read = (uint64_t)blockIdx.x * waves + wave; read < p.n_reads; read += (uint64_t)gridDim.x * waves) {
        wave_lds_fence();
        // striped passes: the caller zeroes the (report_width-wide) report once, rows are only added to
        uint32_t *row_out = STRIPED ? p.report + read * (uint64_t)p.report_width : p.report + read * (uint64_t)(C + 1);
        if (p.status[read] == 2) continue;
        if (p.status[read] == 1) {  // too_short, decided on the host side of the call
            if constexpr (!WIDE && !STRIPED)
                for (uint32_t c = lane; c <= C; c += kWave) row_out[c] = 0;
            if (lane == 0) p.n_kmers[read] = 0;
            continue;
        }
        const uint64_t d0 = p.list_start[read], d1 = p.list_start[read + 1];
        StripeRead sr{nullptr, nullptr};
        if constexpr (STRIPED) sr = StripeRead{p.zero_acc ? p.zero_acc + p.zero_start[read] : nullptr, p.zero_in ? p.zero_in + p.zero_start[read] : nullptr};
        uint32_t nd = 0;
        bool stopped = false;
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};
        if constexpr (WIDE) {
            uint64_t *s_R = reinterpret_cast<uint64_t *>(hist) + p.rs;
            for (uint32_t w = lane; w < p.rs; w += kWave) s_R[w] = 0;
            wave_lds_fence();
        }
        for (uint64_t c0 = d0; c0 < d1 && !stopped; c0 += kWave) {
            const bool have = c0 + lane < d1;
            wave_lds_fence();
            if (have) {
                const uint64_t e = p.list_codes[c0 + lane];
                if (p.bases) {
                    xxh3_seeds_from(BaseReader{p.bases + (e >> 1), k, (uint32_t)(e & 1ull), p.upper}, k, n, HashSel::of(p.mod),
                                    [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod); });
                } else {
                    xxh3_seeds_from(CodeReader{rev_fields(e, k)}, k, n, HashSel::of(p.mod),
                                    [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + lane] = (uint32_t)mod_m(h, p.mod); });
                }
            }
            const uint64_t dmask = __ballot(have);
            wave_lds_fence();
            if constexpr (WIDE) {
                uint64_t *s_words = reinterpret_cast<uint64_t *>(hist), *s_R = s_words + p.rs;
                readid_search_chunk_wide<STRIPED>(p.mat, p.rs, p.w64, n, S, ridx, s_words, s_R, row_out, dmask, nd, stopped, lane,
                                                  STRIPED ? p.report_width - 1 : C, sr, STRIPED ? p.colour_base : 0u, STRIPED ? p.write_nohits : 1u);
            } else {   // the chunk's entries are dense from lane 0
                readid_search_run<LOG_LPR, NARROW, kReadRunUnroll, kReadPlanes, STRIPED>(p.mat, NARROW ? 1u : 2u << LOG_LPR, n, C, S, ridx, (uint32_t)kWave,
                                                                                         (uint32_t)__popcll(dmask), nd, hist, stopped, vc, R, lane, sr);
            }
            nd += (uint32_t)__popcll(dmask);
        }
        if constexpr (!WIDE) {
            vc.drain(hist, col_word);
            wave_lds_fence();
            if constexpr (STRIPED) {
                if (p.zero_acc) {   // zero pass: nothing was counted
                    for (uint32_t c = lane; c <= C; c += kWave) hist[c] = 0;
                } else {            // this stripe's colours inside the wide row; the no-hits entry from one stripe only
                    for (uint32_t c = lane; c < C; c += kWave) { row_out[p.colour_base + c] = hist[c]; hist[c] = 0; }
                    if (lane == 0) { if (p.write_nohits) row_out[p.report_width - 1] = hist[C]; hist[C] = 0; }
                }
            } else {
                for (uint32_t c = lane; c <= C; c += kWave) { row_out[c] = hist[c]; hist[c] = 0; }
            }
        }
        if (lane == 0) p.n_kmers[read] = (uint32_t)(d1 - d0);
    }
}

// ---- k_readid_slices: the long-read path's search.  A read's distinct k-mers in first-occurrence order are the windows whose bit is
// set in the first-occurrence bitmap, in window order; they are walked by one wave per SLICE (a stretch of consecutive windows), so a
// 100 kb read is searched by a few dozen waves instead of one.  No list is materialised: the wave takes 64 consecutive windows, the
// flagged lanes are this chunk's k-mers in order (a ballot compacts them), their rows go to LDS and through the same in-order run as
// k_readid's.  The reference's rules run over a read's k-mers in order (read_id_mt_pe.rs:66-165):
//   * `-B S`: the colour set R of the first S k-mers — a later slice gathers those S k-mers again to know R (S <= 64 when reads are cut);
//   * an absent row ends the read: a slice records whether it stopped, k_readid_combine adds the slices' rows up to the first that did.
// The codes and bitmap words of the NEXT 64 windows are asked for before the current chunk's rows: under load a dependent global load
// costs as much as a gather round (k_readid_list pays one per 64 k-mers).
template <int LOG_LPR, bool NARROW, bool BYTES = false>
__global__ __launch_bounds__(kBlock, 5) void k_readid_slices(ReadIdSliceParams p) {
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr uint32_t RS = NARROW ? 1u : 2u << LOG_LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const uint32_t C = p.n_colors, k = p.k, n = p.n_hash, S = p.start_sample;
    uint32_t *ridx = reinterpret_cast<uint32_t *>(smem + (size_t)wave * p.wave_bytes);
    uint32_t *hist = ridx + kWave * n;
    for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
    const uint32_t col_word = NARROW ? 0u : 2u * (lane & ((1 << LOG_LPR) - 1));
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    auto rank = [&](uint32_t w) -> uint32_t {
        return p.word_prefix[w >> 5] + (uint32_t)__popc(p.bitmap[w >> 5] & ((1u << (w & 31u)) - 1u));
    };
    auto hash_to = [&](uint64_t code, uint32_t q) {
        if constexpr (BYTES)   // a soft-masked read: the k-mer is a stretch of the read, forward or reverse complement, its case kept
            xxh3_seeds_from(BaseReader{p.bases + (code >> 1), k, (uint32_t)(code & 1ull), 0u}, k, n, HashSel::of(p.mod),
                            [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + q] = (uint32_t)mod_m(h, p.mod); });
        else
            xxh3_seeds_from(CodeReader{rev_fields(code, k)}, k, n, HashSel::of(p.mod), [&](uint32_t sd, uint64_t h) { ridx[sd * kWave + q] = (uint32_t)mod_m(h, p.mod); });
    };
    for (uint32_t sl = blockIdx.x * waves + wave; sl < p.n_slices; sl += gridDim.x * waves) {
        wave_lds_fence();
        const ReadSlice s = p.slices[sl];
        if (((p.bytes_read && p.bytes_read[s.read]) != 0) != BYTES) continue;   // the other instantiation's read
        const bool multi = (s.part >> 31) != 0;
        const uint32_t part = s.part & 0x7FFFFFFFu;
        const uint32_t r0 = (uint32_t)p.wstart[s.read], r1 = (uint32_t)p.wend[s.read];   // the read's windows
        const uint32_t dread = rank(r0);
        uint32_t q_run = rank(s.w0) - dread;   // order index of the slice's first k-mer inside its read
        uint32_t *row_out = multi ? p.partial + (uint64_t)sl * (C + 2) : p.report + (uint64_t)s.read * (C + 1);
        bool stopped = false;
        VCount<kReadPlanes, NARROW> vc;
        vc.clear();
        V16 R{0, 0};
        if (S > 0 && q_run > 0) {   // the colours of the read's first min(S, q_run) k-mers (they lie in earlier slices)
            const uint32_t t = q_run < S ? q_run : S;   // <= 64: the host cuts reads only then
            uint32_t have = 0;
            for (uint32_t w = r0; have < t && w < r1; w += kWave) {   // (they exist: q_run counts them)
                const uint32_t mine = w + lane;
                const bool flag = mine < r1 && ((p.bitmap[mine >> 5] >> (mine & 31u)) & 1u);
                const uint64_t fm = __ballot(flag);
                const uint32_t q = have + (uint32_t)__popcll(fm & lt_mask);
                if (flag && q < t) hash_to(p.codes[mine], q);
                have += (uint32_t)__popcll(fm);
            }
            wave_lds_fence();
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll, kReadPlanes, false>(p.mat, RS, n, C, S, ridx, (uint32_t)kWave, t, 0u, hist, stopped, vc, R, lane);
            vc.clear();   // only R is wanted: those k-mers are counted by the slices that hold them
            wave_lds_fence();
            for (uint32_t c = lane; c < p.hist_pad; c += kWave) hist[c] = 0;
            // stopped: an earlier slice stops there too and the combine step never reaches this one; nothing to search
        }
        uint32_t wn = s.w0 + lane;
        uint64_t code_next = wn < s.w1 ? p.codes[wn] : 0ull;
        uint32_t word_next = wn < s.w1 ? p.bitmap[wn >> 5] : 0u;
        for (uint32_t w = s.w0; w < s.w1 && !stopped; w += kWave) {
            const uint32_t mine = w + lane;
            const uint64_t code = code_next;
            const bool flag = mine < s.w1 && ((word_next >> (mine & 31u)) & 1u);
            wn = mine + kWave;
            code_next = wn < s.w1 ? p.codes[wn] : 0ull;
            word_next = wn < s.w1 ? p.bitmap[wn >> 5] : 0u;
            const uint64_t fm = __ballot(flag);
            if (!fm) continue;   // (wave-uniform: 64 windows without a first occurrence)
            wave_lds_fence();
            if (flag) hash_to(code, (uint32_t)__popcll(fm & lt_mask));
            const uint32_t cnt = (uint32_t)__popcll(fm);
            wave_lds_fence();
            readid_search_run<LOG_LPR, NARROW, kReadRunUnroll, kReadPlanes, false>(p.mat, RS, n, C, S, ridx, (uint32_t)kWave, cnt, q_run, hist, stopped, vc, R, lane);
            q_run += cnt;
        }
        vc.drain(hist, col_word);
        wave_lds_fence();
        for (uint32_t c = lane; c <= C; c += kWave) { row_out[c] = hist[c]; hist[c] = 0; }
        if (lane == 0) {
            if (multi) row_out[C + 1] = stopped ? 1u : 0u;
            if (part == 0) p.n_kmers[s.read] = rank(r1) - dread;
        }
    }
}

// one workgroup per read that was cut into several slices: the slices' rows in order, up to and including the first that stopped.  A thread
// per colour: its loads do not depend on one another.  (One WAVE per read, five colours a lane: 250 us for the 150 reads of 244 slices that
// 150 Mbases of megabase reads are.)
__global__ __launch_bounds__(kBlock) void k_readid_combine(const ReadCombine *comb, uint32_t n_comb, const uint32_t *partial, uint32_t C, uint32_t *report) {
    __shared__ uint32_t s_last;
    const uint32_t i = blockIdx.x;
    if (i >= n_comb) return;
    const ReadCombine rc = comb[i];
    if (threadIdx.x == 0) s_last = rc.n_slices;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rc.n_slices; j += blockDim.x)   // the first slice that stopped
        if (partial[(uint64_t)(rc.first_slice + j) * (C + 2) + C + 1] != 0) atomicMin(&s_last, j + 1);
    __syncthreads();
    const uint32_t last = s_last;   // slices that count
    for (uint32_t c = threadIdx.x; c <= C; c += blockDim.x) {
        uint32_t acc = 0;
        for (uint32_t j = 0; j < last; ++j) acc += partial[(uint64_t)(rc.first_slice + j) * (C + 2) + c];
        report[(uint64_t)rc.read * (C + 1) + c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers

template <typename KernelT>
static hipError_t launch_readid_one(KernelT kernel, const ReadIdParams &p, int waves_per_block, int grid, hipStream_t stream) {
    const size_t shmem = (size_t)waves_per_block * p.wave_bytes;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves_per_block * kWave), shmem, stream, p);
    return hipGetLastError();
}

// colour-stripe passes (zero_acc / zero_in set): separate instantiations, so that the plain kernels carry none of that logic
template <bool MINI>
static hipError_t launch_readid_packed_striped(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return launch_readid_one(k_readid<0, false, true, MINI, false, true>, p, wpb, grid, stream);
    if (p.rs == 1) return launch_readid_one(k_readid<0, true, false, MINI, false, true>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid<0, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid<1, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid<2, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid<3, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid<4, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid<5, false, false, MINI, false, true>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid<6, false, false, MINI, false, true>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

// the one-u64-per-slot set (p.idx_bits > 0): only built for the six-waves-per-SIMD kernel of whole k-mer indexes
static hipError_t launch_readid_slot4(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return hipErrorInvalidValue;
    if (p.rs == 1) return launch_readid_one(k_readid<0, true, false, false, true, false, false, true>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid<0, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid<1, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid<2, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid<3, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid<4, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid<5, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid<6, false, false, false, true, false, false, true>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}
static hipError_t launch_readid_packed_table(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return hipErrorInvalidValue;
    if (p.rs == 1) return launch_readid_one(k_readid<0, true, false, false, true, false, true>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid<0, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid<1, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid<2, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid<3, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid<4, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid<5, false, false, false, true, false, true>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid<6, false, false, false, true, false, true>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

template <bool MINI, bool DENSE>
static hipError_t launch_readid_packed(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.rs > 128) return launch_readid_one(k_readid<0, false, true, MINI, DENSE>, p, wpb, grid, stream);
    if (p.rs == 1) return launch_readid_one(k_readid<0, true, false, MINI, DENSE>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid<0, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid<1, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid<2, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid<3, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid<4, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid<5, false, false, MINI, DENSE>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid<6, false, false, MINI, DENSE>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

// k <= 32, no lower-case base: one block per p.reads_per_block reads.  The 6-waves-per-SIMD build is used when six
// workgroups' LDS fit the CU's 160 KiB.
hipError_t launch_readid(const ReadIdParams &p, int waves_per_block, hipStream_t stream) {
    const int grid = (int)((p.n_reads + p.reads_per_block - 1) / p.reads_per_block);
    if (p.zero_acc || p.zero_in)
        return p.m_size ? launch_readid_packed_striped<true>(p, waves_per_block, grid, stream) : launch_readid_packed_striped<false>(p,
            waves_per_block, grid, stream);
    // more waves fit a CU than the 96-VGPR build can run (5 per SIMD)
    const bool dense = ((160u * 1024u) / ((size_t)waves_per_block * p.wave_bytes)) * (size_t)waves_per_block > 20 &&
                       ((p.mod.flags >> 8) & 0xFFu) == kHashV08;
    // (the host lays the 8-byte slots out only then)
    if (p.slot4) return dense && !p.m_size ? launch_readid_slot4(p, waves_per_block, grid, stream) : hipErrorInvalidValue;
    if (p.idx_bits) return dense && !p.m_size ? launch_readid_packed_table(p, waves_per_block, grid, stream) : hipErrorInvalidValue;
    if (p.m_size)
        return dense ? launch_readid_packed<true, true>(p, waves_per_block, grid, stream)
                     : launch_readid_packed<true, false>(p, waves_per_block, grid, stream);
    return dense ? launch_readid_packed<false, true>(p, waves_per_block, grid, stream)
                 : launch_readid_packed<false, false>(p, waves_per_block, grid, stream);
}

// byte-string keys: the reads listed in p.redo_list (count on the device), or all of them
hipError_t launch_readid_bytes(const ReadIdParams &p, int wpb, int grid, hipStream_t stream) {
    if (p.zero_acc || p.zero_in) {
        if (p.rs > 128) return launch_readid_one(k_readid_bytes<0, false, true, true>, p, wpb, grid, stream);
        if (p.rs == 1) return launch_readid_one(k_readid_bytes<0, true, false, true>, p, wpb, grid, stream);
        switch (log2u(p.rs / 2)) {
        case 0: return launch_readid_one(k_readid_bytes<0, false, false, true>, p, wpb, grid, stream);
        case 1: return launch_readid_one(k_readid_bytes<1, false, false, true>, p, wpb, grid, stream);
        case 2: return launch_readid_one(k_readid_bytes<2, false, false, true>, p, wpb, grid, stream);
        case 3: return launch_readid_one(k_readid_bytes<3, false, false, true>, p, wpb, grid, stream);
        case 4: return launch_readid_one(k_readid_bytes<4, false, false, true>, p, wpb, grid, stream);
        case 5: return launch_readid_one(k_readid_bytes<5, false, false, true>, p, wpb, grid, stream);
        case 6: return launch_readid_one(k_readid_bytes<6, false, false, true>, p, wpb, grid, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (p.rs > 128) return launch_readid_one(k_readid_bytes<0, false, true>, p, wpb, grid, stream);
    if (p.rs == 1) return launch_readid_one(k_readid_bytes<0, true, false>, p, wpb, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_one(k_readid_bytes<0, false, false>, p, wpb, grid, stream);
    case 1: return launch_readid_one(k_readid_bytes<1, false, false>, p, wpb, grid, stream);
    case 2: return launch_readid_one(k_readid_bytes<2, false, false>, p, wpb, grid, stream);
    case 3: return launch_readid_one(k_readid_bytes<3, false, false>, p, wpb, grid, stream);
    case 4: return launch_readid_one(k_readid_bytes<4, false, false>, p, wpb, grid, stream);
    case 5: return launch_readid_one(k_readid_bytes<5, false, false>, p, wpb, grid, stream);
    case 6: return launch_readid_one(k_readid_bytes<6, false, false>, p, wpb, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

template <typename KernelT>
static hipError_t launch_readid_list_one(KernelT kernel, const ReadIdListParams &p, int grid, hipStream_t stream) {
    const size_t shmem = (size_t)(kBlock / kWave) * p.wave_bytes;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), shmem, stream, p);
    return hipGetLastError();
}

template <bool STRIPED>
static hipError_t launch_readid_list_sel(const ReadIdListParams &p, int grid, hipStream_t stream) {
    if (p.rs > 128) return launch_readid_list_one(k_readid_list<0, false, true, STRIPED>, p, grid, stream);
    if (p.rs == 1) return launch_readid_list_one(k_readid_list<0, true, false, STRIPED>, p, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_list_one(k_readid_list<0, false, false, STRIPED>, p, grid, stream);
    case 1: return launch_readid_list_one(k_readid_list<1, false, false, STRIPED>, p, grid, stream);
    case 2: return launch_readid_list_one(k_readid_list<2, false, false, STRIPED>, p, grid, stream);
    case 3: return launch_readid_list_one(k_readid_list<3, false, false, STRIPED>, p, grid, stream);
    case 4: return launch_readid_list_one(k_readid_list<4, false, false, STRIPED>, p, grid, stream);
    case 5: return launch_readid_list_one(k_readid_list<5, false, false, STRIPED>, p, grid, stream);
    case 6: return launch_readid_list_one(k_readid_list<6, false, false, STRIPED>, p, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_readid_list(const ReadIdListParams &p, int grid, hipStream_t stream) {
    if (p.n_reads == 0) return hipSuccess;
    return (p.zero_acc || p.zero_in) ? launch_readid_list_sel<true>(p, grid, stream) : launch_readid_list_sel<false>(p, grid, stream);
}

template <typename KernelT>
static hipError_t launch_readid_slices_one(KernelT kernel, const ReadIdSliceParams &p, int grid, hipStream_t stream) {
    const size_t shmem = (size_t)(kBlock / kWave) * p.wave_bytes;
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), shmem, stream, p);
    return hipGetLastError();
}
hipError_t launch_readid_slices(const ReadIdSliceParams &p, int grid, hipStream_t stream, bool bytes) {   // rows of at most 1 KiB (rs <= 128), whole indices
    if (p.n_slices == 0) return hipSuccess;
    if (p.rs > 128) return hipErrorInvalidValue;
    if (bytes) {   // the soft-masked reads of the batch
        if (p.rs == 1) return launch_readid_slices_one(k_readid_slices<0, true, true>, p, grid, stream);
        switch (log2u(p.rs / 2)) {
        case 0: return launch_readid_slices_one(k_readid_slices<0, false, true>, p, grid, stream);
        case 1: return launch_readid_slices_one(k_readid_slices<1, false, true>, p, grid, stream);
        case 2: return launch_readid_slices_one(k_readid_slices<2, false, true>, p, grid, stream);
        case 3: return launch_readid_slices_one(k_readid_slices<3, false, true>, p, grid, stream);
        case 4: return launch_readid_slices_one(k_readid_slices<4, false, true>, p, grid, stream);
        case 5: return launch_readid_slices_one(k_readid_slices<5, false, true>, p, grid, stream);
        case 6: return launch_readid_slices_one(k_readid_slices<6, false, true>, p, grid, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (p.rs == 1) return launch_readid_slices_one(k_readid_slices<0, true>, p, grid, stream);
    switch (log2u(p.rs / 2)) {
    case 0: return launch_readid_slices_one(k_readid_slices<0, false>, p, grid, stream);
    case 1: return launch_readid_slices_one(k_readid_slices<1, false>, p, grid, stream);
    case 2: return launch_readid_slices_one(k_readid_slices<2, false>, p, grid, stream);
    case 3: return launch_readid_slices_one(k_readid_slices<3, false>, p, grid, stream);
    case 4: return launch_readid_slices_one(k_readid_slices<4, false>, p, grid, stream);
    case 5: return launch_readid_slices_one(k_readid_slices<5, false>, p, grid, stream);
    case 6: return launch_readid_slices_one(k_readid_slices<6, false>, p, grid, stream);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_readid_combine(const ReadCombine *d_comb, uint32_t n_comb, const uint32_t *d_partial, uint32_t n_colors, uint32_t *d_report,
                                 hipStream_t stream) {
    if (n_comb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_readid_combine, dim3(n_comb), dim3(kBlock), 0, stream, d_comb, n_comb, d_partial, n_colors, d_report);
    return hipGetLastError();
}

// the first use of a kernel loads its translation unit's whole code object (tens of milliseconds for this file's instantiations):
// asking for a kernel's attributes does the same without a launch (cid_warmup)
hipError_t warm_readid() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_readid_check_caps));
}

}  // namespace cid
