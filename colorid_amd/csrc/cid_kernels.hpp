// Kernel parameter blocks and launchers shared by the kernel files (cid_search / cid_readid / cid_index .hip) and the ABI (cid_api_*.hip, cid_kmerset.hip).
#pragma once
#include "cid_device.hpp"

namespace cid {

constexpr int kBlock = 256;  // 4 waves; every wave works on its own 64-k-mer tiles
#ifndef CID_READID_ALIAS
#define CID_READID_ALIAS 1   // k_readid: the search-phase histogram shares the hash table's LDS region (0: separate regions, A/B builds)
#endif
constexpr uint32_t kNoKey = 0xFFFFFFFFu;   // a k-mer set built for an index: the sort key of a window without a k-mer (cid_partition.hpp, cid_windows.hpp)
constexpr int kPlanes = 8;   // bit-sliced per-colour counters per lane: drained every 255 k-mers

struct SearchParams {
    const uint64_t *mat;   // dense index, row r at mat + r*rs
    uint32_t rs;           // row stride in u64 words (1, or a power of two 2..128)
    uint32_t w64;          // words that carry colours = ceil(n_colors/64)
    uint32_t n_colors;
    uint32_t n_hash;
    uint32_t k;
    uint32_t c_pad;        // n_colors rounded up to even (LDS counter arrays)
    uint32_t wave_bytes;   // LDS bytes per wave: k-mer image + hash rows
    uint32_t want_unique;
    uint32_t tiles_per_block;  // block b owns tiles [b*tpb, (b+1)*tpb); grid = ceil(n_tiles/tpb)
    ModMagic mod;
    const uint8_t *kmers;  // n_kmers * k ASCII bytes, 16-byte aligned (or nullptr when codes != nullptr)
    const uint64_t *codes; // alternative input: canonical upper-case k-mers as 2-bit codes, base 0 most significant
    const uint32_t *freq;  // or nullptr
    uint64_t n_kmers;
    // a5 outputs (accumulated with atomics; caller zeroes)
    uint64_t *hits;
    uint64_t *n_unique;
    uint64_t *sum_unique_freq;
    uint32_t *unique_colour;
    // a4 outputs (caller presets and_words to all-ones, *missing to 0)
    uint64_t *and_words;
    int *missing;
    // colour-striped indices (this index holds colours [colour_base, colour_base + n_colors) of a wider one): the
    // per-k-mer facts that need every stripe are accumulated in caller-provided arrays instead of being decided here
    uint32_t colour_base;
    uint32_t *fact;        // [n_kmers] packed stripe fact, see stripe_fact_merge            (a5; nullptr = not striped)
    uint32_t *zero_acc;    // [n_kmers] &= bit s set iff row s is all-zero in this stripe  (a4; nullptr = not striped)
    // persistent, XCD-aware scheduling of k_search_count (nullptr = one contiguous tile range per block)
    uint32_t *queues;      // 8 work-queue heads, 32 words apart, zeroed before the launch
    int persist_grid;      // resident blocks: n_cu x blocks per CU
    uint32_t mixed;        // 32-byte rows: fetch each k-mer's last row through the scalar cache (gather_and_mixed32)
    uint32_t unroll;       // rows of >= 64 bytes: sub-passes whose loads are issued together (1 or 2)
};

struct InsertParams {
    uint64_t *mat;
    uint32_t rs;
    uint32_t n_hash;
    uint32_t k;
    uint32_t n_colors;
    uint32_t tiles_per_block;
    uint32_t colour;  // used when colour_of_kmer == nullptr
    uint32_t m_size;  // > 0: a minimizer (.mxi) index — the Bloom key is find_minimizer(kmer, m_size) (build.rs:455-459)
    ModMagic mod;
    const uint8_t *kmers;       // ASCII, or nullptr when codes != nullptr
    const uint64_t *codes;      // 2-bit canonical codes (k <= 32)
    const uint32_t *colour_of_kmer;
    uint64_t n_kmers;
};

struct ReadIdParams {
    const uint64_t *mat;
    uint32_t rs, w64, n_colors, n_hash, k;
    ModMagic mod;
    const uint8_t *bases;       // concatenated, quality-masked reads
    const uint64_t *seq_off;    // [n_seqs+1]
    const uint64_t *read_seq0;  // [n_reads+1]
    uint64_t n_reads;
    uint32_t stride_d, start_sample;
    uint32_t m_size;            // > 0: minimizer index — the per-read set holds minimizers (kmer.rs:363-394), hashed with length m_size
    uint32_t bases_cap;         // LDS bytes per wave for one read(-pair)'s bases (multiple of 16)
    uint32_t win_cap;           // max windows (= max distinct k-mers) of one read(-pair)
    uint32_t hist_pad;          // n_colors+1 rounded up to a multiple of 4
    uint32_t table_slots;       // power of two >= 1.5 x win_cap: the packed path's per-read k-mer set
    uint32_t idx_bits;          // > 0: one u64 per slot, code << idx_bits | first window index (2k + idx_bits <= 63; k_readid<..., PACKED>)
    uint32_t slot4;             // > 0: one u32 per slot, the earliest position of a window holding the slot's k-mer (k_readid<..., SLOT4>)
    uint32_t stage_bytes;       // k_readid: LDS bytes per wave for the next read's raw bases (multiple of 16, <= 1024; 0 = no read-ahead)
    uint32_t wave_bytes;        // LDS bytes per wave
    uint32_t reads_per_block;
    uint32_t *report;           // [n_reads][n_colors+1]
    uint32_t *n_kmers;          // [n_reads]
    uint8_t *status;            // [n_reads]
    const uint8_t *skip;        // NULL, or [n_reads]: non-zero = this read belongs to the sort-based path (k_readid_list), leave it alone
    uint32_t *redo_count;       // k_readid appends the reads it cannot pack (lower-case bases) to redo_list; k_readid_bytes
    uint32_t *redo_list;        //   works through that list (redo_list == NULL: through all reads, k > 32)
    // colour stripes (this index = colours [colour_base, colour_base + n_colors) of a wider one; all three NULL/0 otherwise).
    // "A row is absent" (read_id_mt_pe.rs:81-89, :126-128) means: all-zero in EVERY stripe, so the rule runs in two passes:
    //   zero pass  (zero_acc != NULL): every distinct k-mer of every read is looked up, nothing is counted;
    //              zero_acc[zero_start[read] + q] &= bit s set iff row s of the read's q-th k-mer is all-zero in this stripe
    //              (q = rank in the read's first-occurrence order — the same in the LDS kernels and in k_readid_list, so a read
    //              may take a different kernel in different stripes; zero_start = any prefix leaving >= windows(read) words per read);
    //   count pass (zero_in  != NULL): the ordered search, with "absent" read from the accumulated masks instead of this stripe's rows;
    //              report rows are report_width wide, this stripe's colours land at [colour_base ..), the no-hits entry at
    //              [report_width - 1] is written by the stripe with write_nohits set.
    uint32_t *zero_acc;
    const uint32_t *zero_in;
    const uint64_t *zero_start;   // [n_reads]
    uint32_t colour_base, report_width, write_nohits;
};

struct ReadIdListParams {  // k_readid_list: per-read distinct k-mers already in first-occurrence order
    const uint64_t *mat;
    uint32_t rs, w64, n_colors, n_hash, k;   // k = length of the listed keys (k-mers, or minimizers for a .mxi index)
    ModMagic mod;
    const uint64_t *list_codes;   // canonical 2-bit codes, base 0 most significant; with `bases`: key location << 1 | reverse-complement
    const uint8_t *bases;         // non-NULL: keys are byte strings inside `bases` (k > 32 or lower-case bases), k bytes each
    uint32_t upper;               // byte keys are upper-cased before hashing (.mxi minimizers)
    const uint64_t *list_start;   // [n_reads+1] offsets into list_codes
    uint64_t n_reads;
    uint32_t start_sample;
    uint32_t hist_pad, wave_bytes;
    uint32_t *report;
    uint32_t *n_kmers;
    const uint8_t *status;        // set by the caller: 1 = too_short, 2 = not this kernel's read (k_readid handles it)
    // colour stripes, as in ReadIdParams (all NULL/0: a whole index)
    uint32_t *zero_acc;
    const uint32_t *zero_in;
    const uint64_t *zero_start;
    uint32_t colour_base, report_width, write_nohits;
};

// The long-read path's in-order search (k_readid_slices): a read's ordered list of distinct k-mers is cut into slices of consecutive
// windows, one wave per slice; reads cut into several slices leave partial rows that k_readid_combine adds up in slice order.
struct ReadSlice {
    uint32_t read;
    uint32_t w0, w1;   // the slice's windows [w0, w1), numbered over the whole batch
    uint32_t part;     // index of the slice inside its read | 1 << 31 when the read has more than one slice
};
struct ReadCombine { uint32_t read, first_slice, n_slices, pad; };
struct ReadIdSliceParams {
    const uint64_t *mat;
    uint32_t rs, w64, n_colors, n_hash, k;   // k = length of the listed keys (k-mers, or minimizers for a .mxi index)
    ModMagic mod;
    const uint64_t *codes;        // every window's canonical 2-bit code (windows numbered over the batch; a read's start at a multiple of 32)
    const uint64_t *wstart, *wend;   // [n_reads]: a read's windows [wstart, wend)
    const uint32_t *bitmap;       // bit w: window w is the first occurrence of its k-mer in its read
    const uint32_t *word_prefix;  // exclusive prefix of the bitmap words' popcounts: rank(w) = word_prefix[w >> 5] + popc(bits below w)
    const ReadSlice *slices;
    uint32_t n_slices;
    uint32_t start_sample;
    uint32_t hist_pad, wave_bytes;
    uint32_t *report;
    uint32_t *n_kmers;
    uint32_t *partial;            // [n_slices][n_colors + 2]: counts | no-hits | stopped — only rows of multi-slice reads are written
    // soft-masked reads (k_long_bytes): bytes_read[r] != 0 — read r's entries in `codes` are (byte offset in bases << 1 | reverse complement)
    // of byte-string k-mers; the plain kernel leaves those reads to the BYTES instantiation and the other way round.  NULL: no such reads
    const uint8_t *bytes_read;
    const uint8_t *bases;
};
hipError_t launch_readid_slices(const ReadIdSliceParams &p, int grid, hipStream_t stream, bool bytes = false);
hipError_t launch_readid_combine(const ReadCombine *d_comb, uint32_t n_comb, const uint32_t *d_partial, uint32_t n_colors, uint32_t *d_report,
                                 hipStream_t stream);

size_t search_smem_bytes(const SearchParams &p);
hipError_t launch_readid_list(const ReadIdListParams &p, int grid, hipStream_t stream);
hipError_t launch_readid(const ReadIdParams &p, int waves_per_block, hipStream_t stream);
hipError_t launch_readid_bytes(const ReadIdParams &p, int waves_per_block, int grid, hipStream_t stream);
hipError_t launch_readid_check_caps(const ReadIdParams &p, uint8_t *skip, hipStream_t stream);
hipError_t launch_unique_finalize(const uint32_t *fact, const uint32_t *freq, uint64_t n_kmers,
                                  uint32_t n_colors_total, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour,
                                  hipStream_t stream);
int grid_for(uint64_t n_kmers, uint32_t tiles_per_block);
hipError_t launch_search_count(const SearchParams &p, hipStream_t stream);
int search_count_blocks_per_cu(const SearchParams &p);
hipError_t launch_search_perfect(const SearchParams &p, hipStream_t stream);
hipError_t launch_put_rows(uint64_t *mat, uint32_t rs, const uint64_t *d_row_ids, const uint32_t *d_words, uint32_t w32,
                           uint64_t n_rows, hipStream_t stream);
hipError_t launch_put_records(uint64_t *mat, uint32_t rs, const uint32_t *d_records, uint32_t w32_rec, uint32_t w_off, uint32_t w32_take,
                              uint64_t n_records, uint64_t bloom_size, uint32_t n_colors, uint32_t *d_err, hipStream_t stream);
hipError_t launch_get_rows(const uint64_t *mat, uint32_t rs, const uint64_t *d_row_ids, uint32_t *d_words, uint32_t w32,
                           uint64_t n_rows, hipStream_t stream);
hipError_t launch_insert_kmers(const InsertParams &p, hipStream_t stream);

}  // namespace cid
