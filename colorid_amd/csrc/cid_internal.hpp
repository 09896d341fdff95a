// Internals shared by the translation units of libcolorid_hip.so (not part of the ABI).
#pragma once
#include <vector>
#include "cid_kernels.hpp"

struct cid_ctx;
struct cid_index;
struct cid_kmerset;

namespace cid {

int fail(int code, const char *fmt, ...);  // records the thread's last error message, returns `code`
int ctx_device(const cid_ctx *c);
hipStream_t ctx_stream(const cid_ctx *c);
int ctx_order_bits(const cid_ctx *c);
int ctx_n_cu(const cid_ctx *c);
// the ctx's own stream, its second (copy) stream and two untimed events for calls that overlap an upload with kernels; free between calls
hipStream_t ctx_own_stream(const cid_ctx *c);
hipStream_t ctx_copy_stream(const cid_ctx *c);
hipEvent_t ctx_event(const cid_ctx *c, int i);   // cid_ctx_tune "order_bits"
// Device scratch that survives the call: hipMalloc of a GiB-sized block costs tens of milliseconds here (the driver clears
// it), so blocks go back to a per-ctx cache instead of hipFree and the next batch takes them again.  All users run on the
// ctx stream, which orders a block's last kernel before its next owner's first.  Objects holding such blocks
// (cid_kmerset, sparse read_id results) must be destroyed before their ctx.
int ctx_alloc(cid_ctx *c, size_t bytes, void **out);
void ctx_free(cid_ctx *c, void *p);
uint32_t kmerset_k(const cid_kmerset *ks);
cid_ctx *kmerset_ctx(const cid_kmerset *ks);
uint32_t index_k(const cid_index *ix);
uint32_t index_rs(const cid_index *ix);
ModMagic index_mod(const cid_index *ix);
uint32_t index_n_colors(const cid_index *ix);
uint32_t index_n_hash(const cid_index *ix);
uint32_t index_m_size(const cid_index *ix);
const uint64_t *index_matrix(const cid_index *ix);

// Bloom insert of 2-bit codes already on the device into one colour (build.rs:62-66 with the k-mer map on the GPU)
int index_insert_codes(cid_index *ix, const uint64_t *d_codes, size_t n, uint32_t k, uint32_t colour);

// read_id over colour stripes (ReadIdParams::zero_acc ...): which pass this launch is, and where its results go; all zero = a whole index
struct StripePass {
    uint32_t *zero_acc = nullptr;
    const uint32_t *zero_in = nullptr;
    const uint64_t *zero_start = nullptr;   // device, [n_reads]
    uint32_t colour_base = 0, report_width = 0, write_nohits = 0;
    bool on() const { return zero_acc || zero_in; }
};

// read_id for reads that do not fit (or badly fit) a wave's LDS (cid_readlong.hip): per-read k-mer sets by workgroup-wide LDS hash
// tables, the ordered search by slices of a read.  Everything — bases AND offsets — is on the device; the work lists are made there
// (round 6).  d_route == NULL: every read; else only reads with d_route[r] == 1 — the others get status 2 and nothing else is written
// for them.  clear_wide: zero the whole report first when rows are wider than 128 words (those kernels count in place).  h_seq_off /
// h_read_seq0: the same offsets on the host when the caller has them (NULL: downloaded IF a read needs the sorting path).
// d_route_stats (4 words, long_route_launch's) come down into route_stats with the plan's totals.  Waits for the stream twice: for the
// lists' sizes (64 bytes) before its kernels, and at its end.
int readid_long(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *d_seq_off, const uint64_t *d_read_seq0,
                size_t n_reads, uint32_t stride_d, uint32_t start_sample, const uint8_t *d_route, bool clear_wide, uint32_t *d_report,
                uint32_t *d_n_kmers, uint8_t *d_status, const StripePass &sp = StripePass(), const uint64_t *h_seq_off = nullptr,
                const uint64_t *h_read_seq0 = nullptr, const uint32_t *d_route_stats = nullptr, uint32_t *route_stats = nullptr);
// d_route[r] = 1: read r has at least long_from bases (the long-read path's), 0: the LDS kernels', 3: beyond cap_bytes / cap_win (what a
// device-pointer caller stated as maxima; long_beyond_launch gives those status 3 and an empty row).  d_stats[4]: long reads, LDS-kernel
// reads, the longest of the latter in bases and in windows.  Asynchronous.
int long_route_launch(cid_ctx *c, const uint64_t *d_seq_off, const uint64_t *d_read_seq0, size_t n_reads, uint32_t k, uint32_t stride_d, uint64_t long_from,
                      uint64_t cap_bytes, uint64_t cap_win, uint8_t *d_route, uint32_t *d_stats);
int long_beyond_launch(cid_ctx *c, const uint8_t *d_route, size_t n_reads, uint32_t report_width, uint32_t *d_report, uint32_t *d_n_kmers, uint8_t *d_status);
// the same by a global radix sort of every window of the batch (cid_kmerset_cold.hip; round 1's path): byte-string keys — k > 32, or the
// reads with a lower-case base that readid_long hands back (merge_status: only the routed reads' statuses are written).  Offsets on the host.
int readid_long_sorted(cid_ctx *c, const cid_index *ix, const uint8_t *d_bases, const uint64_t *seq_off, const uint64_t *read_seq0,
                       size_t n_reads, uint32_t stride_d, uint32_t start_sample, const uint8_t *route, bool clear_wide, uint32_t *d_report,
                       uint32_t *d_n_kmers, uint8_t *d_status, const StripePass &sp = StripePass(), bool merge_status = false);

// a5 / a4 on k-mers that are already on the device as 2-bit codes (k <= 32); outputs go to HOST buffers
int search_count_codes(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, const uint32_t *d_counts, size_t n, uint32_t k,
                       uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
int search_perfect_codes(cid_ctx *c, const cid_index *ix, const uint64_t *d_codes, size_t n, uint32_t k, uint32_t *and_words_le,
                         int *any_row_missing);

// rocPRIM behind plain calls (cid_kmerset_cold.hip — the one translation unit that includes it; its code object of some thousand kernels is
// loaded when the first of these is called): stable LSD radix sorts on bits [b0, b1), a run-length count of sorted keys.  Asynchronous on `st`.
int cold_sort_keys_u64(cid_ctx *c, hipStream_t st, const uint64_t *in, uint64_t *out, size_t n, unsigned b0, unsigned b1);
int cold_sort_pairs_u64_u32(cid_ctx *c, hipStream_t st, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned b0,
                            unsigned b1);
int cold_sort_pairs_u32_u64(cid_ctx *c, hipStream_t st, const uint32_t *kin, uint32_t *kout, const uint64_t *vin, uint64_t *vout, size_t n, unsigned b0,
                            unsigned b1);
int cold_run_length_u64(cid_ctx *c, hipStream_t st, const uint64_t *sorted, size_t n, uint64_t *uniq, uint32_t *runs, uint64_t *d_n_runs);

// a finalized set's device arrays (codes ascending unless reordered; counts u32) and the ctx they live in
int kmerset_view(const cid_kmerset *ks, cid_ctx **ctx, const uint64_t **codes, const uint32_t **counts, uint64_t *n, uint32_t *k);
// replace a finalized 2-bit-code set's contents by the merge (sort by code, add counts of equal codes) of `total` pairs on its device
int kmerset_assign_merged(cid_kmerset *ks, const uint64_t *d_codes_in, const uint32_t *d_counts_in, size_t total);
// d_modes[c] = the most frequent multiplicity among the k-mers whose unique colour is c (ties -> the smallest; 0 = none); asynchronous
int unique_freq_modes(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, uint32_t C, uint64_t *d_modes);
// the same in two asynchronous steps (cid_reports.hip): after _begin d_modes holds the modes over the multiplicities below the table's
// width and w->ovf_count[0] (device) the number of k-mers beyond it; _finish counts those in when there are any and frees w's arrays
struct ModeWork {
    uint64_t *ovf = nullptr;
    unsigned long long *best = nullptr, *ovf_count = nullptr;
    uint32_t C = 0;
};
int unique_freq_modes_begin(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, uint32_t C, uint64_t *d_modes, ModeWork *w);
int unique_freq_modes_finish(cid_ctx *c, ModeWork *w, unsigned long long n_ovf, uint64_t *d_modes);
// sorted (colour << 32 | multiplicity) keys and their k-mer counts over the k-mers with a unique colour (host vectors; synchronous)
int unique_freq_hist(cid_ctx *c, const uint32_t *d_uc, const uint32_t *d_freq, uint64_t n, std::vector<uint64_t> &keys, std::vector<uint32_t> &counts);
int kmerset_view_ascii(const cid_kmerset *ks, cid_ctx **ctx, const uint8_t **ascii, const uint32_t **counts, uint64_t *n, uint32_t *k);   // k > 32 sets

// the same three for sets of byte-string k-mers (k > 32): d_ascii = n x k bytes on the device
int index_insert_ascii(cid_index *ix, const uint8_t *d_ascii, size_t n, uint32_t k, uint32_t colour);
int search_count_ascii(cid_ctx *c, const cid_index *ix, const uint8_t *d_ascii, const uint32_t *d_counts, size_t n, uint32_t k,
                       uint64_t *hits, uint64_t *n_unique, uint64_t *sum_unique_freq, uint32_t *unique_colour);
int search_perfect_ascii(cid_ctx *c, const cid_index *ix, const uint8_t *d_ascii, size_t n, uint32_t k, uint32_t *and_words_le, int *any_row_missing);

// the non-zero rows of [row_begin, row_begin + n_rows) as .bxi row records in a host buffer (cid_kmerset.hip)
int index_get_records(cid_ctx *c, const cid_index *ix, uint64_t row_begin, uint64_t n_rows, uint8_t *records, uint64_t *n_records);

// dense report rows (device) -> per-row (colour, count) lists, ascending colour; outputs are ctx_alloc'ed for the caller (return them with ctx_free)
int compact_report(cid_ctx *c, const uint32_t *d_report, uint32_t width, uint64_t n_rows, uint64_t **d_row_start, uint32_t **d_colours,
                   uint32_t **d_counts, uint64_t *n_entries);

// block-gzip members on the device (cid_inflate.hip): member i = in[in_off, +in_len) (header, DEFLATE data, CRC-32, ISIZE), its text to
// out[out_off, +out_len); status[i] = 0 or the reason it is corrupt.  Asynchronous on `stream`.
struct BgzfMember { uint32_t in_off, in_len, out_off, out_len; };
// d_scratch: bgzf_inflate_scratch_bytes(n_members) bytes the launch may use until it has run (match tokens of the wave-parallel kernel, its
// retry list), or NULL (one lane per member)
size_t bgzf_inflate_scratch_bytes(uint32_t n_members);
hipError_t bgzf_inflate_launch(cid_ctx *c, hipStream_t stream, const uint8_t *d_in, const BgzfMember *d_mem, uint32_t n_members, uint8_t *d_out,
                               uint32_t *d_st, void *d_scratch);
const char *bgzf_status_text(uint32_t st);

// load a translation unit's code object ahead of its first kernel launch (cid_warmup)
hipError_t warm_readid();
hipError_t warm_readlong();
hipError_t warm_cold();
hipError_t warm_search();
hipError_t warm_kmerset();
hipError_t warm_reports();
hipError_t warm_inflate();
hipError_t warm_fastq();
hipError_t ctx_side_streams(cid_ctx *c, hipStream_t out[4]);   // created on first use; any thread

}  // namespace cid
