// The reference's workload drivers with the hot loops replaced by C-ABI calls:
//   perfect_search::batch_search / batch_search_mf   (src/perfect_search.rs)  -> cid_search_perfect
//   batch_search_pe::batch_search                    (src/batch_search_pe.rs) -> cid_search_count
// (read_id: drivers_readid.cpp) plus the CPU-side tails of the searches (src/reports.rs).  stdout/stderr/file formats follow the reference;
// where the reference iterates a RandomState HashMap, rows come out in ascending colour id.
#include "drivers_common.hpp"

namespace colorid {

bool g_timing = cli_env("COLORID_TIMING") != nullptr;

// ---------------------------------------------------------------------------------------------- several GPUs
cid_group *g_group = nullptr;
std::vector<cid_index *> g_replicas;   // one handle per rank: replicas of the index, or (g_striped) its colour stripes
bool g_striped = false;
void set_group(cid_group *group, const std::vector<cid_index *> &replicas) { g_group = group; g_replicas = replicas; g_striped = false; }
void set_stripes(cid_group *group, const std::vector<cid_index *> &stripes) { g_group = group; g_replicas = stripes; g_striped = true; }

// the hot-path calls: one GPU, a group with replicas (query sharded), or a group with colour stripes (index sharded)
static int hot_search_count(cid_ctx *ctx, const Bigsi &b, const uint8_t *kmers, const uint32_t *freq, size_t n, uint64_t *hits, uint64_t *nu,
                            uint64_t *sf, uint32_t *uc) {
    if (g_striped) return cid_group_stripes_search_count(g_group, g_replicas.data(), kmers, freq, n, hits, nu, sf, uc);
    return g_group ? cid_group_search_count(g_group, g_replicas.data(), kmers, freq, n, hits, nu, sf, uc)
                   : cid_search_count(ctx, b.index, kmers, freq, n, hits, nu, sf, uc);
}
static int hot_search_count_set(cid_ctx *ctx, const Bigsi &b, const cid_kmerset *ks, uint64_t *hits, uint64_t *nu, uint64_t *sf, uint32_t *uc) {
    if (g_striped) return cid_group_stripes_search_count_set(g_group, g_replicas.data(), ks, hits, nu, sf, uc);
    return g_group ? cid_group_search_count_set(g_group, g_replicas.data(), ks, hits, nu, sf, uc) : cid_search_count_set(ctx, b.index, ks, hits, nu, sf, uc);
}
static int hot_search_perfect(cid_ctx *ctx, const Bigsi &b, const uint8_t *kmers, size_t n, uint32_t *words, int *missing) {
    if (g_striped) return cid_group_stripes_search_perfect(g_group, g_replicas.data(), kmers, n, words, missing);
    return g_group ? cid_group_search_perfect(g_group, g_replicas.data(), kmers, n, words, missing) : cid_search_perfect(ctx, b.index, kmers, n, words, missing);
}
static int hot_search_perfect_set(cid_ctx *ctx, const Bigsi &b, const cid_kmerset *ks, uint32_t *words, int *missing) {
    if (g_striped) return cid_group_stripes_search_perfect_set(g_group, g_replicas.data(), ks, words, missing);
    return g_group ? cid_group_search_perfect_set(g_group, g_replicas.data(), ks, words, missing) : cid_search_perfect_set(ctx, b.index, ks, words, missing);
}

// ---------------------------------------------------------------------------------------------- reports.rs

double false_prob(double m, double k, double n) { return std::pow(1.0 - std::pow(M_E, -((k * (n + 0.5)) / (m - 1.0))), k); }

// reports.rs:8-48: hits / n_ref_kmers > cov -> query, K, accession, cov, mean, mode, n_unique
// `modes`: the per-colour mode of the unique-hit k-mer frequencies when the device computed it (cid_search_count_set_report);
// NULL: derived here from the per-k-mer unique colours and multiplicities (reports.rs:65-77; ties -> smallest value)
static void generate_report(const std::string &query, const Bigsi &b, const std::vector<uint64_t> &hits,
                            const std::vector<uint64_t> &n_unique, const std::vector<uint64_t> &sum_freq,
                            const std::vector<uint32_t> &unique_colour, const uint32_t *counts, size_t n_kmers, double cov,
                            const std::vector<uint64_t> *modes = nullptr) {
    const size_t C = b.colors.size();
    std::vector<std::map<uint32_t, uint64_t>> occ(modes ? 0 : C);
    if (!modes)
        for (size_t j = 0; j < n_kmers; ++j)
            if (unique_colour[j] != CID_NOT_UNIQUE) occ[unique_colour[j]][counts[j]] += 1;
    for (size_t c = 0; c < C; ++c) {
        if (!hits[c]) continue;
        double mean = 0.0;
        uint64_t modus = 0, specific = 0;
        if (n_unique[c]) {
            mean = (double)sum_freq[c] / (double)n_unique[c];
            if (modes) modus = (*modes)[c];
            else {
                uint64_t bestc = 0;
                for (auto &kv : occ[c])
                    if (kv.second > bestc) { bestc = kv.second; modus = kv.first; }
            }
            specific = n_unique[c];
        }
        const double genome_cov = (double)hits[c] / (double)b.n_ref_kmers[c];
        if (genome_cov > cov)
            printf("%s\t%zu\t%s\t%.2f\t%.2f\t%llu\t%llu\n", query.c_str(), n_kmers, b.colors[c].c_str(), genome_cov, mean,
                   (unsigned long long)modus, (unsigned long long)specific);
    }
}

static void generate_report_gene(const std::string &query, const Bigsi &b, const std::vector<uint64_t> &hits, size_t num_kmers,
                                 double cov) {  // reports.rs:50-62
    for (size_t c = 0; c < b.colors.size(); ++c) {
        if (!hits[c]) continue;
        const double gene_match = (double)hits[c] / (double)num_kmers;
        if (gene_match >= cov) printf("%s\t%s\t%zu\t%.3f\n", query.c_str(), b.colors[c].c_str(), num_kmers, gene_match);
    }
}

// ---------------------------------------------------------------------------------------------- perfect_search.rs

// ---------------------------------------------------------------------------------------------- GPU k-mer counting
//  the k-mer map is built and kept on the device (2-bit codes for k <= 32, byte strings beyond);
// COLORID_HOST_KMERS=1 forces the host map.

bool gpu_counting_enabled(uint64_t k) { return k <= 128 && !cli_env("COLORID_HOST_KMERS"); }
static bool gpu_counting(const Bigsi &b) { return gpu_counting_enabled(b.k_size); }


// host-only helper of the CPU tests: what the record pipeline hands to the GPU calls, one line per read — id, then its masked mates
void debug_records(const std::string &f1, const std::string *f2, uint8_t q) {
    stream_fastq_records(f1, f2, q, true, [&](ReadBatch &&piece) {
        for (size_t r = 0; r < piece.size(); ++r) {
            fputs(piece.id(r), stdout);
            for (uint64_t sq = piece.read_seq0[r]; sq < piece.read_seq0[r + 1]; ++sq) {
                fputc('\t', stdout);
                fwrite(piece.bases.data() + piece.seq_off[sq], 1, (size_t)(piece.seq_off[sq + 1] - piece.seq_off[sq]), stdout);
            }
            fputc('\n', stdout);
        }
    });
}

cid_kmerset *count_fasta_gpu(cid_ctx *ctx, uint64_t k, const std::vector<std::string> &seqs, const cid_index *target) {
    cid_kmerset *ks = nullptr;
    CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
    if (target) CID_TRY(cid_kmerset_set_target_index(ks, target));
    SeqBatch sb;
    for (const std::string &s : seqs) sb.push(s);
    CID_TRY(cid_kmerset_add_seqs(ks, sb.bases.data(), sb.off.data(), sb.n(), 0));
    CID_TRY(cid_kmerset_finalize(ks, nullptr));
    return ks;
}

// fastq(.gz) SE or PE (kmer.rs:461-510 / :581-655) in batches of 256 MB of bases: add(batch) returns the add_seqs code;
// false = a batch held lower-case bases (CID_ERR_UNSUPPORTED): count the file on the host
template <typename Add>
static bool stream_fastq_batches(const std::string &f1, const std::string *f2, uint8_t q, Add &&add) {
    SeqBatch sb;
    bool ok = true;
    auto flush = [&]() {
        if (sb.n() == 0 || !ok) return;
        const int rc = add(sb);
        if (rc == CID_ERR_UNSUPPORTED) ok = false;
        else if (rc != CID_OK) die("cid_kmerset_add_seqs: %s", cid_last_error());
        sb.clear();
    };
    stream_fastq_records(f1, f2, q, false, [&](ReadBatch &&piece) {
        if (!ok) return;   // (the rest of the file is still read: the caller falls back to the host map, which reads it again)
        const uint64_t b0 = sb.bases.size();
        sb.bases.insert(sb.bases.end(), piece.bases.begin(), piece.bases.end());
        for (size_t i = 1; i < piece.seq_off.size(); ++i) sb.off.push_back(b0 + piece.seq_off[i]);
        if (sb.bases.size() >= (256u << 20)) flush();
    });
    flush();
    return ok;
}

// Block-gzip input through the device FASTQ front end: the members go up compressed (the reader's spare threads inflate their share),
// records are cut, masked and packed on the device and their k-mers go straight into the set — no text comes back.  false: the input
// is not for this path (CID_ERR_UNSUPPORTED: lower-case bases, reads longer than a segment) and `ks` holds a partial count.
static bool count_bgzf_on_device(cid_ctx *ctx, cid_kmerset *ks, const std::vector<std::string> &fq, uint8_t q) {
    const size_t n_files = fq.size();
    cid_fastq *fr = nullptr;
    CID_TRY(cid_fastq_create(ctx, (int)n_files, q, &fr));
    const size_t target = read_id_mt_pe::device_fastq_stretch_bytes(0);
    std::unique_ptr<BgzfMemberReader> rd[2];
    for (size_t i = 0; i < n_files; ++i)
        rd[i] = BgzfMemberReader::open(fq[i], target, read_id_mt_pe::device_fastq_host_share(), read_id_mt_pe::device_fastq_host_threads(n_files));
    BgzfStretch st[2][2];   // per file two stretches in turn: the one pushed last stays untouched while its text is still on the bus
    size_t turn[2] = {0, 0}, pending[2] = {0, 0};
    bool more[2] = {true, n_files == 2};
    double ms_read = 0, ms_push = 0, ms_count = 0;
    auto push_next = [&](size_t i) {
        if (!more[i]) return;
        const auto tr = Clock::now();
        BgzfStretch &sx = st[i][turn[i]++ & 1];
        const bool got = rd[i]->next(sx);
        ms_read += ms_since(tr);
        if (!got) { more[i] = false; return; }
        const auto tp = Clock::now();
        const bool host_part = sx.host_text_bytes > 0;
        // (only the device's members travel: they come first in the stretch; the rest was inflated here)
        const size_t dev_bytes = sx.device_members ? (size_t)sx.off[sx.device_members - 1] + sx.len[sx.device_members - 1] : 0;
        CID_TRY(cid_fastq_push_bgzf(fr, (int)i, sx.bytes.data(), dev_bytes, sx.off.data(), sx.len.data(), sx.text_len.data(), sx.device_members,
                                    sx.bytes.pinned ? CID_FASTQ_KEEP : 0));   // (waited for in the next push of this file; two buffers in turn)
        // ALWAYS the second push, also when it is empty: a classify step takes two pushes per file = exactly one stretch, and the wait
        // for the copy of the stretch before (CID_FASTQ_KEEP) happens here, before the reader gets that stretch's buffer back
        CID_TRY(cid_fastq_push_text(fr, (int)i, host_part ? sx.host_text.p : nullptr, host_part ? sx.host_text_bytes : 0,
                                    (sx.last ? CID_FASTQ_LAST : 0) | (host_part && sx.host_text.pinned ? CID_FASTQ_KEEP : 0)));
        if (sx.last) more[i] = false;
        ms_push += ms_since(tp);
        ++pending[i];
    };
    for (size_t i = 0; i < n_files; ++i) push_next(i);
    bool ok = true;
    while (ok && (pending[0] || pending[1])) {
        for (size_t i = 0; i < n_files; ++i) push_next(i);   // the stretch after this one: inflated while this one is counted
        uint64_t n = 0;
        const auto tc = Clock::now();
        const int rc = cid_fastq_count_kmers(fr, ks, 2, &n);
        ms_count += ms_since(tc);
        for (size_t i = 0; i < n_files; ++i) if (pending[i]) --pending[i];
        if (rc == CID_ERR_UNSUPPORTED) ok = false;
        else if (rc != CID_OK) die("%s", cid_last_error());
    }
    cid_fastq_destroy(fr);
    if (g_timing) fprintf(stderr, "timing: device front end: waiting for the file reader %.0f ms, push (H2D of the members) %.0f ms, counting %.0f ms\n", ms_read, ms_push, ms_count);
    return ok;
}

// nullptr = the file holds lower-case bases: count it on the host
cid_kmerset *count_fastq_gpu(cid_ctx *ctx, uint64_t k, const std::string &f1, const std::string *f2, uint8_t q, const cid_index *target) {
    cid_kmerset *ks = nullptr;
    CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
    if (target) CID_TRY(cid_kmerset_set_target_index(ks, target));
    {
        std::vector<std::string> fq{f1};
        if (f2) fq.push_back(*f2);
        if (k <= 32 && read_id_mt_pe::device_fastq_wanted(fq, fq.size())) {
            const auto t_dev = Clock::now();
            if (count_bgzf_on_device(ctx, ks, fq, q)) {
                CID_TRY(cid_kmerset_finalize(ks, nullptr));
                if (g_timing) fprintf(stderr, "timing: query k-mers counted through the device front end in %.0f ms\n", ms_since(t_dev));
                return ks;
            }
            cid_kmerset_destroy(ks);   // (a partial count) — the host reads the files again
            CID_TRY(cid_kmerset_create(ctx, (uint32_t)k, &ks));
            if (target) CID_TRY(cid_kmerset_set_target_index(ks, target));
        }
    }
    const bool ok = stream_fastq_batches(f1, f2, q, [&](const SeqBatch &sb) { return cid_kmerset_add_seqs(ks, sb.bases.data(), sb.off.data(), sb.n(), 1); });
    if (!ok) { cid_kmerset_destroy(ks); return nullptr; }
    CID_TRY(cid_kmerset_finalize(ks, nullptr));
    return ks;
}

int64_t auto_cutoff_gpu(cid_kmerset *ks) {
    size_t nb = 0;
    CID_TRY(cid_kmerset_count_histogram(ks, nullptr, nullptr, 0, &nb));
    std::vector<uint32_t> mult(nb);
    std::vector<uint64_t> cnt(nb);
    CID_TRY(cid_kmerset_count_histogram(ks, mult.data(), cnt.data(), nb, &nb));
    std::map<uint64_t, uint64_t> hm;
    for (size_t i = 0; i < nb; ++i) hm[mult[i]] = cnt[i];
    uint64_t n = 0;
    CID_TRY(cid_kmerset_size(ks, &n));
    return auto_cutoff_from_histogram(hm, n);
}

// The query's k-mer map on the GPU side of the search drivers: one cid_kmerset (one GPU, or a striped index whose ranks all need
// the whole set), or — a replicated group, k <= 32 — a cid_group_kmerset counted over all ranks, each of which then searches its
// own range of the set (COLORID_ONE_GPU_KMERS=1 keeps the counting on rank 0).
struct GpuSet {
    cid_kmerset *one = nullptr;
    cid_group_kmerset *many = nullptr;
    explicit operator bool() const { return one || many; }
    uint64_t size() const {
        uint64_t n = 0;
        if (many) CID_TRY(cid_group_kmerset_size(many, &n)); else CID_TRY(cid_kmerset_size(one, &n));
        return n;
    }
    void clean(uint64_t t) { if (many) CID_TRY(cid_group_kmerset_clean(many, t)); else CID_TRY(cid_kmerset_clean(one, t)); }
    void counts(uint32_t *out) const { if (many) CID_TRY(cid_group_kmerset_download(many, nullptr, out)); else CID_TRY(cid_kmerset_download(one, nullptr, out)); }
    int64_t auto_cutoff() const {
        if (!many) return auto_cutoff_gpu(one);
        size_t nb = 0;
        CID_TRY(cid_group_kmerset_count_histogram(many, nullptr, nullptr, 0, &nb));
        std::vector<uint32_t> mult(nb);
        std::vector<uint64_t> cnt(nb);
        CID_TRY(cid_group_kmerset_count_histogram(many, mult.data(), cnt.data(), nb, &nb));
        std::map<uint64_t, uint64_t> hm;
        for (size_t i = 0; i < nb; ++i) hm[mult[i]] = cnt[i];
        return auto_cutoff_from_histogram(hm, size());
    }
    void destroy() { if (many) cid_group_kmerset_destroy(many); if (one) cid_kmerset_destroy(one); many = nullptr; one = nullptr; }
};
static bool count_over_group(uint64_t k) { return g_group && !g_striped && k <= 32 && !cli_env("COLORID_ONE_GPU_KMERS"); }

static GpuSet count_fasta_set(cid_ctx *ctx, uint64_t k, const std::vector<std::string> &seqs, const cid_index *target) {
    GpuSet gs;
    if (!count_over_group(k)) { gs.one = count_fasta_gpu(ctx, k, seqs, target); return gs; }
    CID_TRY(cid_group_kmerset_create(g_group, (uint32_t)k, &gs.many));
    SeqBatch sb;
    for (const std::string &s : seqs) sb.push(s);
    CID_TRY(cid_group_kmerset_add_seqs(gs.many, sb.bases.data(), sb.off.data(), sb.n(), 0));
    CID_TRY(cid_group_kmerset_finalize(gs.many, nullptr));
    return gs;
}
static GpuSet count_fastq_set(cid_ctx *ctx, uint64_t k, const std::string &f1, const std::string *f2, uint8_t q, const cid_index *target) {
    GpuSet gs;
    if (!count_over_group(k)) { gs.one = count_fastq_gpu(ctx, k, f1, f2, q, target); return gs; }
    CID_TRY(cid_group_kmerset_create(g_group, (uint32_t)k, &gs.many));
    const bool ok = stream_fastq_batches(f1, f2, q, [&](const SeqBatch &sb) { return cid_group_kmerset_add_seqs(gs.many, sb.bases.data(), sb.off.data(), sb.n(), 1); });
    if (!ok) { gs.destroy(); return gs; }
    CID_TRY(cid_group_kmerset_finalize(gs.many, nullptr));
    return gs;
}
static int hot_search_count_gpuset(cid_ctx *ctx, const Bigsi &b, const GpuSet &gs, uint64_t *hits, uint64_t *nu, uint64_t *sf, uint32_t *uc) {
    if (gs.many) return cid_group_search_count_parts(g_group, g_replicas.data(), gs.many, hits, nu, sf, uc);
    return hot_search_count_set(ctx, b, gs.one, hits, nu, sf, uc);
}
static int hot_search_perfect_gpuset(cid_ctx *ctx, const Bigsi &b, const GpuSet &gs, uint32_t *words, int *missing) {
    if (gs.many) return cid_group_search_perfect_parts(g_group, g_replicas.data(), gs.many, words, missing);
    return hot_search_perfect_set(ctx, b, gs.one, words, missing);
}

static void print_perfect(const Bigsi &b, const std::string &label, size_t n_kmers, const std::vector<uint32_t> &words, int missing);

static void perfect_one(cid_ctx *ctx, const Bigsi &b, const std::string &label, const KmerMap &km) {
    const uint32_t w32 = (uint32_t)((b.colors.size() + 31) / 32);
    std::vector<uint32_t> words(w32);
    int missing = 0;
    CID_TRY(hot_search_perfect(ctx, b, km.keys(), km.size(), words.data(), &missing));
    print_perfect(b, label, km.size(), words, missing);
}

static void print_perfect(const Bigsi &b, const std::string &label, size_t n_kmers, const std::vector<uint32_t> &words, int missing) {
    if (missing) {
        fprintf(stderr, "No perfect hits!\n");
        return;
    }
    size_t n_hits = 0;
    for (size_t c = 0; c < b.colors.size(); ++c) n_hits += (words[c / 32] >> (c % 32)) & 1u;
    fprintf(stderr, "%zu hits\n", n_hits);
    for (size_t c = 0; c < b.colors.size(); ++c)
        if ((words[c / 32] >> (c % 32)) & 1u) printf("%s\t%s\t%zu\t1.00\n", label.c_str(), b.colors[c].c_str(), n_kmers);
}

void perfect_search::batch_search(cid_ctx *ctx, const std::vector<std::string> &files, const Bigsi &b) {
    for (const std::string &file : files) {
        fprintf(stderr, "Counting k-mers, this may take a while!\n");
        if (gpu_counting(b)) {
            GpuSet ks = count_fasta_set(ctx, b.k_size, read_fasta(file), b.index);
            const uint64_t n = ks.size();
            fprintf(stderr, "%llu kmers in query\n", (unsigned long long)n);
            if (n == 0) {
                fprintf(stderr, "Warning! no kmers in query; maybe your kmer length is larger than your query length?\n");
            } else {
                std::vector<uint32_t> words((b.colors.size() + 31) / 32);
                int missing = 0;
                CID_TRY(hot_search_perfect_gpuset(ctx, b, ks, words.data(), &missing));
                print_perfect(b, file, n, words, missing);
            }
            ks.destroy();
            continue;
        }
        KmerMap km((uint32_t)b.k_size);
        kmerize_vector(read_fasta(file), 1, km);
        fprintf(stderr, "%zu kmers in query\n", km.size());
        if (km.size() == 0) {
            fprintf(stderr, "Warning! no kmers in query; maybe your kmer length is larger than your query length?\n");
            continue;
        }
        perfect_one(ctx, b, file, km);
    }
}

void perfect_search::batch_search_mf(cid_ctx *ctx, const std::vector<std::string> &files, const Bigsi &b) {
    for (const std::string &file : files) {
        std::vector<std::string> labels, seqs;
        read_fasta_mf(file, labels, seqs);
        for (size_t i = 0; i < labels.size(); ++i) {
            if (i >= seqs.size()) die("index out of bounds: the len is %zu but the index is %zu", seqs.size(), i);  // sequences[i]
            KmerMap km((uint32_t)b.k_size);
            if (!kmerize_string(seqs[i], km)) {
                printf("Warning! no kmers in query '%s'; maybe your kmer length is larger than your query length?\n", labels[i].c_str());
                continue;
            }
            fprintf(stderr, "%zu kmers in query\n", km.size());
            perfect_one(ctx, b, labels[i], km);
        }
    }
}

// ---------------------------------------------------------------------------------------------- batch_search_pe.rs

void batch_search_pe::batch_search(cid_ctx *ctx, const std::vector<std::string> &files1, const std::vector<std::string> &files2,
                                   const Bigsi &b, int64_t filter, double cov, bool gene_search, uint8_t qual_offset) {
    const size_t C = b.colors.size();
    for (size_t i = 0; i < files1.size(); ++i) {
        const std::string &file1 = files1[i];
        const bool gz = file1.size() >= 2 && file1.compare(file1.size() - 2, 2, "gz") == 0;
        if (gz && !files2.empty() && i >= files2.size()) die("index out of bounds: the len is %zu but the index is %zu", files2.size(), i);
        if (gz && files2.empty()) fprintf(stderr, "%s\nCounting k-mers, this may take a while!\n", file1.c_str());
        else if (gz) fprintf(stderr, "Paired end: %s %s\nCounting k-mers, this may take a while!\n", file1.c_str(), files2[i].c_str());
        else fprintf(stderr, "%s\nCounting k-mers, this may take a while!\n", file1.c_str());  // anything else is FASTA (:106-123)
        // which cutoff applies (batch_search_pe.rs:34-39 fastq, :111-121 FASTA): -1 = auto_cutoff
        const bool fasta_gene = !gz && gene_search;
        const bool use_auto = !fasta_gene && filter < 0;
        if (!gz && !gene_search && filter < 0) fprintf(stderr, "no gene search\n");

        std::vector<uint64_t> hits(C), n_unique(C), sum_freq(C), modes;
        bool have_modes = false;
        std::vector<uint32_t> uc, counts;
        size_t n_kmers = 0;
        GpuSet ks;
        if (gpu_counting(b)) {
            ks = gz ? count_fastq_set(ctx, b.k_size, file1, files2.empty() ? nullptr : &files2[i], qual_offset, b.index)
                    : count_fasta_set(ctx, b.k_size, read_fasta(file1), b.index);
        }
        if (ks) {  // the k-mer map lives on the device(s)
            uint64_t t = fasta_gene ? 0 : (uint64_t)(filter < 0 ? 0 : filter);
            if (use_auto) { const int64_t a = ks.auto_cutoff(); if (a < 0) die("auto_cutoff: histogram too short"); t = (uint64_t)a; }
            ks.clean(t);
            n_kmers = ks.size();
            fprintf(stderr, "%zu k-mers in query\n", n_kmers);
            const auto t0 = Clock::now();
            if (!gene_search && !g_group) {   // the whole report on the device: nothing per k-mer comes back
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_search_count_set_report(ctx, b.index, ks.one, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else if (!gene_search && g_striped && ks.one) {   // the same over colour stripes: the report is finished on rank 0's device
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_group_stripes_search_count_set_report(g_group, g_replicas.data(), ks.one, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else if (!gene_search && ks.many) {   // the same over the ranks' parts: the (colour, multiplicity) histograms add up
                modes.assign(C, 0);
                have_modes = true;
                CID_TRY(cid_group_search_count_parts_report(g_group, g_replicas.data(), ks.many, hits.data(), n_unique.data(), sum_freq.data(), modes.data()));
            } else {
                if (!gene_search) { uc.resize(n_kmers); counts.resize(n_kmers); }
                CID_TRY(hot_search_count_gpuset(ctx, b, ks, hits.data(), gene_search ? nullptr : n_unique.data(),
                                                gene_search ? nullptr : sum_freq.data(), gene_search ? nullptr : uc.data()));
                if (!gene_search) ks.counts(counts.data());
            }
            if (gz) fprintf(stderr, "Search: %ld sec\n", secs_since(t0));
            ks.destroy();
        } else {  // host k-mer map (k > 32, lower-case fastq, or COLORID_HOST_KMERS)
            fprintf(stderr, "k-mer map on the host\n");
            KmerMap km((uint32_t)b.k_size);
            if (gz && files2.empty()) kmers_from_fq_qual(file1, qual_offset, km);
            else if (gz) kmers_fq_pe_qual(file1, files2[i], qual_offset, km);
            else kmerize_vector(read_fasta(file1), 1, km);
            uint64_t t = fasta_gene ? 0 : (uint64_t)(filter < 0 ? 0 : filter);
            if (use_auto) { const int64_t a = km.auto_cutoff(); if (a < 0) die("auto_cutoff: histogram too short"); t = (uint64_t)a; }
            km.clean(t);
            n_kmers = km.size();
            fprintf(stderr, "%zu k-mers in query\n", n_kmers);
            const auto t0 = Clock::now();
            if (!gene_search) uc.resize(n_kmers);
            CID_TRY(hot_search_count(ctx, b, km.keys(), km.counts().data(), km.size(), hits.data(),
                                     gene_search ? nullptr : n_unique.data(), gene_search ? nullptr : sum_freq.data(),
                                     gene_search ? nullptr : uc.data()));
            if (gz) fprintf(stderr, "Search: %ld sec\n", secs_since(t0));
            counts = km.counts();
        }
        if (!gene_search) generate_report(file1, b, hits, n_unique, sum_freq, uc, counts.data(), n_kmers, cov, have_modes ? &modes : nullptr);
        else generate_report_gene(file1, b, hits, n_kmers, cov);
    }
}

}  // namespace colorid
