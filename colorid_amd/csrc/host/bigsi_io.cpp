// .bxi (bincode 1.x of BigsyMapNew, src/bigsi.rs:19-27 / SURVEY.md App. A) <-> device-resident index,
// and the index builder (src/build.rs:15-130) with the Bloom inserts done on the GPU.
#include <algorithm>
#include <cstring>
#include <future>
#include <thread>

#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "colorid_host.hpp"

namespace colorid {

#define CID_TRY(expr)                                                  \
    do {                                                               \
        if ((expr) != CID_OK) die("%s: %s", #expr, cid_last_error()); \
    } while (0)

namespace {

struct BufReader {  // big sequential reads; the file is parsed once, front to back
    FILE *f;
    std::vector<uint8_t> buf;
    size_t pos = 0, end = 0;
    explicit BufReader(const std::string &path) : f(fopen(path.c_str(), "rb")), buf(1u << 20) {
        if (!f) die("Can't open index!: %s", path.c_str());
    }
    ~BufReader() { fclose(f); }
    void need(size_t n) {
        if (end - pos >= n) return;
        memmove(buf.data(), buf.data() + pos, end - pos);
        end -= pos;
        pos = 0;
        if (n > buf.size()) buf.resize(n);
        while (end < n) {
            size_t got = fread(buf.data() + end, 1, buf.size() - end, f);
            if (got == 0) die("can't deserialize: unexpected end of file");
            end += got;
        }
    }
    // bulk: what is buffered first, then straight from the file — large reads with several pread() threads (a copy out of the
    // page cache runs at one core's memory speed, ~10 GB/s; the upload that follows takes 50 GB/s)
    void read_exact(uint8_t *dst, size_t n) {
        if (n == 0) return;
        const size_t take = std::min(n, end - pos);
        memcpy(dst, buf.data() + pos, take);
        pos += take;
        size_t got = take;
        static const int n_threads = [] { const char *e = cli_env("COLORID_IO_THREADS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
        if (n - got >= (32u << 20) && n_threads > 1) {
            const off_t at = ftello(f);   // the FILE's logical position == the next byte this reader has not seen
            if (at >= 0) {
                const size_t rest = n - got, per = (rest + (size_t)n_threads - 1) / (size_t)n_threads;
                std::vector<int> bad((size_t)n_threads, 0);
                std::vector<std::thread> th;
                auto work = [&](int t) {
                    size_t o = (size_t)t * per;
                    const size_t stop = std::min(rest, o + per);
                    while (o < stop) {
                        const ssize_t g = pread(fileno(f), dst + got + o, stop - o, at + (off_t)o);
                        if (g <= 0) { bad[(size_t)t] = 1; return; }
                        o += (size_t)g;
                    }
                };
                for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
                work(0);
                for (auto &x : th) x.join();
                for (int b : bad) if (b) die("can't deserialize: unexpected end of file");
                if (fseeko(f, at + (off_t)rest, SEEK_SET) != 0) die("can't deserialize: seek failed");
                return;
            }
        }
        while (got < n) {
            const size_t g = fread(dst + got, 1, n - got, f);
            if (g == 0) die("can't deserialize: unexpected end of file");
            got += g;
        }
    }
    uint64_t tell() { return (uint64_t)ftello(f) - (end - pos); }   // the file offset of the next byte this reader hands out
    void seek(uint64_t at) {
        if (fseeko(f, (off_t)at, SEEK_SET) != 0) die("can't deserialize: seek failed");
        pos = end = 0;
    }
    uint64_t u64() {
        need(8);
        uint64_t v;
        memcpy(&v, buf.data() + pos, 8);  // little-endian host
        pos += 8;
        return v;
    }
    std::string str() {
        const uint64_t n = u64();
        if (n > (1u << 30)) die("can't deserialize: string of %llu bytes", (unsigned long long)n);
        need(n);
        std::string s(reinterpret_cast<const char *>(buf.data() + pos), n);
        pos += n;
        return s;
    }
};

void w64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }

// The index file mapped into memory before the GPU context exists (bigsi_read_ahead, called first thing by the subcommands that load
// an index): threads touch its pages — out of the page cache, or off the disk — while the runtime starts up (0.1-0.3 s), and the
// loader then hands the row records to cid_index_put_records straight from the mapping: no copy into a buffer of ours at all
// (fread's copy of a 2.8 GB file was most of the 0.15 s the load took).  COLORID_INDEX_MMAP=0 keeps the buffered reader.
struct MappedIndex {
    const uint8_t *p = nullptr;
    size_t n = 0;
};
std::mutex g_map_mu;
std::map<std::string, MappedIndex> g_mapped;

}  // namespace

void bigsi_read_ahead(const std::string &path) {
    if (const char *e = cli_env("COLORID_INDEX_MMAP")) if (atoi(e) == 0) return;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return;   // (the loader reports it)
    struct stat st;
    // small files: nothing to gain; files beyond 64 GiB (a striped index of several GPUs' worth) stay with the sequential buffered
    // reader: four threads touching such a file ahead of the loader would fight it for the disk and the page cache
    if (fstat(fd, &st) != 0 || st.st_size < (off_t)(64 << 20) || st.st_size > (off_t)(64ll << 30)) { close(fd); return; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return;
    (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
    MappedIndex mi;
    mi.p = static_cast<const uint8_t *>(m);
    mi.n = (size_t)st.st_size;
    {
        std::lock_guard<std::mutex> lk(g_map_mu);
        g_mapped[path] = mi;
    }
    const int n_threads = 4;
    for (int t = 0; t < n_threads; ++t)
        std::thread([mi, t, n_threads] {   // page tables filled in ahead of the upload (detached: the process outlives them or exits)
            const size_t per = (mi.n + (size_t)n_threads - 1) / (size_t)n_threads, lo = (size_t)t * per, hi = std::min(mi.n, lo + per);
            for (size_t o = lo; o < hi; o += 4096) (void)*(volatile const uint8_t *)(mi.p + o);
        }).detach();
}

namespace {
MappedIndex mapped_index(const std::string &path) {
    std::lock_guard<std::mutex> lk(g_map_mu);
    auto it = g_mapped.find(path);
    return it == g_mapped.end() ? MappedIndex() : it->second;
}
}  // namespace

Bigsi read_bigsi(cid_ctx *ctx, const std::string &path, int hash_variant, bool meta_only, cid_group *group, std::vector<cid_index *> *stripes) {
    BufReader r(path);
    Bigsi b;
    b.bloom_size = r.u64();
    b.num_hash = r.u64();
    b.k_size = r.u64();
    const bool mini = path.size() >= 4 && path.compare(path.size() - 4, 4, ".mxi") == 0;   // the suffix selects the struct (main.rs:723)
    if (mini) b.m_size = r.u64();                                                            // BigsyMapMiniNew.m_size
    const uint64_t nc = r.u64();
    if (nc == 0 || nc > (1u << 24)) die("can't deserialize: %llu colours", (unsigned long long)nc);
    b.colors.assign(nc, std::string());
    for (uint64_t i = 0; i < nc; ++i) {
        const uint64_t id = r.u64();
        std::string name = r.str();
        if (id >= nc) die("can't deserialize: colour id %llu of %llu", (unsigned long long)id, (unsigned long long)nc);
        b.colors[id] = std::move(name);
    }
    const uint32_t w32 = (uint32_t)((nc + 31) / 32);
    const uint64_t n_rows = r.u64();
    if (!meta_only && stripes) {   // one colour stripe per rank of the group (an index larger than one GPU's HBM)
        if (mini) die("Error: an index with minimizers (.mxi) cannot be striped over GPUs");
        int n_ranks = 0;
        CID_TRY(cid_group_size(group, &n_ranks));
        stripes->assign((size_t)n_ranks, nullptr);
        CID_TRY(cid_group_stripes_create(group, b.bloom_size, (uint32_t)b.num_hash, (uint32_t)b.k_size, (uint32_t)nc, hash_variant, stripes->data()));
    } else if (!meta_only) {
        CID_TRY(cid_index_create(ctx, b.bloom_size, (uint32_t)b.num_hash, (uint32_t)b.k_size, (uint32_t)nc, hash_variant, &b.index));
        if (mini) CID_TRY(cid_index_set_minimizer(b.index, (uint32_t)b.m_size));
    }
    // The row records go to the device as they sit in the file — { u64 row ; u64 W32 ; W32 x u32 ; u64 nbits } each — and are
    // parsed and checked there (cid_index_put_records); a second thread reads the next chunk while this one is uploaded.
    const size_t rec = 24 + 4ull * w32;
    if (n_rows > (~0ull) / rec) die("can't deserialize: %llu rows", (unsigned long long)n_rows);
    if (meta_only) {   // `info`: no device; the records are still checked, as deserialising them would
        std::vector<uint8_t> chunk;
        for (uint64_t left = n_rows; left;) {
            const size_t nr = (size_t)std::min<uint64_t>(left, (16u << 20) / rec + 1);
            chunk.resize(nr * rec);
            r.read_exact(chunk.data(), nr * rec);
            for (size_t i = 0; i < nr; ++i) {
                uint64_t nw, nbits;
                memcpy(&nw, chunk.data() + i * rec + 8, 8);
                memcpy(&nbits, chunk.data() + i * rec + 16 + 4ull * w32, 8);
                if (nw != w32) die("can't deserialize: row with %llu words, expected %u", (unsigned long long)nw, w32);
                if (nbits != nc) die("can't deserialize: row of %llu bits, expected %llu", (unsigned long long)nbits, (unsigned long long)nc);
            }
            left -= nr;
        }
    } else if (const MappedIndex mi = mapped_index(path); mi.p) {   // straight from the mapping (bigsi_read_ahead)
        const uint64_t at = r.tell();
        if (at > mi.n || n_rows * rec > mi.n - at) die("can't deserialize: unexpected end of file");
        const size_t chunk_recs = std::max<size_t>(1, (256u << 20) / rec);
        for (uint64_t done = 0; done < n_rows;) {
            const size_t nr = (size_t)std::min<uint64_t>(n_rows - done, chunk_recs);
            const uint8_t *src = mi.p + at + done * rec;
            const int rc = stripes ? cid_group_stripes_put_records(group, stripes->data(), src, nr) : cid_index_put_records(b.index, src, nr);
            if (rc != CID_OK) die("can't deserialize: %s", cid_last_error());
            done += nr;
        }
        r.seek(at + n_rows * rec);
    } else {
        const size_t chunk_recs = std::max<size_t>(1, (128u << 20) / rec);
        std::vector<uint8_t> bufs[2];
        size_t have[2] = {0, 0};
        uint64_t left = n_rows;
        auto fill = [&](int which) {
            const size_t nr = (size_t)std::min<uint64_t>(left, chunk_recs);
            bufs[which].resize(nr * rec);
            r.read_exact(bufs[which].data(), nr * rec);
            have[which] = nr;
            left -= nr;
        };
        fill(0);
        int cur = 0;
        while (have[cur]) {
            std::thread prefetch([&, cur] { fill(cur ^ 1); });
            const int rc = stripes ? cid_group_stripes_put_records(group, stripes->data(), bufs[cur].data(), have[cur])
                                   : cid_index_put_records(b.index, bufs[cur].data(), have[cur]);
            prefetch.join();
            if (rc != CID_OK) die("can't deserialize: %s", cid_last_error());
            cur ^= 1;
        }
    }
    b.n_ref_kmers.assign(nc, 0);
    std::map<std::string, uint64_t> by_name;
    for (uint64_t c = 0; c < nc; ++c) by_name[b.colors[c]] = c;
    const uint64_t n_ref = r.u64();
    for (uint64_t i = 0; i < n_ref; ++i) {
        std::string name = r.str();
        const uint64_t v = r.u64();
        auto it = by_name.find(name);
        if (it != by_name.end()) b.n_ref_kmers[it->second] = v;
    }
    if (!meta_only && stripes) {
        for (cid_index *ix : *stripes) CID_TRY(cid_index_finalize(ix));
    } else if (!meta_only) CID_TRY(cid_index_finalize(b.index));
    return b;
}

void save_bigsi(const std::string &path, const Bigsi &b) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) die("problems preparing serialized data for writing: %s", path.c_str());
    const uint64_t nc = b.colors.size();
    const uint32_t w32 = (uint32_t)((nc + 31) / 32);
    w64(f, b.bloom_size); w64(f, b.num_hash); w64(f, b.k_size);
    if (b.m_size) w64(f, b.m_size);
    w64(f, nc);
    for (uint64_t c = 0; c < nc; ++c) { w64(f, c); w64(f, b.colors[c].size()); fwrite(b.colors[c].data(), 1, b.colors[c].size(), f); }
    // rows come back from the device in ascending order; all-zero rows are dropped (build.rs:123-127)
    const long count_pos = ftell(f);
    w64(f, 0);
    uint64_t n_rows = 0;
    const size_t rec = 24 + 4ull * w32;
    const uint64_t chunk = std::max<uint64_t>(1, (128u << 20) / rec);   // records are formatted on the device; the host writes bytes
    std::vector<uint8_t> records(chunk * rec);
    for (uint64_t r0 = 0; r0 < b.bloom_size; r0 += chunk) {
        const uint64_t nr = std::min<uint64_t>(chunk, b.bloom_size - r0);
        uint64_t got = 0;
        CID_TRY(cid_index_get_records(b.index, r0, nr, records.data(), &got));
        if (got && fwrite(records.data(), rec, got, f) != got) die("problems writing %s", path.c_str());
        n_rows += got;
    }
    w64(f, nc);
    for (uint64_t c = 0; c < nc; ++c) { w64(f, b.colors[c].size()); fwrite(b.colors[c].data(), 1, b.colors[c].size(), f); w64(f, b.n_ref_kmers[c]); }
    fseek(f, count_pos, SEEK_SET);
    w64(f, n_rows);
    fclose(f);
}

// tab_to_map (build.rs:15-31): accession \t file [\t file2]; later lines overwrite earlier ones
std::map<std::string, std::vector<std::string>> tab_to_map(const std::string &ref_tsv) {
    std::map<std::string, std::vector<std::string>> refs;  // std::map iterates sorted == accessions.sort() (build.rs:105)
    LineReader r(ref_tsv);
    std::string line;
    while (r.next(line)) {
        std::vector<std::string> v;
        size_t p = 0;
        while (true) {
            size_t e = line.find('\t', p);
            v.push_back(line.substr(p, e == std::string::npos ? std::string::npos : e - p));
            if (e == std::string::npos) break;
            p = e + 1;
        }
        if (v.size() < 2) die("reference file line without a tab: '%s'", line.c_str());
        refs[v[0]] = v.size() == 2 ? std::vector<std::string>{v[1]} : std::vector<std::string>{v[1], v[2]};
    }
    return refs;
}

size_t hashcheck(cid_ctx *ctx, Bigsi &b, const std::string &ref_tsv, uint8_t quality, const char *const *variant_names, std::vector<double> &worst) {
    if (b.m_size) die("hashcheck works on k-mer indices (.bxi); a minimizer index shares its hash with the .bxi built by the same binary");
    const auto refs = tab_to_map(ref_tsv);
    const size_t C = b.colors.size();
    std::vector<uint64_t> hits(C);
    size_t n_checked = 0;
    printf("variant\taccession\tkmers\tpresent\tfraction\n");
    for (auto &kv : refs) {
        const auto it = std::find(b.colors.begin(), b.colors.end(), kv.first);
        if (it == b.colors.end()) { fprintf(stderr, "%s is not an accession of the index: skipped\n", kv.first.c_str()); continue; }
        const uint32_t colour = (uint32_t)(it - b.colors.begin());
        const std::vector<std::string> &v = kv.second;
        const bool is_gz = v.size() == 2 || (v[0].size() >= 2 && v[0].compare(v[0].size() - 2, 2, "gz") == 0);
        // the accession's k-mers as build counts them (reads: with the automatic cutoff, the build default)
        cid_kmerset *ks = nullptr;
        KmerMap km((uint32_t)b.k_size);
        uint64_t nk = 0;
        if (gpu_counting_enabled(b.k_size))
            ks = is_gz ? count_fastq_gpu(ctx, b.k_size, v[0], v.size() == 2 ? &v[1] : nullptr, quality) : count_fasta_gpu(ctx, b.k_size, read_fasta(v[0]));
        if (ks) {
            if (is_gz) { const int64_t t = auto_cutoff_gpu(ks); CID_TRY(cid_kmerset_clean(ks, (uint64_t)(t < 0 ? 0 : t))); }
            CID_TRY(cid_kmerset_size(ks, &nk));
        } else {
            if (v.size() == 2) kmers_fq_pe_qual(v[0], v[1], quality, km);
            else if (is_gz) kmers_from_fq_qual(v[0], quality, km);
            else kmerize_vector(read_fasta(v[0]), 1, km);
            if (is_gz) { const int64_t t = km.auto_cutoff(); km.clean((uint64_t)(t < 0 ? 0 : t)); }
            nk = km.size();
        }
        if (nk == 0) { fprintf(stderr, "%s has no k-mers: skipped\n", kv.first.c_str()); if (ks) cid_kmerset_destroy(ks); continue; }
        for (int hv = 0; hv < CID_HASH_VARIANTS; ++hv) {
            CID_TRY(cid_index_set_hash_variant(b.index, hv));
            if (ks) CID_TRY(cid_search_count_set(ctx, b.index, ks, hits.data(), nullptr, nullptr, nullptr));
            else CID_TRY(cid_search_count(ctx, b.index, km.keys(), nullptr, km.size(), hits.data(), nullptr, nullptr, nullptr));
            const double frac = (double)hits[colour] / (double)nk;
            printf("%s\t%s\t%llu\t%llu\t%.4f\n", variant_names[hv], kv.first.c_str(), (unsigned long long)nk, (unsigned long long)hits[colour], frac);
            if (frac < worst[hv]) worst[hv] = frac;
        }
        if (ks) cid_kmerset_destroy(ks);
        ++n_checked;
    }
    CID_TRY(cid_index_set_hash_variant(b.index, CID_HASH_XXH3_V08));
    return n_checked;
}

Bigsi build_single(cid_ctx *ctx, const std::string &ref_tsv, uint64_t bloom, uint64_t hashes, uint64_t k, uint8_t quality,
                   int64_t cutoff, int hash_variant, uint64_t m_size) {
    const auto refs = tab_to_map(ref_tsv);
    Bigsi b;
    b.bloom_size = bloom; b.num_hash = hashes; b.k_size = k; b.m_size = m_size;
    for (auto &kv : refs) b.colors.push_back(kv.first);
    b.n_ref_kmers.assign(b.colors.size(), 0);
    CID_TRY(cid_index_create(ctx, bloom, (uint32_t)hashes, (uint32_t)k, (uint32_t)b.colors.size(), hash_variant, &b.index));
    if (m_size) CID_TRY(cid_index_set_minimizer(b.index, (uint32_t)m_size));   // the inserts below then key on find_minimizer(kmer, m)
    uint32_t colour = 0, counter = 1;
    // the NEXT accession's input is read (FASTA) or starts inflating (fastq.gz) on a helper thread while the GPU counts and inserts
    // this one's k-mers
    auto gz_of = [](const std::vector<std::string> &v) { return v.size() == 2 || (v[0].size() >= 2 && v[0].compare(v[0].size() - 2, 2, "gz") == 0); };
    std::future<std::vector<std::string>> ahead;
    auto look_ahead = [&](std::map<std::string, std::vector<std::string>>::const_iterator it) {
        if (it == refs.end() || !gpu_counting_enabled(k)) return;
        const std::vector<std::string> &v = it->second;
        if (gz_of(v) && k <= 32 && read_id_mt_pe::device_fastq_wanted(v, v.size())) {   // block gzip: the members go up compressed (count_fastq_gpu)
            for (const std::string &f : v)
                BgzfMemberReader::prefetch(f, read_id_mt_pe::device_fastq_stretch_bytes(0), read_id_mt_pe::device_fastq_host_share(),
                                           read_id_mt_pe::device_fastq_host_threads(v.size()));
        } else if (gz_of(v)) for (const std::string &f : v) LineReader::prefetch(f);
        else { const std::string path = v[0]; ahead = std::async(std::launch::async, [path] { return read_fasta(path); }); }
    };
    look_ahead(refs.begin());
    for (auto it = refs.begin(); it != refs.end(); ++it) {
        const auto &kv = *it;
        fprintf(stderr, "Adding %s to index (%u/%zu)\n", kv.first.c_str(), counter++, refs.size());
        const std::vector<std::string> &v = kv.second;
        const bool is_gz = gz_of(v);
        cid_kmerset *ks = nullptr;
        std::vector<std::string> fasta_now;
        if (gpu_counting_enabled(k) && !is_gz) fasta_now = ahead.get();
        look_ahead(std::next(it));
        if (gpu_counting_enabled(k))
            ks = is_gz ? count_fastq_gpu(ctx, k, v[0], v.size() == 2 ? &v[1] : nullptr, quality) : count_fasta_gpu(ctx, k, fasta_now);
        if (ks) {  // the accession's k-mer map never leaves HBM: count, clean, Bloom-insert
            if (is_gz) {
                uint64_t t = (uint64_t)(cutoff < 0 ? 0 : cutoff);
                if (cutoff == -1) { const int64_t a = auto_cutoff_gpu(ks); if (a < 0) die("auto_cutoff: histogram too short"); t = (uint64_t)a; }
                CID_TRY(cid_kmerset_clean(ks, t));
            } else if (cutoff != -1) {
                CID_TRY(cid_kmerset_clean(ks, (uint64_t)cutoff));   // build.rs:86-91: FASTA is only cleaned with an explicit -f
            }
            uint64_t nk = 0;
            CID_TRY(cid_kmerset_size(ks, &nk));
            b.n_ref_kmers[colour] = (m_size && is_gz) ? 0 : nk;   // build_single_mini records it for FASTA accessions only (Q13)
            CID_TRY(cid_index_insert_kmerset(b.index, ks, colour));
            cid_kmerset_destroy(ks);
        } else {
            KmerMap km((uint32_t)k);
            auto clean_reads = [&]() {
                if (cutoff == -1) { const int64_t t = km.auto_cutoff(); if (t < 0) die("auto_cutoff: histogram too short"); km.clean((uint64_t)t); }
                else km.clean((uint64_t)cutoff);
            };
            if (v.size() == 2) { kmers_fq_pe_qual(v[0], v[1], quality, km); clean_reads(); }
            else if (is_gz) { kmers_from_fq_qual(v[0], quality, km); clean_reads(); }
            else {
                kmerize_vector(read_fasta(v[0]), 1, km);
                if (cutoff != -1) km.clean((uint64_t)cutoff);
            }
            b.n_ref_kmers[colour] = (m_size && is_gz) ? 0 : km.size();
            CID_TRY(cid_index_insert_kmers(b.index, km.keys(), colour, km.size()));
        }
        ++colour;
    }
    LineReader::drop_prefetched();
    BgzfMemberReader::drop_prefetched();
    return b;
}

}  // namespace colorid
