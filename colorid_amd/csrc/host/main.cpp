// `colorid` command line: the reference's drop-in surface for the query path (src/main.rs) —
//   search  (src/main.rs:127-217, :555-628)   read_id (:241-328, :704-868)   batch_id (:329-418, :869-888: read_id over a sample sheet)
//   build   (:31-126, :466-554; needed to produce .bxi files)   info (:218-240, :630-703)
// Same flag letters, defaults, stdout/stderr/file formats.  Extra flags: --device N, --gpus N | --devices a,b,.. (search, read_id:
// the query is sharded over the GPUs), --hash xxh3_v08|xxh3_v07.  Extra command:
// hashcheck (which hash variant was an index built with).
// Minimizer indices (.mxi): build -m [-v M], info, read_id, batch_id.  Not provided (outside the query path): read_filter.
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>

#include "colorid_host.hpp"

using namespace colorid;

namespace {

// Which way a run leaves is decided once, here, and `colorid --help` says how:
//   COLORID_FAST_EXIT=1      always the short way;  COLORID_FAST_EXIT=0 or COLORID_FULL_TEARDOWN=1: always the orderly way;
//   neither set              the short way unless something in the process may still have output to write at exit: a preloaded library
//                            (LD_PRELOAD, /etc/ld.so.preload), a profiler's tool library (rocprofv3 writes its files from an exit handler),
//                            coverage counters (LLVM_PROFILE_FILE, GCOV_PREFIX).
// Either way every output file is closed and stdout / stderr are flushed — and checked — before (leave()).
const char *g_exit_reason = "nothing in the process writes at exit";
bool tool_may_write_at_exit() {
    for (const char *v : {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "HSA_TOOLS_LIB", "LLVM_PROFILE_FILE", "GCOV_PREFIX"})
        if (const char *e = getenv(v)) if (*e) { g_exit_reason = v; return true; }   // (not a switch: other tools' variables that announce something writing at exit)
    if (FILE *f = fopen("/etc/ld.so.preload", "r")) {
        int ch;
        bool any = false;
        while ((ch = fgetc(f)) != EOF) if (!isspace(ch)) { any = true; break; }
        fclose(f);
        if (any) { g_exit_reason = "/etc/ld.so.preload"; return true; }
    }
    return false;
}
bool decide_orderly_exit() {
    if (const char *e = cli_env("COLORID_FAST_EXIT")) { g_exit_reason = "COLORID_FAST_EXIT"; return atoi(e) == 0; }
    if (cli_env("COLORID_FULL_TEARDOWN")) { g_exit_reason = "COLORID_FULL_TEARDOWN"; return true; }
    return tool_may_write_at_exit();
}
bool g_orderly_exit = decide_orderly_exit();


struct Args {
    std::map<std::string, std::vector<std::string>> values;  // canonical long name -> values
    std::map<std::string, bool> flags;
    bool has(const std::string &k) const { return values.count(k) && !values.at(k).empty(); }
    const std::string &one(const std::string &k) const { return values.at(k)[0]; }
};

struct OptSpec { char shrt; const char *lng; bool takes_value; bool multi; };

Args parse(int argc, char **argv, int first, const std::vector<OptSpec> &spec) {
    Args a;
    for (int i = first; i < argc; ++i) {
        std::string tok = argv[i];
        const OptSpec *o = nullptr;
        if (tok.size() > 2 && tok[0] == '-' && tok[1] == '-') {
            for (auto &s : spec) if (tok.substr(2) == s.lng) o = &s;
        } else if (tok.size() == 2 && tok[0] == '-') {
            for (auto &s : spec) if (s.shrt == tok[1]) o = &s;
        }
        if (!o) die("error: Found argument '%s' which wasn't expected, or isn't valid in this context", tok.c_str());
        if (!o->takes_value) { a.flags[o->lng] = true; continue; }
        if (i + 1 >= argc) die("error: The argument '--%s' requires a value but none was supplied", o->lng);
        a.values[o->lng].push_back(argv[++i]);
        while (o->multi && i + 1 < argc && argv[i + 1][0] != '-') a.values[o->lng].push_back(argv[++i]);
    }
    return a;
}

template <typename T>
T num_or(const Args &a, const char *k, T dflt) {  // value_t!(..).unwrap_or(dflt): a value that does not parse gives the default
    if (!a.has(k)) return dflt;
    char *end = nullptr;
    const std::string &s = a.one(k);
    if constexpr (std::is_floating_point<T>::value) {
        const double v = strtod(s.c_str(), &end);
        return (end && *end == 0 && !s.empty()) ? (T)v : dflt;
    } else {
        const long long v = strtoll(s.c_str(), &end, 10);
        return (end && *end == 0 && !s.empty()) ? (T)v : dflt;
    }
}

bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// --gpus N (devices 0..N-1) or --devices a,b,... (an id may repeat: several ranks on one GPU): reads / k-mers are sharded over
// the ranks, the index is replicated, per-accession counters are all-reduced (RCCL over xGMI).  The reference's -t (rayon threads,
// src/main.rs:718-721) has no meaning here and is ignored with a note.
// --placement striped (with --gpus / --devices): the INDEX is sharded instead — rank r holds a stripe of colours of every row, every
// rank sees the whole query (an index larger than one GPU's HBM).
struct Gpus {
    cid_group *group = nullptr;            // nullptr: one GPU (--device)
    cid_ctx *ctx = nullptr;                // rank 0's context (or the only one)
    std::vector<cid_index *> replicas;     // replicas, or the stripes
    bool striped = false;
};

std::vector<int> device_list(const Args &a) {
    std::vector<int> ids;
    if (a.has("devices")) {
        const std::string &s = a.one("devices");
        size_t p = 0;
        while (p <= s.size()) {
            const size_t e = s.find(',', p);
            const std::string tok = s.substr(p, e == std::string::npos ? std::string::npos : e - p);
            char *end = nullptr;
            const long v = strtol(tok.c_str(), &end, 10);
            if (tok.empty() || *end) die("--devices expects a comma-separated list of device ids, got '%s'", s.c_str());
            ids.push_back((int)v);
            if (e == std::string::npos) break;
            p = e + 1;
        }
    } else if (a.has("gpus")) {
        const int n = num_or<int>(a, "gpus", 1);
        if (n < 1) die("--gpus expects a positive number");
        for (int i = 0; i < n; ++i) ids.push_back(i);
    }
    return ids;
}

// COLORID_TIMING=1: wall-clock milliseconds of the CLI's phases on stderr
const bool g_timing = cli_env("COLORID_TIMING") != nullptr;
const auto g_t_start = std::chrono::steady_clock::now();
void phase_done(const char *what) {
    if (!g_timing) return;
    static auto last = g_t_start;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "timing: %s %.0f ms (at %.0f ms)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - g_t_start).count());
    last = now;
}

// RCCL prints its version banner (and whatever NCCL_DEBUG asks for) to the C stdout while communicators are made and destroyed; a
// drop-in `colorid search` prints result rows only.  The library never touches a file descriptor, so the CLI does it here, around
// those two calls and nowhere near the searches: while an object of this class lives, fd 1 is fd 2.  (Nothing else of this process
// writes to stdout at those two moments: the banner row is printed before, the result rows between them.)
class StdoutToStderr {
  public:
    StdoutToStderr() {
        fflush(stdout);
        saved_ = dup(1);
        if (saved_ >= 0) dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved_ >= 0) { dup2(saved_, 1); close(saved_); }
    }
  private:
    int saved_;
};

Gpus make_gpus(const Args &a) {
    Gpus g;
    if (a.has("bigsi")) bigsi_read_ahead(a.one("bigsi"));   // the index file's pages come in beside the runtime's start-up
    if (a.has("threads"))
        fprintf(stderr, "note: -t %s is ignored: the search runs on the GPU (--gpus N shards the query over N GPUs)\n", a.one("threads").c_str());
    const std::vector<int> ids = device_list(a);
    if (a.has("placement")) {
        const std::string &pl = a.one("placement");
        if (pl == "striped") g.striped = true;
        else if (pl != "replicated") die("--placement expects replicated or striped, got '%s'", pl.c_str());
    }
    const bool one_rank_group = ids.size() == 1 && (cli_env("COLORID_REDUCE") || g.striped);   // the group path with a single rank
    if (ids.size() <= 1 && !one_rank_group) {
        if (g.striped) die("--placement striped needs --gpus N or --devices a,b,...");
        const int dev = ids.empty() ? num_or<int>(a, "device", 0) : ids[0];
        if (cid_ctx_create(dev, &g.ctx) != CID_OK) die("cannot open GPU %d: %s (colorid has no CPU search path)", dev, cid_last_error());
        return g;
    }
    {
        StdoutToStderr quiet;
        if (cid_group_create(ids.data(), (int)ids.size(), &g.group) != CID_OK) die("cannot open %zu GPUs: %s", ids.size(), cid_last_error());
        g_orderly_exit = true;   // (RCCL communicators are alive from here on: such a run always leaves through release())
    }
    if (cid_group_ctx(g.group, 0, &g.ctx) != CID_OK) die("%s", cid_last_error());
    int rccl = 0;
    cid_group_uses_rccl(g.group, &rccl);
    if (g.striped)
        fprintf(stderr, "%zu ranks, one colour stripe of the index each; per-k-mer facts are summed %s\n", ids.size(),
                rccl ? "with RCCL all-reduce" : "with peer copies");
    else
        fprintf(stderr, "%zu ranks; per-accession counters are reduced %s\n", ids.size(), rccl ? "with RCCL all-reduce" : "through the host");
    return g;
}

// The device code of the kernels a command will launch is loaded on a helper thread while the index loads (cid_warmup): the
// runtime would otherwise load it inside the first search / read_id call (~60 ms for the read_id kernels).  The helper thread is on the
// critical path when the index is small (a 1.7 GB index loads in 42 ms, the read_id units in 64-94): nothing goes in that does not pay for
// itself — not CID_WARM_COLD (40 ms for a path few inputs take), not CID_WARM_PIPES (the index upload has woken the bus and the queues);
// round 6 measured both: profiles/r06_warm_ab.txt.
std::thread warm_async(const Gpus &g, unsigned what) {
    std::vector<cid_ctx *> ctxs;
    if (g.group) {
        int n = 0;
        cid_group_size(g.group, &n);
        for (int r = 0; r < n; ++r) { cid_ctx *c = nullptr; if (cid_group_ctx(g.group, r, &c) == CID_OK) ctxs.push_back(c); }
    } else ctxs.push_back(g.ctx);
    if (const char *e = cli_env("COLORID_WARM")) what = (unsigned)strtoul(e, nullptr, 0);   // (A/B runs: another mask, 0 = nothing ahead of its first use)
    return std::thread([ctxs, what] { for (cid_ctx *c : ctxs) (void)cid_warmup(c, what); });   // (a failure here shows up in the first real call)
}

void replicate(Gpus &g, Bigsi &b) {   // after the index is loaded on rank 0
    if (!g.group) return;
    if (g.striped) { set_stripes(g.group, g.replicas); return; }   // load_index put the stripes in place
    int n = 0;
    cid_group_size(g.group, &n);
    g.replicas.assign((size_t)n, nullptr);
    if (cid_group_replicate_index(g.group, b.index, g.replicas.data()) != CID_OK) die("replicating the index: %s", cid_last_error());
    set_group(g.group, g.replicas);
}

// The process is about to end and its results are written and closed: freeing gigabytes of HBM object by object, unloading the code
// objects and the runtime's own teardown buy nothing — the driver reclaims a process's memory in one go — but cost 0.1-0.2 s of a
// 0.7 s `read_id`.  One-GPU runs therefore skip release() and leave through leave(); COLORID_FULL_TEARDOWN=1 (the sanitizer runs,
// anybody embedding the drivers) keeps the orderly way, and so do multi-GPU runs (RCCL communicators are shut down properly).

void release(Gpus &g, Bigsi &b) {
    if (!g_orderly_exit && !g.group) { cid_ctx_synchronize(g.ctx); return; }
    g_orderly_exit = true;
    for (cid_index *ix : g.replicas)
        if (ix && ix != b.index) cid_index_destroy(ix);
    if (b.index) cid_index_destroy(b.index);
    if (g.group) { StdoutToStderr quiet; cid_group_destroy(g.group); }   // owns the contexts
    else cid_ctx_destroy(g.ctx);
}

cid_ctx *make_ctx(const Args &a) {
    cid_ctx *ctx = nullptr;
    if (a.has("threads"))
        fprintf(stderr, "note: -t %s is ignored: k-mer counting and Bloom inserts run on the GPU\n", a.one("threads").c_str());
    const int dev = num_or<int>(a, "device", 0);
    if (cid_ctx_create(dev, &ctx) != CID_OK) die("cannot open GPU %d: %s (colorid has no CPU search path)", dev, cid_last_error());
    return ctx;
}

const char *const kHashNames[CID_HASH_VARIANTS] = {"xxh3_v08", "xxh3_v07"};

int hash_variant(const Args &a) {
    if (!a.has("hash")) return CID_HASH_XXH3_V08;
    for (int v = 0; v < CID_HASH_VARIANTS; ++v)
        if (a.one("hash") == kHashNames[v]) return v;
    die("unknown --hash '%s' (available: xxh3_v08, xxh3_v07)", a.one("hash").c_str());
}

Bigsi load_index(cid_ctx *ctx, const Args &a, bool meta_only = false, Gpus *gpus = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "Loading index\n");
    // The file format carries no hash id and the reference's hash crate (xxh3 ^0.1.1) cannot be run here: an index written by
    // this program matches its own --hash; for one written by the Rust binary, `colorid hashcheck` decides which --hash applies.
    if (!meta_only && !a.has("hash") && !cli_env("COLORID_QUIET"))
        fprintf(stderr, "note: --hash defaults to xxh3_v08 (published XXH3); parity with an index built by the Rust colorid is unverified — "
                        "run `colorid hashcheck -b <index> -r <ref_file>` once to find the variant it was built with\n");
    const bool striped = gpus && gpus->striped;
    Bigsi b = read_bigsi(ctx, a.one("bigsi"), hash_variant(a), meta_only, striped ? gpus->group : nullptr, striped ? &gpus->replicas : nullptr);
    fprintf(stderr, "Index loaded in %ld seconds\n",
            (long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count());
    return b;
}

const std::vector<OptSpec> kCommon = {{0, "device", true, false}, {0, "hash", true, false}, {0, "gpus", true, false}, {0, "devices", true, false},
                                      {0, "placement", true, false}};

std::vector<OptSpec> with_common(std::vector<OptSpec> v) {
    v.insert(v.end(), kCommon.begin(), kCommon.end());
    return v;
}

int cmd_build(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'r', "refs", true, false}, {'k', "kmer", true, false},
                                                     {'n', "num_hashes", true, false}, {'s', "bloom", true, false}, {'m', "minimizer", false, false},
                                                     {'v', "value", true, false}, {'t', "threads", true, false}, {'Q', "quality", true, false},
                                                     {'f', "filter", true, false}}));
    for (const char *req : {"bigsi", "refs", "kmer", "num_hashes", "bloom"})
        if (!a.has(req)) die("error: The following required arguments were not provided: --%s", req);
    printf(" Ref_file : %s\n Bigsi file : %s\nK-mer size: %s\nBloom filter parameters: num hashes %s, filter size %s\n",
           a.one("refs").c_str(), a.one("bigsi").c_str(), a.one("kmer").c_str(), a.one("num_hashes").c_str(), a.one("bloom").c_str());
    const bool minimizer = a.flags.count("minimizer") > 0;
    uint64_t m_value = 0;
    if (minimizer) {   // main.rs:480-486; -v defaults to 15 and must parse (the reference unwraps)
        const std::string v = a.has("value") ? a.one("value") : std::string("15");
        char *end = nullptr;
        m_value = strtoull(v.c_str(), &end, 10);
        if (v.empty() || *end) die("called `Result::unwrap()` on an `Err` value: ParseIntError (-v %s)", v.c_str());
        printf("Build with minimizers, minimizer size: %llu\n", (unsigned long long)m_value);
    }
    cid_ctx *ctx = make_ctx(a);
    phase_done("GPU context");
    Bigsi b = build_single(ctx, a.one("refs"), num_or<uint64_t>(a, "bloom", 50000000), num_or<uint64_t>(a, "num_hashes", 4),
                           num_or<uint64_t>(a, "kmer", 31), num_or<uint8_t>(a, "quality", 15), num_or<int64_t>(a, "filter", -1),
                           hash_variant(a), m_value);
    phase_done("accessions counted and inserted");
    printf("Saving BIGSI to file.\n");
    save_bigsi(a.one("bigsi") + (minimizer ? ".mxi" : ".bxi"), b);
    phase_done("index written");
    cid_index_destroy(b.index);
    cid_ctx_destroy(ctx);
    return 0;
}

int cmd_search(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'q', "query", true, true}, {'r', "reverse", true, true},
                                                     {'f', "filter", true, false}, {'p', "p_shared", true, false}, {'g', "gene_search", false, false},
                                                     {'s', "perfect_search", false, false}, {'m', "multi_fasta", false, false},
                                                     {'Q', "quality", true, false}}));
    for (const char *req : {"bigsi", "query"})
        if (!a.has(req)) die("error: The following required arguments were not provided: --%s", req);
    const std::vector<std::string> files1 = a.values.at("query");
    const std::vector<std::string> files2 = a.has("reverse") && a.one("reverse") != "none" ? a.values.at("reverse") : std::vector<std::string>{};
    const int64_t filter = num_or<int64_t>(a, "filter", -1);
    const double cov = num_or<double>(a, "p_shared", 0.35);
    const uint8_t quality = num_or<uint8_t>(a, "quality", 15);
    if (ends_with(a.one("bigsi"), ".mxi")) {
        fprintf(stderr, "Error: An index with minimizers (.mxi) is used, but not available for this function\n");
        return 0;
    }
    if (cli_env("COLORID_GPU_INFLATE") && atoi(cli_env("COLORID_GPU_INFLATE")) > 0) LineReader::inflate_on_gpu(num_or<int>(a, "device", 0));
    // gzip decoding of the first query starts now and runs beside GPU start-up and the index load; a block-gzip query on one GPU goes
    // up compressed instead (cid_fastq_count_kmers: its members are read ahead, the k-mer map is counted from text that never leaves HBM)
    unsigned warm_what = CID_WARM_SEARCH;
    if (!a.flags.count("perfect_search") && ends_with(files1[0], "gz")) {
        std::vector<std::string> fq{files1[0]};
        if (!files2.empty()) fq.push_back(files2[0]);
        const bool one_gpu = !a.has("gpus") && !a.has("devices") && !a.has("placement") && !cli_env("COLORID_REDUCE");
        const bool host_kmers = cli_env("COLORID_HOST_KMERS") != nullptr;
        if (one_gpu && !host_kmers && read_id_mt_pe::device_fastq_wanted(fq, fq.size())) {
            warm_what |= CID_WARM_INFLATE | CID_WARM_FASTQ;
            for (const std::string &f : fq)
                BgzfMemberReader::prefetch(f, read_id_mt_pe::device_fastq_stretch_bytes(0), read_id_mt_pe::device_fastq_host_share(),
                                           read_id_mt_pe::device_fastq_host_threads(fq.size()));
        } else {
            for (const std::string &f : fq) LineReader::prefetch(f);
        }
    }
    Gpus gpus = make_gpus(a);
    phase_done("GPU context");
    std::thread warm = warm_async(gpus, warm_what);
    cid_ctx *ctx = gpus.ctx;
    Bigsi b = load_index(ctx, a, false, &gpus);
    replicate(gpus, b);
    warm.join();
    phase_done("index load");
    if (a.flags.count("perfect_search")) {
        if (a.flags.count("multi_fasta")) perfect_search::batch_search_mf(ctx, files1, b);
        else perfect_search::batch_search(ctx, files1, b);
    } else {
        batch_search_pe::batch_search(ctx, files1, files2, b, filter, cov, a.flags.count("gene_search") > 0, quality);
    }
    phase_done("search");
    LineReader::drop_prefetched();
    BgzfMemberReader::drop_prefetched();
    release(gpus, b);
    phase_done("release");
    return 0;
}

int cmd_info(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'c', "compressed", true, false}}));
    if (!a.has("bigsi")) die("error: The following required arguments were not provided: --bigsi");
    Bigsi b = load_index(nullptr, a, /*meta_only=*/true);
    if (b.m_size)   // main.rs:645-648
        printf("BIGSI parameters:\nBloomfilter-size: %llu\nNumber of hashes: %llu\nK-mer size: %llu\n minimizer size: %llu\n\n",
               (unsigned long long)b.bloom_size, (unsigned long long)b.num_hash, (unsigned long long)b.k_size, (unsigned long long)b.m_size);
    else
        printf("BIGSI parameters:\nBloomfilter-size: %llu\nNumber of hashes: %llu\nK-mer size: %llu\n", (unsigned long long)b.bloom_size,
               (unsigned long long)b.num_hash, (unsigned long long)b.k_size);
    printf("Number of accessions in index: %zu\n", b.colors.size());
    for (size_t c = 0; c < b.colors.size(); ++c)  // colour ids are already in sorted-name order
        printf("%s %llu %.3f\n", b.colors[c].c_str(), (unsigned long long)b.n_ref_kmers[c],
               false_prob((double)b.bloom_size, (double)b.num_hash, (double)b.n_ref_kmers[c]));
    return 0;
}

// the knobs read_id and batch_id share (main.rs:704-717 and :869-886 read the same flags with the same defaults)
struct ClassifyFlags {
    size_t down_sample, batch, bitvector_sample;
    double fp_correct;
    uint8_t quality;
};

ClassifyFlags classify_flags(const Args &a) {
    ClassifyFlags f;
    f.down_sample = num_or<size_t>(a, "down_sample", 1);
    f.fp_correct = std::pow(10.0, -num_or<double>(a, "fp_correct", 3.0));
    f.quality = num_or<uint8_t>(a, "quality", 15);
    f.batch = num_or<size_t>(a, "batch", 50000);
    f.bitvector_sample = num_or<size_t>(a, "bitvector_sample", 3);
    if (f.down_sample == 0 || f.batch == 0) die("attempt to calculate the remainder with a divisor of zero");
    return f;
}

bool one_gpu_run(const Args &a) { return !a.has("gpus") && !a.has("devices") && !a.has("placement") && !cli_env("COLORID_REDUCE"); }

// block-gzip input on one GPU goes up compressed and is inflated there (cid_fastq_*)
bool wants_device_front_end(const Args &a, const std::vector<std::string> &fq) {
    return ends_with(fq[0], ".gz") && one_gpu_run(a) && read_id_mt_pe::device_fastq_wanted(fq, fq.size() > 1 ? 2 : 1);
}

// Start reading a sample's files before whoever classifies them asks: gzip decoding (or, for the device front end, reading the
// compressed members: a stretch or two, the reader's queue) then runs beside GPU start-up and the index load — measured against
// starting it after the context exists, the classification phase of 1 M reads ends 130 ms earlier — or, in batch_id, beside the
// tail of the sample before.
void read_ahead(const std::vector<std::string> &fq, bool device_front_end) {
    if (!ends_with(fq[0], ".gz")) return;
    for (size_t i = 0; i < fq.size() && i < 2; ++i) {
        if (device_front_end)
            BgzfMemberReader::prefetch(fq[i], read_id_mt_pe::device_fastq_stretch_bytes(0), read_id_mt_pe::device_fastq_host_share(),
                                       read_id_mt_pe::device_fastq_host_threads(fq.size() > 1 ? 2 : 1));
        else
            LineReader::prefetch(fq[i]);
    }
}

// one sample through the classifier: PREFIX_reads.txt, then PREFIX_counts.txt from it (main.rs:790-866, read_id_batch.rs:37-95)
void classify_sample(cid_ctx *ctx, Bigsi &b, const std::vector<std::string> &fq, const std::string &prefix, const ClassifyFlags &f) {
    if (ends_with(fq[0], ".gz")) {
        if (fq.size() > 1) read_id_mt_pe::per_read_stream_pe(ctx, fq, b, f.down_sample, f.fp_correct, f.batch, prefix, f.quality, f.bitvector_sample);
        else read_id_mt_pe::per_read_stream_se(ctx, fq, b, f.down_sample, f.fp_correct, f.batch, prefix, f.quality, f.bitvector_sample);
    } else {
        read_id_mt_pe::stream_fasta(ctx, fq, b, f.down_sample, f.fp_correct, f.batch, prefix, f.bitvector_sample);
    }
    phase_done("classification");
    read_counts_five_fields(prefix + "_reads.txt", prefix);
    phase_done("counts file");
}

int cmd_read_id(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'q', "query", true, true}, {'c', "batch", true, false},
                                                     {'t', "threads", true, false}, {'n', "prefix", true, false}, {'d', "down_sample", true, false},
                                                     {'H', "high_mem_load", false, false}, {'p', "fp_correct", true, false},
                                                     {'Q', "quality", true, false}, {'B', "bitvector_sample", true, false}}));
    for (const char *req : {"bigsi", "query", "prefix"})
        if (!a.has(req)) die("error: The following required arguments were not provided: --%s", req);
    const std::vector<std::string> fq = a.values.at("query");
    const ClassifyFlags flags = classify_flags(a);
    const std::string prefix = a.one("prefix");
    if (cli_env("COLORID_GPU_INFLATE") && atoi(cli_env("COLORID_GPU_INFLATE")) > 0) LineReader::inflate_on_gpu(num_or<int>(a, "device", 0));
    const bool device_front_end = wants_device_front_end(a, fq);
    read_ahead(fq, device_front_end);
    Gpus gpus = make_gpus(a);
    phase_done("GPU context");
    std::thread warm = warm_async(gpus, CID_WARM_READID | (device_front_end ? CID_WARM_INFLATE | CID_WARM_FASTQ : 0u));
    cid_ctx *ctx = gpus.ctx;
    Bigsi b = load_index(ctx, a, false, &gpus);
    replicate(gpus, b);
    warm.join();
    phase_done("index load");
    classify_sample(ctx, b, fq, prefix, flags);
    LineReader::drop_prefetched();
    BgzfMemberReader::drop_prefetched();
    release(gpus, b);
    phase_done("release");
    return 0;
}

// batch_id (main.rs:869-888 -> read_id_batch.rs:7-181): a sample sheet `name \t reads1 [\t reads2]`; the index is loaded ONCE and
// every sample goes through read_id's streamers with the prefix NAME_TAG.  The reference walks its FnvHashMap of samples in
// hash order; here they go in name order — every sample writes its own two files, so the order shows only on stderr.
// On this machine the point of the subcommand is the fixed cost: one GPU context and one index upload for the whole sheet,
// and the next sample's files are read ahead while the current one is classified.
int cmd_batch_id(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'q', "query", true, false}, {'T', "tag", true, false},
                                                     {'c', "batch", true, false}, {'t', "threads", true, false}, {'d', "down_sample", true, false},
                                                     {'H', "high_mem_load", false, false}, {'p', "fp_correct", true, false},
                                                     {'Q', "quality", true, false}, {'B', "bitvector_sample", true, false}}));
    for (const char *req : {"bigsi", "query", "tag"})
        if (!a.has(req)) die("error: The following required arguments were not provided: --%s", req);
    const ClassifyFlags flags = classify_flags(a);
    const std::string tag = a.one("tag");
    const auto sheet = tab_to_map(a.one("query"));
    std::vector<std::pair<std::string, std::vector<std::string>>> samples(sheet.begin(), sheet.end());
    if (cli_env("COLORID_GPU_INFLATE") && atoi(cli_env("COLORID_GPU_INFLATE")) > 0) LineReader::inflate_on_gpu(num_or<int>(a, "device", 0));
    std::vector<char> on_device(samples.size(), 0);
    bool any_on_device = false;
    for (size_t i = 0; i < samples.size(); ++i) any_on_device |= (on_device[i] = wants_device_front_end(a, samples[i].second));
    if (!samples.empty()) read_ahead(samples[0].second, on_device[0]);
    Gpus gpus = make_gpus(a);
    phase_done("GPU context");
    std::thread warm = warm_async(gpus, CID_WARM_READID | (any_on_device ? CID_WARM_INFLATE | CID_WARM_FASTQ : 0u));
    cid_ctx *ctx = gpus.ctx;
    Bigsi b = load_index(ctx, a, false, &gpus);
    replicate(gpus, b);
    warm.join();
    phase_done("index load");
    for (size_t i = 0; i < samples.size(); ++i) {
        fprintf(stderr, "Classifying %s\n", samples[i].first.c_str());
        if (i + 1 < samples.size()) read_ahead(samples[i + 1].second, on_device[i + 1]);
        classify_sample(ctx, b, samples[i].second, samples[i].first + "_" + tag, flags);
    }
    LineReader::drop_prefetched();
    BgzfMemberReader::drop_prefetched();
    release(gpus, b);
    phase_done("release");
    return 0;
}

// Which hash variant was this index built with?  The .bxi/.mxi format has no hash id and the reference hashes with a 2019 crate
// (xxh3 ^0.1.1, Cargo.toml:9; call sites src/simple_bloom.rs:19-38) that cannot run here.  A Bloom filter has no false negatives:
// under the RIGHT variant every k-mer of an accession's own sequence file has all its n rows set in the accession's colour
// (src/build.rs:54-99 inserted exactly those k-mers), under a wrong one only the fraction ~ (row density)^n does.  For every
// variant and every accession of the reference list (`-r`, the file `build` was given) this prints that fraction.
int cmd_hashcheck(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, with_common({{'b', "bigsi", true, false}, {'r', "refs", true, false}, {'Q', "quality", true, false}}));
    for (const char *req : {"bigsi", "refs"})
        if (!a.has(req)) die("error: The following required arguments were not provided: --%s", req);
    if (a.has("hash")) die("hashcheck tries every variant: --hash is not accepted");
    cid_ctx *ctx = make_ctx(a);
    fprintf(stderr, "Loading index\n");
    Bigsi b = read_bigsi(ctx, a.one("bigsi"), CID_HASH_XXH3_V08, false);
    std::vector<double> worst(CID_HASH_VARIANTS, 2.0);
    const size_t n_checked = hashcheck(ctx, b, a.one("refs"), num_or<uint8_t>(a, "quality", 15), kHashNames, worst);
    if (n_checked == 0) die("no accession of %s is a colour of %s", a.one("refs").c_str(), a.one("bigsi").c_str());
    int match = -1, n_match = 0;
    for (int v = 0; v < CID_HASH_VARIANTS; ++v)
        if (worst[v] >= 0.999) { match = v; ++n_match; }
    if (n_match == 1) printf("verdict\t%s\tevery accession's own k-mers are present: use --hash %s\n", kHashNames[match], kHashNames[match]);
    else if (n_match == 0) printf("verdict\tnone\tno available hash variant reproduces this index (lowest fractions:");
    else printf("verdict\tambiguous\tmore than one variant fits (index too dense to tell)\n");
    if (n_match == 0) {
        for (int v = 0; v < CID_HASH_VARIANTS; ++v) printf(" %s %.4f", kHashNames[v], worst[v]);
        printf(")\n");
    }
    cid_index_destroy(b.index);
    cid_ctx_destroy(ctx);
    return n_match == 1 ? 0 : 3;
}

// host-only helper used by the CPU tests: distinct canonical k-mers of a file as "kmer\tcount" lines (insertion order)
int cmd_debug_kmers(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {{'q', "query", true, true}, {'k', "kmer", true, false}, {0, "mode", true, false},
                                         {'Q', "quality", true, false}, {'f', "filter", true, false}});
    KmerMap km(num_or<uint32_t>(a, "kmer", 31));
    const std::string mode = a.has("mode") ? a.one("mode") : "vector";
    const std::vector<std::string> q = a.values.at("query");
    if (mode == "vector") kmerize_vector(read_fasta(q[0]), 1, km);
    else if (mode == "fq") kmers_from_fq_qual(q[0], num_or<uint8_t>(a, "quality", 15), km);
    else if (mode == "fqpe") kmers_fq_pe_qual(q[0], q[1], num_or<uint8_t>(a, "quality", 15), km);
    else if (mode == "mf") {
        std::vector<std::string> labels, seqs;
        read_fasta_mf(q[0], labels, seqs);
        for (size_t i = 0; i < labels.size() && i < seqs.size(); ++i) {
            KmerMap one(km.k());
            printf(">%s\t%d\n", labels[i].c_str(), kmerize_string(seqs[i], one) ? (int)one.size() : -1);
        }
        return 0;
    } else die("unknown mode");
    if (a.has("filter")) {
        const int64_t f = num_or<int64_t>(a, "filter", 0);
        if (f < 0) { const int64_t t = km.auto_cutoff(); printf("#auto_cutoff\t%lld\n", (long long)t); if (t >= 0) km.clean((uint64_t)t); }
        else km.clean((uint64_t)f);
    }
    for (size_t e = 0; e < km.size(); ++e) {
        fwrite(km.keys() + e * km.k(), 1, km.k(), stdout);
        printf("\t%u\n", km.counts()[e]);
    }
    return 0;
}

// host-only helper used by the CPU tests: the reads of a FASTQ file (pair) as the record pipeline packs them for read_id / search
int cmd_debug_records(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {{'q', "query", true, true}, {'Q', "quality", true, false}});
    const std::vector<std::string> q = a.values.at("query");
    debug_records(q[0], q.size() > 1 ? &q[1] : nullptr, num_or<uint8_t>(a, "quality", 15));
    return 0;
}

// the end of a subcommand: everything it printed leaves the stdio buffers, then the process ends without the teardown (see release())
int leave(int rc) {
    phase_done("subcommand returned");
    // a result that did not reach its pipe or disk is a failed run, whichever way the process leaves (EX_IOERR)
    const bool lost = fflush(nullptr) != 0 || ferror(stdout);
    if (lost) {
        fprintf(stderr, "colorid: writing the output failed: %s\n", strerror(errno));
        if (rc == 0) rc = 74;
    }
    if (!g_orderly_exit) _exit(rc);
    return rc;
}

}  // namespace

int main(int argc, char **argv) {
    // src/main.rs:16-20: init_log() prints this banner on stdout before anything else
    printf("\n ************** initializing logger *****************\n\n");
    if (argc < 2) {
        fprintf(stderr, "colorid 0.1.4.3 (MI355X)\nUSAGE:\n    colorid <build|search|info|read_id|batch_id|hashcheck> [FLAGS]\n");
        return 1;
    }
    const std::string cmd = argv[1];
    if (cmd == "--help" || cmd == "-h" || cmd == "help") {
        printf("colorid 0.1.4.3 (MI355X)\nUSAGE:\n    colorid <build|search|info|read_id|batch_id|hashcheck> [FLAGS]      (flags: colorid <subcommand> --help)\n\n"
               "ENVIRONMENT:\n"
               "    COLORID_FAST_EXIT=1      leave without the GPU runtime's teardown once the results are written, closed and flushed\n"
               "                             (-0.05 to -0.15 s per run); =0, or COLORID_FULL_TEARDOWN=1: always the orderly exit\n"
               "                             unset: the short way, unless a preloaded library, a profiler's tool library or coverage\n"
               "                             counters are present (they write at exit) or several GPUs are in use (RCCL is shut down)\n"
               "                             this run would leave: %s (%s)\n"
               "    COLORID_INDEX_MMAP=0     read the index through a buffered reader instead of a mapping\n"
               "    COLORID_DEVICE_FASTQ=0   FASTQ text on the host front end only\n",
               g_orderly_exit ? "the orderly way" : "the short way", g_exit_reason);
        return 0;
    }
    if (cmd == "build") return leave(cmd_build(argc, argv));
    if (cmd == "search") return leave(cmd_search(argc, argv));
    if (cmd == "info") return leave(cmd_info(argc, argv));
    if (cmd == "read_id") return leave(cmd_read_id(argc, argv));
    if (cmd == "hashcheck") return leave(cmd_hashcheck(argc, argv));
    if (cmd == "debug-kmers") return cmd_debug_kmers(argc, argv);
    if (cmd == "debug-records") return cmd_debug_records(argc, argv);
    if (cmd == "batch_id") return leave(cmd_batch_id(argc, argv));
    if (cmd == "read_filter") die("'%s' is outside the accelerated query path; use the reference binary", cmd.c_str());
    die("error: Found argument '%s' which wasn't expected", cmd.c_str());
}
