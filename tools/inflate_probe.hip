// Where k_bgzf_inflate's cycles go: the kernel compiled with CID_INFLATE_STAMPS (cycle counters of lane 0 of every wave, summed per part:
// ring refill, decode runs, match copies, CRC) over members made here with zlib from synthetic FASTQ text.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCID_INFLATE_STAMPS -Icolorid_amd/csrc tools/inflate_probe.hip -o tools/bin/inflate_probe -lz
// CID_INFLATE_WAVE=0: the one-lane kernel alone (round 4's); default: one member per wave, 64 lanes on a block's chunks (the stamps then cover the retries only)
// usage: tools/bin/inflate_probe [members] [level] [kind: 0 random bases + 11 quality letters, 1 repetitive (reads of one 5 kb genome, 4 quality letters), 2 the same reads with 41 skewed quality values]
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../colorid_amd/csrc/cid_inflate.hip"

namespace cid {   // what cid_inflate.hip expects from the library around it
int fail(int code, const char *fmt, ...) { (void)fmt; return code; }
int ctx_alloc(cid_ctx *, size_t, void **) { return CID_ERR_NOMEM; }
void ctx_free(cid_ctx *, void *) {}
void *pin_reserve(cid_ctx *, size_t) { return nullptr; }
int slot_reserve(cid_ctx *, int, size_t, void **) { return CID_ERR_NOMEM; }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t n_members = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int kind = argc > 3 ? atoi(argv[3]) : 0;
    std::mt19937_64 rng(1);
    std::string genome(5000, 'A');
    for (char &ch : genome) ch = "ACGT"[rng() & 3];
    std::string text;
    size_t read_no = 0;
    while (text.size() < n_members * 65280) {
        text += "@SRR1234567." + std::to_string(read_no) + " " + std::to_string(read_no) + " length=150\n";
        ++read_no;
        if (kind == 0) for (int i = 0; i < 150; ++i) text += "ACGT"[rng() & 3];
        else { const size_t p = rng() % (genome.size() - 150); text.append(genome, p, 150); }
        text += "\n+\n";
        if (kind == 0) for (int i = 0; i < 150; ++i) text += "FFFFFFFF:,#"[rng() % 11];
        else if (kind == 1) for (int i = 0; i < 150; ++i) text += "FFF:"[(rng() % 16) < 13 ? 0 : rng() & 3];
        else for (int i = 0; i < 150; ++i) text += (char)('!' + (rng() % 100 < 70 ? 37 + rng() % 4 : rng() % 41));   // 41 quality values, skewed: literal-heavy
        text += "\n";
    }
    std::string in;
    std::vector<cid::BgzfMember> mem;
    for (size_t m = 0; m < n_members; ++m) {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(text.data()) + m * 65280;
        std::string body(70000, '\0');
        z_stream zs{};
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<unsigned char *>(src); zs.avail_in = 65280;
        zs.next_out = reinterpret_cast<unsigned char *>(&body[0]); zs.avail_out = (unsigned)body.size();
        deflate(&zs, Z_FINISH);
        body.resize(zs.total_out);
        deflateEnd(&zs);
        const uint32_t crc = (uint32_t)crc32(0, src, 65280), isize = 65280;
        const uint16_t bsize = (uint16_t)(12 + 6 + body.size() + 8 - 1);
        std::string one("\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0", 16);
        one.append(reinterpret_cast<const char *>(&bsize), 2);
        one += body;
        one.append(reinterpret_cast<const char *>(&crc), 4);
        one.append(reinterpret_cast<const char *>(&isize), 4);
        mem.push_back(cid::BgzfMember{(uint32_t)in.size(), (uint32_t)one.size(), (uint32_t)(m * 65280), 65280});
        in += one;
    }
    uint8_t *d_in, *d_out; cid::BgzfMember *d_mem; uint32_t *d_st;
    CHECK(hipMalloc(&d_in, in.size() + 16)); CHECK(hipMalloc(&d_out, n_members * 65280 + 16)); CHECK(hipMalloc(&d_mem, mem.size() * sizeof(mem[0])));
    CHECK(hipMalloc(&d_st, n_members * 4));
    void *d_scratch = nullptr;   // the wave-parallel kernel's match tokens and retry list (CID_INFLATE_WAVE=0: one lane per member)
    CHECK(hipMalloc(&d_scratch, cid::bgzf_inflate_scratch_bytes((uint32_t)n_members)));
    CHECK(hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_mem, mem.data(), mem.size() * sizeof(mem[0]), hipMemcpyHostToDevice));
    cid_ctx ctx;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    ctx.n_cu = prop.multiProcessorCount;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[8] = {0};
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(cid::g_inflate_stamps), zero, sizeof zero));
        CHECK(hipMemset(d_out, 0, n_members * 65280));
        CHECK(hipEventRecord(e0, 0));
        CHECK(cid::bgzf_inflate_launch(&ctx, 0, d_in, d_mem, (uint32_t)n_members, d_out, d_st, d_scratch));
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[8];
        CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(cid::g_inflate_stamps), sizeof st));
        const double waves = (double)((n_members + 1) / 2);
        printf("%zu members, level %d, kind %d: %.2f ms (%.1f MB compressed); cycles per wave: refill %.0fk, decode %.0fk, copies %.0fk, crc %.0fk\n", n_members, level,
               kind, ms, in.size() / 1e6, st[0] / waves / 1e3, st[1] / waves / 1e3, st[2] / waves / 1e3, st[3] / waves / 1e3);
        if (st[4] + st[5] + st[6] + st[7])
            printf("    one member per wave, cycles per member: headers + tables %.0fk, passes %.0fk, writing %.0fk, copies %.0fk\n", st[4] / (double)n_members / 1e3,
                   st[5] / (double)n_members / 1e3, st[6] / (double)n_members / 1e3, st[7] / (double)n_members / 1e3);
    }
    std::vector<uint32_t> stt(n_members);
    std::string out(n_members * 65280, '\0');
    CHECK(hipMemcpy(stt.data(), d_st, n_members * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&out[0], d_out, out.size(), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (uint32_t v : stt) bad += v != 0;
    {   // how many members the wave-parallel kernel left for the one-lane kernel
        uint32_t n_retry = 0;
        CHECK(hipMemcpy(&n_retry, reinterpret_cast<uint8_t *>(d_scratch) + n_members * (size_t)cid::kWaveTokens * sizeof(uint2), 4, hipMemcpyDeviceToHost));
        printf("left for the one-lane kernel: %u of %zu members\n", n_retry, n_members);
    }
    printf("status: %zu bad members; text %s\n", bad, out.compare(0, out.size(), text, 0, out.size()) == 0 ? "identical" : "DIFFERS");
    return bad != 0;
}
