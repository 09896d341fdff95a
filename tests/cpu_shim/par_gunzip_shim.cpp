// CPU test driver of colorid_amd/csrc/host/par_gunzip.hpp: one raw DEFLATE stream decoded in chunks on `threads` threads, input handed over
// in pieces of `read_piece` bytes, the text written to `out_path`; prints "crc <crc32> total <bytes> leftover <bytes> chunks <n> accepted <n>
// serial <n> rounds <n>" on stdout.
// usage: par_gunzip_shim <deflate file> <out path> <chunk_bytes> <n_chunks> <threads> <read_piece>     exit 0 ok, 2 decoder error (message on stderr)
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../colorid_amd/csrc/host/par_gunzip.hpp"

int main(int argc, char **argv) {
    if (argc < 7) return 1;
    FILE *f = fopen(argv[1], "rb");
    FILE *o = fopen(argv[2], "wb");
    if (!f || !o) return 1;
    const size_t chunk = (size_t)atol(argv[3]), n_chunks = (size_t)atol(argv[4]), threads = (size_t)atol(argv[5]), piece = (size_t)atol(argv[6]);
    colorid::ParallelInflate pi(chunk, n_chunks);
    auto reader = [&](uint8_t *dst, size_t cap) -> size_t { return fread(dst, 1, cap < piece ? cap : piece, f); };
    auto sink = [&](const uint8_t *t, size_t n) -> bool { return fwrite(t, 1, n, o) == n; };
    auto pfor = [&](size_t n, const std::function<void(size_t)> &fn) {
        std::vector<std::thread> th;
        for (size_t t = 0; t < threads; ++t)
            th.emplace_back([&, t] { for (size_t i = t; i < n; i += threads) fn(i); });
        for (std::thread &x : th) x.join();
    };
    auto crc = [](uint32_t c, const uint8_t *p, size_t n) -> uint32_t { return (uint32_t)crc32(c, p, (uInt)n); };
    auto comb = [](uint32_t a, uint32_t b, uint64_t n) -> uint32_t { return (uint32_t)crc32_combine(a, b, (z_off_t)n); };
    std::vector<uint8_t> first(100);
    const size_t nf = fread(first.data(), 1, first.size(), f);
    const bool ok = pi.run(first.data(), nf, reader, sink, pfor, crc, comb);
    fclose(o);
    if (!ok) { fprintf(stderr, "%s\n", pi.error()); return 2; }
    printf("crc %u total %llu leftover %zu chunks %llu accepted %llu serial %llu rounds %llu\n", pi.crc(), (unsigned long long)pi.total(), pi.leftover().size(),
           (unsigned long long)pi.stats.chunks, (unsigned long long)pi.stats.accepted, (unsigned long long)pi.stats.serial, (unsigned long long)pi.stats.rounds);
    return 0;
}
