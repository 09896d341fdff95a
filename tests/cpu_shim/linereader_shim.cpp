// Test shim (not product): the CLI's LineReader on one file — number of lines, total bytes and an FNV-1a hash of the lines, so a
// CPU test can compare plain text, single-stream gzip and block gzip (BGZF, inflated by several threads) inputs.
// argv[2]: "copy" (default; next(std::string&)), "view" (next(ptr, len): lines as views into the decoded blocks), "mixed" (the two
// calls alternating), "prefetch" (LineReader::prefetch first — the stream is inflated ahead and taken over by the reader — then views).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "../../colorid_amd/csrc/host/colorid_host.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string mode = argc > 2 ? argv[2] : "copy";
    if (mode == "prefetch") {
        colorid::LineReader::prefetch(argv[1]);
        colorid::LineReader::prefetch(argv[1]);   // a second stream on the same path that nobody takes: dropped at the end
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    uint64_t n = 0, bytes = 0, h = 0xcbf29ce484222325ull;
    {
        colorid::LineReader r(argv[1]);
        std::string line;
        const char *p = nullptr;
        size_t len = 0;
        for (;;) {
            const bool view = mode == "view" || mode == "prefetch" || (mode == "mixed" && (n % 3) != 0);
            if (view) { if (!r.next(p, len)) break; }
            else { if (!r.next(line)) break; p = line.data(); len = line.size(); }
            ++n; bytes += len;
            for (size_t i = 0; i < len; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
            h ^= 0x0a; h *= 0x100000001b3ull;
        }
    }
    colorid::LineReader::drop_prefetched();
    printf("%llu %llu %016llx\n", (unsigned long long)n, (unsigned long long)bytes, (unsigned long long)h);
    return 0;
}
