// Test shim (not product): the CLI's LineReader on one file — number of lines, total bytes and an FNV-1a hash of the lines, so a
// CPU test can compare plain text, single-stream gzip and block gzip (BGZF, inflated by several threads) inputs.
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../colorid_amd/csrc/host/colorid_host.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    colorid::LineReader r(argv[1]);
    std::string line;
    uint64_t n = 0, bytes = 0, h = 0xcbf29ce484222325ull;
    while (r.next(line)) {
        ++n; bytes += line.size();
        for (unsigned char c : line) { h ^= c; h *= 0x100000001b3ull; }
        h ^= 0x0a; h *= 0x100000001b3ull;
    }
    printf("%llu %llu %016llx\n", (unsigned long long)n, (unsigned long long)bytes, (unsigned long long)h);
    return 0;
}
