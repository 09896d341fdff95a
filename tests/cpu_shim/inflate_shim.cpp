// CPU test driver of colorid_amd/csrc/host/fast_inflate.hpp: decodes a raw DEFLATE stream the way LineReader::Impl::run_gzip does — input
// in pieces of `in_chunk` bytes behind the unread rest (16 readable bytes of padding behind them), output in blocks of `out_block` bytes
// with the last 32 KiB of the stream kept in front of every block — and writes the text to `out_path`.
// usage: inflate_shim <deflate file> <out path> <in_chunk> <out_block>     exit 0 ok, 2 decoder error (message on stderr)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../colorid_amd/csrc/host/fast_inflate.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 1;
    FILE *f = fopen(argv[1], "rb");
    FILE *o = fopen(argv[2], "wb");
    if (!f || !o) return 1;
    const size_t in_chunk = (size_t)atol(argv[3]), out_block = (size_t)atol(argv[4]);
    constexpr size_t kHist = 32768;
    std::vector<unsigned char> in(in_chunk + colorid::FastInflate::kMargin + 4096 + 16), blk(kHist + out_block), hist(kHist);
    size_t pos = 0, end = 0, hist_n = 0;
    bool eof = false;
    colorid::FastInflate z;
    z.reset();
    unsigned char *out = blk.data() + kHist;
    for (;;) {
        const unsigned char *ip = in.data() + pos;
        const colorid::FastInflate::Result r = z.run(ip, in.data() + end, eof, out, blk.data() + blk.size());
        pos = (size_t)(ip - in.data());
        if (r == colorid::FastInflate::kError) { fprintf(stderr, "%s\n", z.error()); return 2; }
        if (r == colorid::FastInflate::kNeedInput) {
            if (eof) { fprintf(stderr, "decoder wants input after the end\n"); return 2; }
            memmove(in.data(), in.data() + pos, end - pos);
            end -= pos; pos = 0;
            const size_t room = in.size() - 16 - end;
            const size_t n = fread(in.data() + end, 1, room < in_chunk ? room : in_chunk, f);
            if (n == 0) eof = true;
            end += n;
            memset(in.data() + end, 0, 16);
            continue;
        }
        // output full or the stream's end: hand the block over, keep its last 32 KiB (with what was kept before, if the block is short)
        const size_t made = (size_t)(out - (blk.data() + kHist));
        fwrite(blk.data() + kHist, 1, made, o);
        if (made >= kHist) { memcpy(hist.data(), out - kHist, kHist); hist_n = kHist; }
        else {
            const size_t keep = hist_n + made > kHist ? kHist - made : hist_n;
            memmove(hist.data(), hist.data() + (hist_n - keep), keep);
            memcpy(hist.data() + keep, blk.data() + kHist, made);
            hist_n = keep + made;
        }
        if (r == colorid::FastInflate::kStreamEnd) break;
        memcpy(blk.data() + kHist - hist_n, hist.data(), hist_n);
        out = blk.data() + kHist;
    }
    // what follows the stream in the file (a container's trailer) must be exactly where the decoder stopped
    fprintf(stderr, "unused %zu\n", (end - pos) + 0);
    fclose(o);
    return 0;
}
