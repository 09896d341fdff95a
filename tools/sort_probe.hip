// How fast can 120 M u64 k-mer codes (63 significant bits: 2k + 1 at k = 31) be radix-sorted on one MI355X?  rocPRIM's onesweep with
// its default configuration (8 bits per pass: 8 passes) against wider digits (9 / 10 / 11 bits: 7 / 7 / 6 passes) and other tile shapes.
// Prints one line per configuration: ms per sort (best of 5), passes, GB/s of key traffic (2 x 8 B per key per pass).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_probe.hip -o tools/bin/sort_probe && tools/bin/sort_probe [n_keys] [end_bit]
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../colorid_amd/csrc/cid_kernels.hpp"   // (kNoKey)
#include "../colorid_amd/csrc/cid_partition.hpp"

#define CHECK(e)                                                                     \
    do {                                                                             \
        hipError_t e_ = (e);                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__global__ void k_fill(uint64_t *p, uint64_t n, unsigned end_bit, int mode) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i + 0x9E3779B97F4A7C15ull;   // SplitMix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    if (mode == 1) {          // every k-mer ~60 times (reads at 60x coverage): 2 M distinct codes
        uint64_t y = (z % 2000003ull) + 0x9E3779B97F4A7C15ull;
        y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ull; y = (y ^ (y >> 27)) * 0x94D049BB133111EBull; z = y ^ (y >> 31);
    } else if (mode == 2) {   // repetitive sequence: the keys of a run agree in 36 more bits, only the low 10 + 16 top bits vary
        z = (z & 0xFFFF0000000003FFull) | (((z >> 48) * 0x9E3779B1ull & 0xFFFFFFFFFull) << 10);
    }
    p[i] = end_bit >= 64 ? z : (z & ((1ull << (end_bit - 1)) - 1));   // codes below the sentinel bit, as real windows are
}
__global__ void k_check(const uint64_t *p, uint64_t n, int *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && p[i] > p[i + 1]) atomicAdd(bad, 1);
}

template <class Config>
void run(const char *name, const uint64_t *in, uint64_t *out, size_t n, unsigned end_bit, unsigned bits, int *d_bad) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_keys<Config>(nullptr, tb, in, out, n, 0u, end_bit, hipStreamDefault);
    if (e != hipSuccess) { printf("%-44s refused: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    void *tmp = nullptr;
    CHECK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CHECK(hipEventRecord(e0));
        e = rocprim::radix_sort_keys<Config>(tmp, tb, in, out, n, 0u, end_bit, hipStreamDefault);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        if (e != hipSuccess) { printf("%-44s failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); break; }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    CHECK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, out, (uint64_t)n, d_bad);
    int bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    const unsigned passes = (end_bit + bits - 1) / bits;
    printf("%-44s %7.3f ms  %u passes  %6.0f GB/s of key traffic  tmp %zu MB  %s\n", name, best, passes, 16.0 * n * passes / (best * 1e-3) / 1e9,
           tb >> 20, bad ? "NOT SORTED" : "sorted");
    fflush(stdout);
    CHECK(hipFree(tmp));
}

template <unsigned BS, unsigned IPT, unsigned BITS, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::match>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

// two MSD partition passes (cid_partition.hpp) + every run finished in LDS (k_run_sort); rocPRIM's segmented sort for comparison
static void run_msd(const uint64_t *in, uint64_t *out, uint64_t *tmp_keys, size_t n, unsigned end_bit, int *d_bad) {
    using namespace cid;
    const uint32_t S1 = 1, S2 = kPartBins, S3 = kPartBins * kPartBins;
    const uint32_t top = end_bit - 1;   // the sentinel's bit: real codes live below it (the product clamps the sentinel into the last bin)
    const uint32_t max_tiles = part_max_tiles((uint32_t)n, S2);
    uint32_t *seg0, *seg1, *seg2, *tile_base, *table, *d_info, *hard;
    PartTile *desc;
    CHECK(hipMalloc(&desc, ((size_t)max_tiles + 1) * sizeof(PartTile)));
    CHECK(hipMalloc(&seg0, 2 * 4)); CHECK(hipMalloc(&seg1, (S2 + 1) * 4)); CHECK(hipMalloc(&seg2, (S3 + 1) * 4));
    CHECK(hipMalloc(&tile_base, (S2 + 1) * 4)); CHECK(hipMalloc(&table, (size_t)max_tiles * kPartBins * 4)); CHECK(hipMalloc(&d_info, 1024 * 4)); CHECK(hipMalloc(&hard, (S3 + 1) * 4));
    CHECK(hipMemset(d_info, 0, 64));
    const uint32_t h0[2] = {0, (uint32_t)n};
    CHECK(hipMemcpy(seg0, h0, 8, hipMemcpyHostToDevice));
    size_t scan_tb = 0, seg_tb = 0;
    CHECK(rocprim::exclusive_scan(nullptr, scan_tb, table, table, 0u, (size_t)max_tiles * kPartBins, rocprim::plus<uint32_t>(), hipStreamDefault));
    CHECK(rocprim::segmented_radix_sort_keys(nullptr, seg_tb, tmp_keys, out, (unsigned)n, S3, seg2, seg2 + 1, 0u, top - 16, hipStreamDefault));
    void *scan_tmp, *seg_tmp;
    CHECK(hipMalloc(&scan_tmp, scan_tb)); CHECK(hipMalloc(&seg_tmp, seg_tb ? seg_tb : 16));
    hipEvent_t ev[6];
    for (auto &e : ev) CHECK(hipEventCreate(&e));
    const unsigned grid = 256 * 8;
    auto pass = [&](const uint64_t *src, uint64_t *dst, const uint32_t *seg, uint32_t S, uint32_t shift, uint32_t *seg_next) {
        CHECK(hipMemsetAsync(table, 0, (size_t)max_tiles * kPartBins * 4, hipStreamDefault));
        hipLaunchKernelGGL(k_part_tiles, dim3(1), dim3(kPartBlock), 0, 0, seg, S, tile_base);
        hipLaunchKernelGGL(k_part_tile_desc, dim3((max_tiles + 255) / 256), dim3(256), 0, 0, seg, tile_base, S, kPartBins, desc);
        hipLaunchKernelGGL(k_part_hist, dim3(grid), dim3(kPartBlock), 0, 0, (const PartTile *)desc, src, seg, tile_base, S, shift, 8u, 64u, table, d_info + 8);
        CHECK(rocprim::exclusive_scan(scan_tmp, scan_tb, table, table, 0u, (size_t)max_tiles * kPartBins, rocprim::plus<uint32_t>(), hipStreamDefault));
        hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(kPartBlock), 0, 0, (const PartTile *)desc, src, dst, seg, tile_base, S, shift, 8u, 64u, table);
        hipLaunchKernelGGL(k_part_offsets, dim3((S * kPartBins + 256) / 256), dim3(256), 0, 0, table, tile_base, S, 8u, (uint32_t)n, d_info + 8, seg_next);
        CHECK(hipGetLastError());
    };
    float best[5] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    uint32_t info[2] = {0, 0};
    for (int it = 0; it < 6; ++it) {
        CHECK(hipEventRecord(ev[0]));
        pass(in, out, seg0, S1, top - 8, seg1);
        CHECK(hipEventRecord(ev[1]));
        pass(out, tmp_keys, seg1, S2, top - 16, seg2);
        CHECK(hipEventRecord(ev[2]));
        CHECK(hipMemsetAsync(d_info, 0, 8, hipStreamDefault));
        hipLaunchKernelGGL(k_run_sizes, dim3((S3 + 255) / 256), dim3(256), 0, 0, seg2, S3, 8192u, d_info, d_info + 2, 1000u, d_info + 1);
        CHECK(hipMemcpy(info, d_info, 8, hipMemcpyDeviceToHost));
        CHECK(hipEventRecord(ev[3]));
        if (getenv("PROBE_LSD")) {
            if (info[1] <= 2048) hipLaunchKernelGGL(k_run_sort<8>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 2048u, nullptr, nullptr);
            else if (info[1] <= 4096) hipLaunchKernelGGL(k_run_sort<16>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 4096u, nullptr, nullptr);
            else hipLaunchKernelGGL(k_run_sort<32>, dim3(256 * 2), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 8192u, nullptr, nullptr);
        } else {   // bucket sort; the hard runs (none on random keys) through the radix kernel sized for the largest run
            CHECK(hipMemsetAsync(d_info + 4, 0, 4, hipStreamDefault));
            hipLaunchKernelGGL(k_run_bucket_sort<8>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, d_info + 4, hard);
            if (info[1] > 2048) hipLaunchKernelGGL(k_run_bucket_sort<16>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 2049u, d_info + 4, hard);
            if (info[1] <= 2048) hipLaunchKernelGGL(k_run_sort<8>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 2048u, hard, d_info + 4);
            else if (info[1] <= 4096) hipLaunchKernelGGL(k_run_sort<16>, dim3(grid), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 4096u, hard, d_info + 4);
            else hipLaunchKernelGGL(k_run_sort<32>, dim3(256 * 2), dim3(kPartBlock), 0, 0, tmp_keys, out, seg2, S3, top - 16, 1u, 8192u, hard, d_info + 4);
        }
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(ev[4]));
        CHECK(hipEventSynchronize(ev[4]));
        float a, b, c, d;
        CHECK(hipEventElapsedTime(&a, ev[0], ev[1])); CHECK(hipEventElapsedTime(&b, ev[1], ev[2])); CHECK(hipEventElapsedTime(&c, ev[2], ev[3]));
        CHECK(hipEventElapsedTime(&d, ev[3], ev[4]));
        if (it && a + b + c + d < best[4]) { best[0] = a; best[1] = b; best[2] = c; best[3] = d; best[4] = a + b + c + d; }
    }
    CHECK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, out, (uint64_t)n, d_bad);
    int bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    uint32_t n_hard = 0;
    CHECK(hipMemcpy(&n_hard, d_info + 4, 4, hipMemcpyDeviceToHost));
    printf("%-44s %7.3f ms = pass 1 %.3f + pass 2 %.3f + run sizes %.3f + runs sorted in LDS %.3f (largest of %u runs: %u keys, %u above 8192; %u hard runs)  %s\n",
           "2 MSD passes + LDS run sort", best[4], best[0], best[1], best[2], best[3], S3, info[1], info[0], n_hard, bad || info[0] ? "NOT SORTED" : "sorted");
    // the same runs through rocPRIM's segmented sort, for comparison
    float ms = 0;
    CHECK(hipEventRecord(ev[0]));
    CHECK(rocprim::segmented_radix_sort_keys(seg_tmp, seg_tb, tmp_keys, out, (unsigned)n, S3, seg2, seg2 + 1, 0u, top - 16, hipStreamDefault));
    CHECK(hipEventRecord(ev[1]));
    CHECK(hipEventSynchronize(ev[1]));
    CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    printf("%-44s %7.3f ms\n", "  (rocPRIM segmented sort of the same runs)", ms);
    // checksum: the multiset is unchanged
    fflush(stdout);
}

__global__ void k_sum(const uint64_t *p, uint64_t n, unsigned long long *acc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t v = i < n ? p[i] * 0x9E3779B97F4A7C15ull + (p[i] >> 7) : 0;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, (unsigned long long)v);
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 120000000ull;
    const unsigned end_bit = argc > 2 ? (unsigned)atoi(argv[2]) : 63u;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    uint64_t *in, *out;
    int *d_bad;
    CHECK(hipMalloc(&in, n * 8));
    CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMalloc(&d_bad, 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, in, (uint64_t)n, end_bit, mode);
    CHECK(hipDeviceSynchronize());
    printf("%zu keys, end_bit %u, data mode %d\n", n, end_bit, mode);
    const bool only_msd = argc > 4;
    constexpr auto MATCH = rocprim::block_radix_rank_algorithm::match;
    constexpr auto MEMO = rocprim::block_radix_rank_algorithm::basic_memoize;
    {
        uint64_t *tmp_keys;
        CHECK(hipMalloc(&tmp_keys, n * 8));
        run_msd(in, out, tmp_keys, n, end_bit, d_bad);
        unsigned long long *acc, h[2];
        CHECK(hipMalloc(&acc, 16));
        CHECK(hipMemset(acc, 0, 16));
        hipLaunchKernelGGL(k_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, in, (uint64_t)n, acc);
        hipLaunchKernelGGL(k_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, out, (uint64_t)n, acc + 1);
        CHECK(hipMemcpy(h, acc, 16, hipMemcpyDeviceToHost));
        printf("  multiset checksum in %016llx out %016llx %s\n", h[0], h[1], h[0] == h[1] ? "equal" : "DIFFERENT");
        CHECK(hipFree(tmp_keys));
    }
    run<rocprim::default_config>("default", in, out, n, end_bit, 8, d_bad);
    if (only_msd) return 0;
    run<Cfg<512, 12, 8>>("512x12  8 bits match", in, out, n, end_bit, 8, d_bad);
    run<Cfg<1024, 8, 8>>("1024x8  8 bits match", in, out, n, end_bit, 8, d_bad);
    run<Cfg<512, 16, 8>>("512x16  8 bits match", in, out, n, end_bit, 8, d_bad);
    run<Cfg<256, 16, 8>>("256x16  8 bits match", in, out, n, end_bit, 8, d_bad);
    run<Cfg<512, 12, 9>>("512x12  9 bits match", in, out, n, end_bit, 9, d_bad);
    run<Cfg<256, 16, 9>>("256x16  9 bits match", in, out, n, end_bit, 9, d_bad);
    run<Cfg<256, 24, 9>>("256x24  9 bits match", in, out, n, end_bit, 9, d_bad);
    run<Cfg<512, 12, 10>>("512x12 10 bits match", in, out, n, end_bit, 10, d_bad);
    run<Cfg<256, 16, 10>>("256x16 10 bits match", in, out, n, end_bit, 10, d_bad);
    run<Cfg<256, 24, 10>>("256x24 10 bits match", in, out, n, end_bit, 10, d_bad);
    return 0;
}
