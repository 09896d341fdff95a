// Micro-probe (measurement tool, not product): bare random-row gather rate of gfx950 against the row width —
// 16*LPR bytes per row, LPR lanes x 16 B per row, 4 independent rows per lane group (the n = 4 BIGSI lookup), 50 M rows.
// Usage: gather_probe_wide [n_groups_M]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LPR>
__global__ __launch_bounds__(256) void probe(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g = t / LPR;
    const uint32_t c = (uint32_t)(t % LPR);
    if (g >= n_groups) return;
    uint4 v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = *reinterpret_cast<const uint4 *>(tab + (uint64_t)idx[g * 4 + s] * (16 * LPR) + c * 16);
    uint4 a = v[0];
#pragma unroll
    for (int s = 1; s < 4; ++s) { a.x &= v[s].x; a.y &= v[s].y; a.z &= v[s].z; a.w &= v[s].w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) out[0] = 1;
}
__global__ void fill_idx(uint32_t *idx, uint64_t n, uint32_t n_rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    idx[i] = (uint32_t)(z % n_rows);
}
template <int LPR>
void run(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t threads = n_groups * LPR;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<LPR><<<(unsigned)((threads + 255) / 256), 256>>>(tab, idx, n_groups, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) probe<LPR><<<(unsigned)((threads + 255) / 256), 256>>>(tab, idx, n_groups, out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double rows = 4.0 * n_groups, lines = rows * (16 * LPR > 128 ? 16 * LPR / 128 : 1);
    printf("row %4d B (LPR %2d): %.2f ms  %.1f G rows/s  %.1f G lines/s  %.2f TB/s of row bytes  %.2f TB/s of lines\n", 16 * LPR, LPR, ms,
           rows / ms / 1e6, lines / ms / 1e6, rows * 16 * LPR / ms / 1e9, lines * 128 / ms / 1e9);
}
int main(int argc, char **argv) {
    const uint64_t n_groups = (argc > 1 ? atoll(argv[1]) : 120) * 1000000ull;
    const uint32_t n_rows = 50000000;
    uint8_t *tab; uint32_t *idx, *out;
    CK(hipMalloc(&tab, (size_t)n_rows * 256)); CK(hipMemset(tab, 0x5A, (size_t)n_rows * 256));
    CK(hipMalloc(&idx, n_groups * 4 * 4)); CK(hipMalloc(&out, 4));
    fill_idx<<<(unsigned)((n_groups * 4 + 255) / 256), 256>>>(idx, n_groups * 4, n_rows);
    CK(hipDeviceSynchronize());
    run<2>(tab, idx, n_groups, out); run<4>(tab, idx, n_groups, out); run<8>(tab, idx, n_groups, out); run<16>(tab, idx, n_groups, out);
    return 0;
}
