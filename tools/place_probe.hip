// Micro-probe (measurement tool, not product): does the rate of a random 32-byte-row gather depend on WHICH allocation the
// table sits in (tools/exp_alias3.py: identical indexes differ by 6 %), and can the placement be controlled?  The same gather
// (4 rows per lane pair, like the n = 4 BIGSI lookup; row numbers streamed from a 1.9 GB array) over tables obtained from
// hipMalloc (several, all kept alive) and from the virtual-memory API with the address range aligned to 2 MiB .. 1 GiB.
// Usage: place_probe [table_MiB] [n_groups_M]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gather4(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g = t >> 1;
    if (g >= n_groups) return;
    const uint32_t half = t & 1;
    uint4 v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = *reinterpret_cast<const uint4 *>(tab + (uint64_t)idx[g * 4 + s] * 32 + half * 16);
    uint32_t r = ~0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) r &= v[s].x & v[s].y & v[s].z & v[s].w;
    if (r == 0x12345678u) out[t & 1023] = r;
}
// the same gather + one 4-byte result per group.  W = 0: nontemporal 4-byte store per lane pair (a wave writes one 128-byte line
// per pass, as the search kernel writes its per-k-mer output); 1: the same with a plain store; 2 / 3: a wave works through 8
// passes (256 groups), parks the results in LDS and writes 1 KiB at once, 16 bytes per lane (nontemporal / plain).
template <int W>
__global__ __launch_bounds__(256) void gather4w(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *res) {
    __shared__ uint32_t park[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t half = lane & 1;
    if constexpr (W <= 1) {
        const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const uint64_t g = t >> 1;
        if (g >= n_groups) return;
        uint4 v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = *reinterpret_cast<const uint4 *>(tab + (uint64_t)idx[g * 4 + s] * 32 + half * 16);
        uint32_t r = ~0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) r &= v[s].x & v[s].y & v[s].z & v[s].w;
        r &= __shfl_xor(r, 1);
        if (!half) { if constexpr (W == 0) __builtin_nontemporal_store(r, &res[g]); else res[g] = r; }
    } else {
        const uint64_t g0 = ((uint64_t)blockIdx.x * 4 + wave) * 256;   // this wave's 256 groups
        if (g0 >= n_groups) return;
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const uint64_t g = g0 + i * 32 + (lane >> 1);
            uint32_t r = ~0u;
            if (g < n_groups) {
                uint4 v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = *reinterpret_cast<const uint4 *>(tab + (uint64_t)idx[g * 4 + s] * 32 + half * 16);
#pragma unroll
                for (int s = 0; s < 4; ++s) r &= v[s].x & v[s].y & v[s].z & v[s].w;
            }
            r &= __shfl_xor(r, 1);
            if (!half) park[wave][i * 32 + (lane >> 1)] = r;
        }
        __builtin_amdgcn_wave_barrier();
        const uint4 o = *reinterpret_cast<const uint4 *>(&park[wave][4 * lane]);
        if (g0 + 4 * lane + 3 < n_groups) {
            typedef uint32_t v4 __attribute__((ext_vector_type(4)));
            v4 ov = {o.x, o.y, o.z, o.w};
            if constexpr (W == 2) __builtin_nontemporal_store(ov, reinterpret_cast<v4 *>(&res[g0 + 4 * lane]));
            else *reinterpret_cast<v4 *>(&res[g0 + 4 * lane]) = ov;
        }
    }
}
template <int W>
static float run_w(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *res) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather4w<W>, dim3((unsigned)(W <= 1 ? (n_groups * 2 + 255) / 256 : (n_groups + 1023) / 1024)), dim3(256), 0, 0, tab, idx, n_groups, res);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

static float run(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather4, dim3((unsigned)((n_groups * 2 + 255) / 256)), dim3(256), 0, 0, tab, idx, n_groups, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char **argv) {
    const uint64_t tab_mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1526;
    const uint64_t n_groups = (argc > 2 ? strtoull(argv[2], 0, 10) : 120) * 1000000ull;
    const uint64_t tab_bytes = tab_mib << 20, n_rows = tab_bytes / 32;
    uint32_t *idx, *out;
    CK(hipMalloc(&idx, n_groups * 4 * 4)); CK(hipMalloc(&out, 4096));
    {
        std::vector<uint32_t> h(n_groups * 4);
        uint64_t s = 88172645463325252ull;
        for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)((s >> 11) % n_rows); }
        CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    const double rows = (double)n_groups * 4;
    std::vector<uint8_t *> tabs;
    std::vector<uint32_t *> ress;
    for (int i = 0; i < 8; ++i) {   // plain hipMalloc, all kept
        uint8_t *tab; CK(hipMalloc(&tab, tab_bytes)); CK(hipMemset(tab, 0x5A, tab_bytes));
        const float ms = run(tab, idx, n_groups, out);
        printf("hipMalloc #%d            %p: %.3f ms  %.2f G rows/s\n", i, (void *)tab, ms, rows / ms / 1e6);
        void *gap; CK(hipMalloc(&gap, (size_t)(37 + 11 * i) << 20));
        tabs.push_back(tab);
        uint32_t *res; CK(hipMalloc(&res, n_groups * 4)); ress.push_back(res);
    }
    for (int i = 0; i < 8; i += 4)
        for (int j = 0; j < 8; j += 2) {
            printf("gather + result write: table #%d, results in allocation #%d %p: nt 4 B %.3f ms | plain 4 B %.3f | 1 KiB nt %.3f | 1 KiB plain %.3f\n", i, j, (void *)ress[j],
                   run_w<0>(tabs[i], idx, n_groups, ress[j]), run_w<1>(tabs[i], idx, n_groups, ress[j]), run_w<2>(tabs[i], idx, n_groups, ress[j]), run_w<3>(tabs[i], idx, n_groups, ress[j]));
        }
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const size_t sz = (tab_bytes + grec - 1) / grec * grec;
    for (size_t align : {(size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30, (size_t)2 << 20, (size_t)1 << 30, (size_t)4 << 30}) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, sz, align, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, sz, &prop, 0));
        CK(hipMemMap(va, sz, 0, h, 0));
        hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, sz, &ad, 1));
        CK(hipMemset(va, 0x5A, tab_bytes));
        const float ms = run((const uint8_t *)va, idx, n_groups, out);
        printf("VMM aligned %5zu MiB    %p: %.3f ms  %.2f G rows/s\n", align >> 20, va, ms, rows / ms / 1e6);
    }
    return 0;
}
