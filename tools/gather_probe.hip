// Micro-probe (measurement tool, not product): how fast can gfx950 gather random 32-byte rows out of a table far
// larger than the Infinity Cache, and how many HBM bytes does each load flavour pull per row?
// Usage: gather_probe <variant> <table_MiB> <n_rows_to_read_M> ; run under rocprofv3 --pmc for the byte counters.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct U4 { uint32_t x, y, z, w; };

template <int V> __device__ __forceinline__ U4 ld16(const void *p) {
    U4 r;
    if constexpr (V == 0) { const uint4 v = *reinterpret_cast<const uint4 *>(p); r = U4{v.x, v.y, v.z, v.w}; }
    else if constexpr (V == 1) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p)); r = U4{v.x, v.y, v.z, v.w};
    } else {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 v;
        if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if constexpr (V == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        r = U4{v.x, v.y, v.z, v.w};
    }
    return r;
}

// 2 lanes per 32-byte row, 4 independent rows per lane pair (like the n = 4 BIGSI lookup)
template <int V>
__global__ __launch_bounds__(256) void probe2(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g = t >> 1;
    if (g >= n_groups) return;
    const uint32_t half = t & 1;
    U4 a{~0u, ~0u, ~0u, ~0u};
    if constexpr (V <= 1) {
        U4 v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = ld16<V>(tab + (uint64_t)idx[g * 4 + s] * 32 + half * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) { a.x &= v[s].x; a.y &= v[s].y; a.z &= v[s].z; a.w &= v[s].w; }
    } else {  // asm flavours: issue 4 loads then one wait
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 v0, v1, v2, v3;
        const void *p0 = tab + (uint64_t)idx[g * 4 + 0] * 32 + half * 16;
        const void *p1 = tab + (uint64_t)idx[g * 4 + 1] * 32 + half * 16;
        const void *p2 = tab + (uint64_t)idx[g * 4 + 2] * 32 + half * 16;
        const void *p3 = tab + (uint64_t)idx[g * 4 + 3] * 32 + half * 16;
        if constexpr (V == 2) asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        if constexpr (V == 3) asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        if constexpr (V == 4) asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1 nt\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1 nt\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1 nt\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        if constexpr (V == 5) asm volatile("global_load_dwordx4 %0, %4, off sc0\n\tglobal_load_dwordx4 %1, %5, off sc0\n\tglobal_load_dwordx4 %2, %6, off sc0\n\tglobal_load_dwordx4 %3, %7, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        a.x = v0.x & v1.x & v2.x & v3.x; a.y = v0.y & v1.y & v2.y & v3.y;
        a.z = v0.z & v1.z & v2.z & v3.z; a.w = v0.w & v1.w & v2.w & v3.w;
    }
    const uint32_t r = a.x ^ a.y ^ a.z ^ a.w;
    if (r == 0x12345678u) out[t & 1023] = r;  // keep the loads alive, (almost) never taken
}

// 1 lane per row (2 x 16 B), 4 rows per lane
__global__ __launch_bounds__(256) void probe1(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    uint4 v[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 *p = reinterpret_cast<const uint4 *>(tab + (uint64_t)idx[g * 4 + s] * 32);
        v[2 * s] = p[0]; v[2 * s + 1] = p[1];
    }
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) r ^= v[s].x & v[s].y & v[s].z & v[s].w;
    if (r == 0x12345678u) out[g & 1023] = r;
}

// scalar path: every wave walks its 64 groups with s_load_dwordx8 (32-byte row per scalar load, 64-byte K$ lines)
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void probe_s(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t g = t < n_groups ? t : n_groups - 1;
    const uint4 my = *reinterpret_cast<const uint4 *>(idx + g * 4);
    uint32_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < 64; ++i) {
        const uint32_t r0 = __builtin_amdgcn_readlane(my.x, i), r1 = __builtin_amdgcn_readlane(my.y, i);
        const uint32_t r2 = __builtin_amdgcn_readlane(my.z, i), r3 = __builtin_amdgcn_readlane(my.w, i);
        typedef const __attribute__((address_space(4))) u32x8 *cptr;
        const u32x8 a = *(cptr)(tab + (uint64_t)r0 * 32), b = *(cptr)(tab + (uint64_t)r1 * 32);
        const u32x8 c = *(cptr)(tab + (uint64_t)r2 * 32), d = *(cptr)(tab + (uint64_t)r3 * 32);
        const u32x8 v = a & b & c & d;
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
    }
    if (acc == 0x12345678u) out[t & 1023] = acc;
}

// MIXED gather inside one wave (the shape a mixed k_search_count would have): a wave owns 64 groups; NS of each group's 4 rows come
// through the scalar cache (s_load_dwordx8 = 64-byte lines, 8 in flight per batch, handed to the owning lane pair with
// v_writelane), the other 4 - NS through the vector path (lane pair per row, 16 B per lane, 2 sub-passes of 32 groups).
// v_writelane with an SGPR value needs its lane select in M0 (or an inline constant) on gfx9: one SGPR on the constant bus
#define WL(acc, val, ln) asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(acc) : "s"(val), "s"(ln) : "m0")
template <int NS, int BATCH = 8, int SUBS = 2>
__global__ __launch_bounds__(256) void probe_mix(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t g0 = wave * 64;
    if (g0 >= n_groups) return;
    const uint64_t gl = g0 + lane < n_groups ? g0 + lane : n_groups - 1;
    const uint4 my = *reinterpret_cast<const uint4 *>(idx + gl * 4);      // lane l holds the 4 row numbers of group g0 + l
    typedef const __attribute__((address_space(4))) u32x8 *cptr;
    uint32_t r = 0;
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
        const int kk = sub * 32 + (lane >> 1);                             // the group this lane pair works on
        const uint32_t half = lane & 1;
        const bool use_scalar = sub < SUBS;       // wave-uniform
        uint4 v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s >= 4 - NS && use_scalar) { v[s] = uint4{~0u, ~0u, ~0u, ~0u}; continue; }
            const uint32_t src = s == 0 ? my.x : s == 1 ? my.y : s == 2 ? my.z : my.w;
            const uint32_t row = (uint32_t)__shfl((int)src, kk, 64);
            v[s] = *reinterpret_cast<const uint4 *>(tab + (uint64_t)row * 32 + half * 16);
        }
        uint32_t sc0 = ~0u, sc1 = ~0u, sc2 = ~0u, sc3 = ~0u;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!use_scalar) break;
            const uint32_t srcv = s == 0 ? my.w : my.z;
            uint32_t t0 = ~0u, t1 = ~0u, t2 = ~0u, t3 = ~0u;      // this lane's 16 bytes of its group's row
#pragma unroll 1
            for (int b = 0; b < 32; b += BATCH) {
                u32x8 q[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const uint32_t row = __builtin_amdgcn_readlane(srcv, sub * 32 + b + u);
                    q[u] = *(cptr)(tab + (uint64_t)row * 32);
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int l0 = 2 * (b + u);
                    WL(t0, q[u][0], l0); WL(t0, q[u][4], l0 + 1);
                    WL(t1, q[u][1], l0); WL(t1, q[u][5], l0 + 1);
                    WL(t2, q[u][2], l0); WL(t2, q[u][6], l0 + 1);
                    WL(t3, q[u][3], l0); WL(t3, q[u][7], l0 + 1);
                }
            }
            sc0 &= t0; sc1 &= t1; sc2 &= t2; sc3 &= t3;
        }
        uint4 a{sc0, sc1, sc2, sc3};
#pragma unroll
        for (int s = 0; s < 4; ++s) { a.x &= v[s].x; a.y &= v[s].y; a.z &= v[s].z; a.w &= v[s].w; }
        r ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (r == 0x12345678u) out[threadIdx.x & 1023] = r;
}

// scalar PREFETCH: one s_load_dword per row pulls (only) the row's 64-byte half line into L2; mode 0 = prefetch only,
// mode 1 = prefetch this wave's 256 rows, then gather them with the vector path (do they now hit in L2?)
template <int MODE>
__global__ __launch_bounds__(256) void probe_pf(const uint8_t *tab, const uint32_t *idx, uint64_t n_groups, uint32_t *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one lane per group of 4 rows
    const uint64_t g = t < n_groups ? t : n_groups - 1;
    const uint4 my = *reinterpret_cast<const uint4 *>(idx + g * 4);
    typedef const __attribute__((address_space(4))) uint32_t *cptr;
    uint32_t acc = 0;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) {
        const uint32_t r0 = __builtin_amdgcn_readlane(my.x, i), r1 = __builtin_amdgcn_readlane(my.y, i);
        const uint32_t r2 = __builtin_amdgcn_readlane(my.z, i), r3 = __builtin_amdgcn_readlane(my.w, i);
        acc ^= *(cptr)(tab + (uint64_t)r0 * 32) ^ *(cptr)(tab + (uint64_t)r1 * 32) ^ *(cptr)(tab + (uint64_t)r2 * 32) ^ *(cptr)(tab + (uint64_t)r3 * 32);
    }
    uint32_t r = acc;
    if (MODE == 1) {
        // vector gather of the same rows: lane pair (2j, 2j+1) takes group j's... here simply 1 lane per row, 32 B
        uint4 v[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t row = s == 0 ? my.x : s == 1 ? my.y : s == 2 ? my.z : my.w;
            const uint4 *p = reinterpret_cast<const uint4 *>(tab + (uint64_t)row * 32);
            v[2 * s] = p[0]; v[2 * s + 1] = p[1];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) r ^= v[s].x & v[s].y & v[s].z & v[s].w;
    }
    if (r == 0x12345678u) out[t & 1023] = r;
}

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const uint64_t tab_mib = argc > 2 ? strtoull(argv[2], 0, 10) : 1526;
    const uint64_t n_groups = (argc > 3 ? strtoull(argv[3], 0, 10) : 120) * 1000000ull;
    const int alloc_mode = argc > 4 ? atoi(argv[4]) : 0;  // 0 hipMalloc, 1 fine-grained, 2 uncached
    const uint64_t tab_bytes = tab_mib << 20, n_rows = tab_bytes / 32;
    uint8_t *tab; uint32_t *idx, *out;
    if (alloc_mode == 0) CK(hipMalloc(&tab, tab_bytes));
    else CK(hipExtMallocWithFlags((void **)&tab, tab_bytes, alloc_mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    CK(hipMemset(tab, 0x5A, tab_bytes));
    CK(hipMalloc(&idx, n_groups * 4 * 4)); CK(hipMalloc(&out, 4096));
    std::vector<uint32_t> h(n_groups * 4);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)((s >> 11) % n_rows); }
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (variant == 10) {  // vector gather (3/4 of the groups) and scalar gather (1/4) concurrently on two streams
        hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
        const int sfrac = argc > 5 ? atoi(argv[5]) : 4;
        const uint64_t gs = n_groups / sfrac, gv = n_groups - gs;
        for (int mode = 0; mode < 3; ++mode) {  // 0 both, 1 vector part alone, 2 scalar part alone
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                if (mode != 2) hipLaunchKernelGGL(probe2<0>, dim3((unsigned)((gv * 2 + 255) / 256)), dim3(256), 0, sa, tab, idx, gv, out);
                if (mode != 1) hipLaunchKernelGGL(probe_s, dim3((unsigned)((gs + 255) / 256)), dim3(256), 0, sb, tab, idx + gv * 4, gs, out);
                CK(hipDeviceSynchronize());
                const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms < best) best = ms;
            }
            printf("hybrid 1/%d scalar, mode %d: %.3f ms (vector %llu M rows, scalar %llu M rows)\n", sfrac, mode, best,
                   (unsigned long long)(gv * 4 / 1000000), (unsigned long long)(gs * 4 / 1000000));
        }
        return 0;
    }
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        if (variant == 11) hipLaunchKernelGGL(probe_pf<0>, dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 12) hipLaunchKernelGGL(probe_pf<1>, dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 13) hipLaunchKernelGGL((probe_mix<1, 8, 2>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 14) hipLaunchKernelGGL((probe_mix<2, 8, 2>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 15) hipLaunchKernelGGL((probe_mix<1, 4, 2>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 16) hipLaunchKernelGGL((probe_mix<1, 8, 1>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 17) hipLaunchKernelGGL((probe_mix<1, 16, 2>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 18) hipLaunchKernelGGL((probe_mix<2, 8, 1>), dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 8) hipLaunchKernelGGL(probe_s, dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else if (variant == 9) hipLaunchKernelGGL(probe1, dim3((n_groups + 255) / 256), dim3(256), 0, 0, tab, idx, n_groups, out);
        else {
            const unsigned grid = (unsigned)((n_groups * 2 + 255) / 256);
            switch (variant) {
            case 0: hipLaunchKernelGGL(probe2<0>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            case 1: hipLaunchKernelGGL(probe2<1>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            case 2: hipLaunchKernelGGL(probe2<2>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            case 3: hipLaunchKernelGGL(probe2<3>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            case 4: hipLaunchKernelGGL(probe2<4>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            case 5: hipLaunchKernelGGL(probe2<5>, dim3(grid), dim3(256), 0, 0, tab, idx, n_groups, out); break;
            default: printf("bad variant\n"); return 1;
            }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double rows = (double)n_groups * 4;
    printf("variant %d alloc %d table %llu MiB: %.3f ms, %.2f G rows/s, %.2f TB/s of 32-B rows (+idx %.2f TB/s)\n", variant, alloc_mode,
           (unsigned long long)tab_mib, best, rows / best / 1e6, rows * 32 / best / 1e9, rows * 36 / best / 1e9);
    return 0;
}
