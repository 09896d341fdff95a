// How fast are LDS atomics on gfx950?  One workgroup of 1024 threads per CU, every thread N operations on pseudo-random slots of a
// 32 768-slot table: plain write, plain read, atomicMin without return, atomicCAS without a use of its result, atomicCAS with the
// result used (ds_cmpst_rtn), atomicAdd with return.  Prints cycles per wave-instruction (s_memtime around the loop, wave 0).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_atomic_probe tools/lds_atomic_probe.hip && tools/bin/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kSlots = 32768, kN = 64;
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, unsigned *sink, unsigned mask) {
    __shared__ unsigned table[kSlots];
    for (int i = threadIdx.x; i < kSlots; i += 1024) table[i] = 0xFFFFFFFFu;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < kN; ++i) {
        x = x * 1664525u + 1013904223u;
        const unsigned at = (x >> 10) & mask;
        if (MODE == 0) table[at] = x;
        if (MODE == 1) acc += table[at];
        if (MODE == 2) atomicMin(&table[at], x);
        if (MODE == 3) acc += atomicCAS(&table[at], 0xFFFFFFFFu, x);
        if (MODE == 4) acc += atomicAdd(&table[at], 1u);
        if (MODE == 5) acc += atomicMin(&table[at], x);
        if (MODE == 6) { const unsigned c = __hip_atomic_load(&table[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (c == 0xFFFFFFFFu) acc += atomicCAS(&table[at], 0xFFFFFFFFu, x); }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345) sink[0] = acc;
}
template <int MODE>
void run(const char *name, unsigned mask) {
    unsigned long long *d; unsigned *s;
    hipMalloc(&d, 256 * 8); hipMalloc(&s, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, s, mask);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    // 16 waves x kN wave-instructions per workgroup
    printf("%-44s mask %5x: %8.0f cycles per workgroup, %6.1f cycles per wave-instruction (16 waves share the CU), %5.2f lanes/cycle/CU\n", name, mask, sum / 256, sum / 256 / (16.0 * kN),
           1024.0 * kN / (sum / 256));
    hipFree(d); hipFree(s);
}
int main() {
    for (unsigned mask : {0x7FFFu, 0x3Fu}) {
        run<0>("ds_write_b32", mask);
        run<1>("ds_read_b32", mask);
        run<2>("atomicMin, no return (ds_min_u32)", mask);
        run<3>("atomicCAS, result used (ds_cmpst_rtn_b32)", mask);
        run<4>("atomicAdd, result used (ds_add_rtn_u32)", mask);
        run<5>("atomicMin, result used (ds_min_rtn_u32)", mask);
        run<6>("load, then CAS where empty", mask);
    }
    return 0;
}
