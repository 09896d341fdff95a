// How fast does this host feed the GPU?  H2D of 2 GiB from pageable memory, from pinned memory, and from pageable memory staged
// through two pinned buffers by T copying threads (the shape a pipelined host-pointer entry point would have).  Measurement tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using Clk = std::chrono::steady_clock;
static double ms(Clk::time_point a, Clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

static void par_memcpy(char *dst, const char *src, size_t n, int T) {
    std::vector<std::thread> th;
    const size_t per = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= n) break;
        th.emplace_back([=] { memcpy(dst + o, src + o, n - o < per ? n - o : per); });
    }
    for (auto &t : th) t.join();
}

int main() {
    const size_t N = 2ull << 30, CH = 64ull << 20;
    char *d; CK(hipMalloc(&d, N));
    char *pageable = (char *)malloc(N); memset(pageable, 1, N);
    char *pinned; CK(hipHostMalloc(&pinned, N)); memset(pinned, 2, N);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = Clk::now(); CK(hipMemcpy(d, pageable, N, hipMemcpyHostToDevice)); auto t1 = Clk::now();
        printf("pageable hipMemcpy: %.1f ms = %.1f GB/s\n", ms(t0, t1), N / ms(t0, t1) / 1e6);
        t0 = Clk::now(); CK(hipMemcpy(d, pinned, N, hipMemcpyHostToDevice)); t1 = Clk::now();
        printf("pinned   hipMemcpy: %.1f ms = %.1f GB/s\n", ms(t0, t1), N / ms(t0, t1) / 1e6);
    }
    char *stage[2]; CK(hipHostMalloc(&stage[0], CH)); CK(hipHostMalloc(&stage[1], CH));
    hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
    for (int T : {1, 2, 4, 8, 16}) {
        auto t0 = Clk::now();
        size_t i = 0;
        for (size_t o = 0; o < N; o += CH, ++i) {
            const int b = (int)(i & 1);
            if (i >= 2) CK(hipEventSynchronize(ev[b]));
            par_memcpy(stage[b], pageable + o, CH, T);
            CK(hipMemcpyAsync(d + o, stage[b], CH, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[b], s));
        }
        CK(hipStreamSynchronize(s));
        auto t1 = Clk::now();
        printf("staged through 2 x 64 MiB pinned, %2d copy threads: %.1f ms = %.1f GB/s\n", T, ms(t0, t1), N / ms(t0, t1) / 1e6);
    }
    // D2H
    auto t0 = Clk::now(); CK(hipMemcpy(pageable, d, N, hipMemcpyDeviceToHost)); auto t1 = Clk::now();
    printf("D2H pageable: %.1f GB/s\n", N / ms(t0, t1) / 1e6);
    t0 = Clk::now(); CK(hipMemcpy(pinned, d, N, hipMemcpyDeviceToHost)); t1 = Clk::now();
    printf("D2H pinned:   %.1f GB/s\n", N / ms(t0, t1) / 1e6);
    return 0;
}
