// How long do hipMalloc / hipFree of GB-sized blocks take on this box?  (decides whether the sort-based read_id path
// should keep its scratch between calls)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    using C = std::chrono::steady_clock;
    hipFree(nullptr);
    for (size_t gb : {1, 1, 4, 4}) {
        void *p = nullptr;
        auto t0 = C::now();
        hipError_t e = hipMalloc(&p, gb << 30);
        auto t1 = C::now();
        hipMemset(p, 0, 64);
        hipDeviceSynchronize();
        auto t2 = C::now();
        hipFree(p);
        auto t3 = C::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        printf("%zu GiB: hipMalloc %.2f ms (%s), first touch %.2f ms, hipFree %.2f ms\n", gb, ms(t0, t1), hipGetErrorString(e), ms(t1, t2), ms(t2, t3));
    }
    return 0;
}
