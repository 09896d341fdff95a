// Scratch helpers shared by the translation units that run rocPRIM primitives on a ctx's stream (cid_kmerset.hip, cid_reports.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/colorid_hip.h"
#include "cid_internal.hpp"

namespace {

using cid::fail;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(CID_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
struct DevBuf {  // scoped device scratch from the ctx's block cache (hipMalloc of GiB-sized blocks costs tens of ms on this platform)
    cid_ctx *c;
    T *p = nullptr;
    explicit DevBuf(cid_ctx *ctx) : c(ctx) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) cid::ctx_free(c, p); }
    int alloc(size_t n) {
        void *q = nullptr;
        const int rc = cid::ctx_alloc(c, (n ? n : 1) * sizeof(T), &q);
        p = static_cast<T *>(q);
        return rc;
    }
    T *release() { T *q = p; p = nullptr; return q; }   // the new owner returns it with ctx_free
};

unsigned grid_for_n(uint64_t n) { return (unsigned)((n + 255) / 256); }

struct SatAdd {   // u32 multiplicities saturate instead of wrapping
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { const uint32_t s = a + b; return s < a ? 0xFFFFFFFFu : s; }
};

}  // namespace
