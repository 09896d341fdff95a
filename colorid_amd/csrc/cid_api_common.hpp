// Shared by the translation units behind the C ABI (cid_api_ctx / _index / _search / _readid .hip); not part of the ABI.
#pragma once
#include "../../include/colorid_hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "cid_host_math.hpp"
#include "cid_internal.hpp"
#include "cid_objects.hpp"

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return cid::fail(CID_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace cid {

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Work per block: enough blocks to balance 256 CUs dynamically, few enough that the per-block flush of the
// LDS counters (<= 3*C global atomics) stays negligible.
inline uint32_t pick_tiles_per_block(const cid_ctx *c, uint64_t n_kmers) {
    const uint64_t n_tiles = (n_kmers + kWave - 1) / kWave;
    uint64_t tpb = n_tiles / ((uint64_t)c->n_cu * 32);
    if (tpb < 4) tpb = 4;
    if (tpb > 256) tpb = 256;
    return (uint32_t)tpb;
}

}  // namespace cid
