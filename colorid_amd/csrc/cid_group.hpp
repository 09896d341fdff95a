// Shared by the translation units that implement cid_group (cid_group.hip: replicated index; cid_group_stripes.hip: colour stripes).
#pragma once
#include "../../include/colorid_hip.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cid_objects.hpp"

namespace cidg {

using cid::fail;

#define CIDG_HIP_TRY(expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return cid::fail(CID_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the handful of RCCL entry points used (rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclUint64 = 5, ncclSum = 0)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && AllReduce && AllGather && GroupStart && GroupEnd && GetErrorString;
    }
};
constexpr int kNcclUint32 = 3, kNcclUint64 = 5, kNcclSum = 0;

}  // namespace cidg

namespace cidg {
// One persistent host thread per rank (beyond rank 0, which runs on the caller's thread): a group call used to start and join N
// threads, and a thread's start and end map and unmap its stack — calls that queue on the address-space lock of a process whose
// other threads fault pages (the CLI's readers and packers); per batch of a read_id run that cost more than the call's own work.
class RankThreads {
  public:
    ~RankThreads() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_go_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // body(r) for r = 0 .. n-1; rank 0 on the calling thread
    void run(int n, const std::function<void(int)> &body) {
        if (n <= 1) { if (n == 1) body(0); return; }
        std::unique_lock<std::mutex> lk(mu_);
        while ((int)threads_.size() < n - 1) {
            const int r = (int)threads_.size() + 1;
            threads_.emplace_back([this, r] { loop(r); });
            seen_.push_back(gen_);
        }
        body_ = &body; n_ = n; left_ = n - 1; ++gen_;
        lk.unlock();
        cv_go_.notify_all();
        body(0);
        lk.lock();
        cv_done_.wait(lk, [&] { return left_ == 0; });
        body_ = nullptr;
    }
  private:
    void loop(int r) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_go_.wait(lk, [&] { return stop_ || seen_[(size_t)r - 1] != gen_; });
            if (stop_) return;
            seen_[(size_t)r - 1] = gen_;
            if (r >= n_ || !body_) continue;   // (a call over fewer ranks than there are threads)
            const std::function<void(int)> *b = body_;
            lk.unlock();
            (*b)(r);
            lk.lock();
            if (--left_ == 0) cv_done_.notify_one();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    std::vector<std::thread> threads_;
    std::vector<uint64_t> seen_;
    const std::function<void(int)> *body_ = nullptr;
    uint64_t gen_ = 0;
    int n_ = 0, left_ = 0;
    bool stop_ = false;
};
}  // namespace cidg

struct cid_group {
    cidg::RankThreads rank_threads;
    std::vector<cid_ctx *> ctx;
    std::vector<int> dev;
    bool use_rccl = false;
    cidg::Rccl rccl;
    std::vector<void *> comms;
    // per rank: "my buffer is ready" / "my slice is reduced" events of the striped reductions (cid_group_stripes.hip), made on first use
    std::vector<hipEvent_t> ev_ready, ev_reduced;
    // sparse read_id results of the last cid_group_readid_count_sparse (per rank: rows and entries)
    std::vector<uint64_t> sp_rows, sp_entries;
    // colour stripes (cid_group_stripes_*): first colour of every rank's stripe + the total, of the last striped read_id call
    bool sp_striped = false;
    std::vector<uint32_t> sp_base;
};

namespace cidg {


// contiguous, balanced partition (the same rule as colorid_amd/dist.py shard_bounds): sizes differ by at most one
inline void shard_bounds(size_t n_units, int rank, int world, size_t *lo, size_t *hi) {
    const size_t base = n_units / (size_t)world, rem = n_units % (size_t)world;
    *lo = (size_t)rank * base + ((size_t)rank < rem ? (size_t)rank : rem);
    *hi = *lo + base + ((size_t)rank < rem ? 1 : 0);
}

// the same with every boundary on a multiple of 64 units (byte-string k-mers: a shard's first k-mer must sit on a 16-byte boundary)
inline void shard_bounds64(size_t n_units, int rank, int world, size_t *lo, size_t *hi) {
    const size_t blocks = (n_units + 63) / 64;
    shard_bounds(blocks, rank, world, lo, hi);
    *lo *= 64; *hi *= 64;
    if (*lo > n_units) *lo = n_units;
    if (*hi > n_units) *hi = n_units;
}

// run fn(rank) on one host thread per rank (a cid_ctx is used by one thread at a time); returns the first failure, whose
// message is re-recorded on the calling thread (cid_last_error is thread-local)
template <typename F>
int for_each_rank(cid_group *g, F &&fn) {
    const int n = (int)g->ctx.size();
    std::vector<int> rc(n, CID_OK);
    std::vector<std::string> msg(n);
    auto body = [&](int r) {
        rc[r] = fn(r);
        if (rc[r] != CID_OK) msg[r] = cid_last_error();
    };
    g->rank_threads.run(n, body);
    for (int r = 0; r < n; ++r)
        if (rc[r] != CID_OK) return fail(rc[r], "rank %d (device %d): %s", r, g->dev[r], msg[r].c_str());
    return CID_OK;
}

inline int check_replicas(const cid_group *g, cid_index *const *replicas) {
    if (!g || !replicas) return fail(CID_ERR_INVALID, "null group/replicas");
    for (size_t r = 0; r < g->ctx.size(); ++r) {
        const int rc = cid::check_ready(g->ctx[r], replicas[r]);
        if (rc) return rc;
        if (replicas[r]->n_colors != replicas[0]->n_colors || replicas[r]->k != replicas[0]->k || replicas[r]->m != replicas[0]->m ||
            replicas[r]->n_hash != replicas[0]->n_hash)
            return fail(CID_ERR_INVALID, "replica %zu differs from replica 0 in shape", r);
    }
    return CID_OK;
}

// sum of one u64[count] (elem_bytes 8) or u32[count] (4) device array per rank, every rank ends with the total; RCCL on the
// ranks' ctx streams, or through the host (synchronous).  Touches no file descriptor: what RCCL may print (its version banner,
// NCCL_DEBUG output) is printed at communicator creation, which a host that needs a clean stdout wraps itself (host/main.cpp)
int allreduce_sum(cid_group *g, void *const *d_bufs, size_t count, int elem_bytes);
// the colour-striped half of cid_group_readid_sparse_fetch (cid_group_stripes.hip)
int stripes_sparse_fetch(cid_group *g, uint64_t *row_start, uint32_t *colours, uint32_t *counts);

}  // namespace cidg
