// see quade_pool.h
#include "quade_pool.h"

#include <stdlib.h>

#include <map>
#include <mutex>

#include "../../include/quade_hip.h"

namespace {
struct Pool {
    std::mutex m;
    std::multimap<size_t, void*> free_[16];  // per device: capacity -> allocation
    size_t held = 0;
    size_t limit = [] {
        const char* e = getenv("QUADE_POOL_GB");
        const double gb = e && *e ? atof(e) : 64.0;
        return gb <= 0 ? (size_t)0 : (size_t)(gb * 1073741824.0);
    }();
};
struct PinnedPool {
    std::mutex m;
    std::multimap<size_t, void*> free_;
    size_t held = 0;
    size_t limit = [] {
        const char* e = getenv("QUADE_POOL_PINNED_GB");
        const double gb = e && *e ? atof(e) : 4.0;
        return gb <= 0 ? (size_t)0 : (size_t)(gb * 1073741824.0);
    }();
};
PinnedPool& pinned() {
    static PinnedPool* p = new PinnedPool();
    return *p;
}
Pool& pool() {
    static Pool* p = new Pool();  // (never destroyed: the runtime may be gone before static destructors run)
    return *p;
}
}  // namespace

hipError_t qd_pool_get(size_t want, void** p, size_t* cap) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Pool& P = pool();
    if (dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> g(P.m);
        auto it = P.free_[dev].lower_bound(want);
        // (an allocation much larger than asked for stays where it is: the request it was made for will come again)
        if (it != P.free_[dev].end() && it->first <= want + want / 2 + (1u << 20)) {
            *p = it->second;
            *cap = it->first;
            P.held -= it->first;
            P.free_[dev].erase(it);
            return hipSuccess;
        }
    }
    e = hipMalloc(p, want);
    if (e != hipSuccess && dev >= 0 && dev < 16) {  // out of memory with allocations idle in the list: give them back and try again
        {
            std::lock_guard<std::mutex> g(P.m);
            for (auto& kv : P.free_[dev]) (void)hipFree(kv.second), P.held -= kv.first;
            P.free_[dev].clear();
        }
        (void)hipGetLastError();
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess) *cap = want;
    return e;
}

void qd_pool_put(void* p, size_t cap) {
    if (!p) return;
    (void)hipDeviceSynchronize();
    int dev = -1;
    Pool& P = pool();
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && cap >= (1u << 20)) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device == dev) {  // (released on the device it lives on: the usual case)
            std::lock_guard<std::mutex> g(P.m);
            if (P.held + cap <= P.limit) {
                P.free_[dev].emplace(cap, p);
                P.held += cap;
                return;
            }
        }
    }
    (void)hipFree(p);
}

hipError_t qd_pool_get_pinned(size_t want, void** p, size_t* cap) {
    PinnedPool& P = pinned();
    {
        std::lock_guard<std::mutex> g(P.m);
        auto it = P.free_.lower_bound(want);
        if (it != P.free_.end() && it->first <= want + want / 2 + (1u << 20)) {
            *p = it->second;
            *cap = it->first;
            P.held -= it->first;
            P.free_.erase(it);
            return hipSuccess;
        }
    }
    const hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

void qd_pool_put_pinned(void* p, size_t cap) {
    if (!p) return;
    PinnedPool& P = pinned();
    if (cap >= (1u << 20)) {
        std::lock_guard<std::mutex> g(P.m);
        if (P.held + cap <= P.limit) {
            P.free_.emplace(cap, p);
            P.held += cap;
            return;
        }
    }
    (void)hipHostFree(p);
}

extern "C" int qd_pool_trim(void) {
    {
        PinnedPool& H = pinned();
        std::lock_guard<std::mutex> g(H.m);
        for (auto& kv : H.free_) (void)hipHostFree(kv.second);
        H.free_.clear();
        H.held = 0;
    }
    Pool& P = pool();
    std::lock_guard<std::mutex> g(P.m);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 16; ++d) {
        if (P.free_[d].empty()) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
        for (auto& kv : P.free_[d]) (void)hipFree(kv.second);
        P.free_[d].clear();
    }
    P.held = 0;
    (void)hipSetDevice(cur);
    return QD_OK;
}
