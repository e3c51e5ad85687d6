// HBM probe 7 (measurement tool): does ONE interleaved input stream beat FOUR parallel ones?
// Same bytes as the dual 8+8 kernel (32 B in + 2 B out per pair, a lane = 2 pairs, nt loads, sc1 dword store,
// tiles strided over an oversubscribed grid), no matching.  layout 0: four arrays, lane i reads 16 B of each;
// layout 1: one array of the same total size, four consecutive 1 KiB-per-wave loads per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d,
                                           uint32_t* out, int64_t n_units) {
    const int64_t ntiles = n_units / BLOCK;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t i = t * BLOCK + threadIdx.x;
        v4u32 v0, v1, v2, v3;
        if (LAYOUT == 0) {
            v0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(a + i * 16));
            v1 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(b + i * 16));
            v2 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(c + i * 16));
            v3 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(d + i * 16));
        } else {
            const uint8_t* base = a + t * (int64_t)BLOCK * 64 + threadIdx.x * 16;
            v0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(base));
            v1 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(base + BLOCK * 16));
            v2 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(base + BLOCK * 32));
            v3 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(base + BLOCK * 48));
        }
        const v4u32 x = v0 ^ v1 ^ v2 ^ v3;
        __hip_atomic_store(out + i, x.x ^ x.y ^ x.z ^ x.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int probe7(int layout, int block, int grid, const void* a, const void* b, const void* c, const void* d, void* out,
                      int64_t n_units, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(L, B)                                                                                                  \
    if (layout == L && block == B) {                                                                              \
        hipLaunchKernelGGL((k<L, B>), dim3(grid), dim3(B), 0, st, (const uint8_t*)a, (const uint8_t*)b,          \
                           (const uint8_t*)c, (const uint8_t*)d, (uint32_t*)out, n_units);                        \
        return (int)hipGetLastError();                                                                            \
    }
    GO(0, 256) GO(1, 256) GO(0, 512) GO(1, 512)
    return -1;
}
