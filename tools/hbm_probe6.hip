// HBM probe 6 (measurement tool): cache-policy bits on the 4-byte output store of the demux shape
// (4 nt-read streams of 16 B per lane): plain, nt, sc1 (write-through), sc0 sc1, sc0 sc1 nt.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned long v2u64 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 5) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const v2u64* a, const v2u64* b, const v2u64* c, const v2u64* d,
                                           uint32_t* out, int64_t n_vec) {
    const int64_t ntiles = n_vec / BLOCK;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t i = t * BLOCK + threadIdx.x;
        v2u64 v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(b + i);
        v2u64 v2 = __builtin_nontemporal_load(c + i), v3 = __builtin_nontemporal_load(d + i);
        unsigned long x = v0.x ^ v1.y ^ (v2.x * 3) ^ v3.y ^ v0.y ^ v1.x ^ v2.y ^ v3.x;
        st<MODE>(out + i, (uint32_t)x ^ (uint32_t)(x >> 32));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int probe6(int mode, int block, int grid, const void* a, const void* b, const void* c, const void* d,
                      void* out, int64_t n_vec, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define GO(M, B)                                                                                             \
    if (mode == M && block == B) {                                                                           \
        hipLaunchKernelGGL((k<M, B>), dim3(grid), dim3(B), 0, s, (const v2u64*)a, (const v2u64*)b, (const v2u64*)c, \
                           (const v2u64*)d, (uint32_t*)out, n_vec);                                          \
        return (int)hipGetLastError();                                                                       \
    }
    GO(0, 256) GO(1, 256) GO(2, 256) GO(3, 256) GO(4, 256) GO(5, 256) GO(0, 512) GO(1, 512) GO(2, 512) GO(3, 512) GO(4, 512) GO(5, 512)
    return -1;
}
