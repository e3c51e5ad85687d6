// HBM probe 4 (measurement tool): how does the cost of the output stream scale with its size?
// 4 nt-read streams (16 B per lane each); a dword is stored for one tile out of every STRIDE tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned long v2u64 __attribute__((ext_vector_type(2)));

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const v2u64* a, const v2u64* b, const v2u64* c, const v2u64* d,
                                           uint32_t* out, int64_t n_vec, int stride, int wide) {
    const int64_t ntiles = (n_vec + BLOCK - 1) / BLOCK;
    u64 acc = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t i = t * BLOCK + threadIdx.x;
        if (i < n_vec) {
            v2u64 v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(b + i);
            v2u64 v2 = __builtin_nontemporal_load(c + i), v3 = __builtin_nontemporal_load(d + i);
            u64 x = v0.x ^ v1.y ^ (v2.x * 3) ^ v3.y ^ v0.y ^ v1.x ^ v2.y ^ v3.x;
            acc ^= x;
            if (stride > 0 && (t % stride) == 0) {
                if (wide == 0) out[i] = (uint32_t)x ^ (uint32_t)(x >> 32);
                else if (wide == 1) reinterpret_cast<u64*>(out)[i] = x;
                else reinterpret_cast<v2u64*>(out)[i] = v0 ^ v1;
            }
        }
    }
    if (stride == 0) out[(int64_t)blockIdx.x * BLOCK + threadIdx.x] = (uint32_t)acc;
}

extern "C" int probe4(int block, int grid, const void* a, const void* b, const void* c, const void* d, void* out,
                      int64_t n_vec, int stride, int wide, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (block == 256) hipLaunchKernelGGL((k<256>), dim3(grid), dim3(256), 0, st, (const v2u64*)a, (const v2u64*)b, (const v2u64*)c, (const v2u64*)d, (uint32_t*)out, n_vec, stride, wide);
    else hipLaunchKernelGGL((k<512>), dim3(grid), dim3(512), 0, st, (const v2u64*)a, (const v2u64*)b, (const v2u64*)c, (const v2u64*)d, (uint32_t*)out, n_vec, stride, wide);
    return (int)hipGetLastError();
}
