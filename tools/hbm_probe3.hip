// HBM probes, third set (measurement tool): the demux read shape (4 streams, 16 B per lane) with
// different ways of writing the 2-bytes-per-pair output.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned long long u64;

__device__ __forceinline__ ulong2 ldnt(const ulong2* p) {
    ulong2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
}

// STORE: 0 none (one store per thread at the end), 1 = 4 B plain, 2 = 4 B nt, 3 = 16 B plain per 4 units, 4 = 16 B nt
template <int STORE, int UNITS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const ulong2* a, const ulong2* b, const ulong2* c, const ulong2* d,
                                           uint32_t* out, int64_t n_vec) {
    const int64_t tile = (int64_t)BLOCK * UNITS;
    const int64_t ntiles = (n_vec + tile - 1) / tile;
    u64 acc = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        ulong2 v[UNITS][4];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t i = t * tile + (int64_t)u * BLOCK + threadIdx.x;
            if (i < n_vec) {
                v[u][0] = ldnt(a + i);
                v[u][1] = ldnt(b + i);
                v[u][2] = ldnt(c + i);
                v[u][3] = ldnt(d + i);
            }
        }
        uint32_t r[UNITS];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            u64 x = v[u][0].x ^ v[u][1].y ^ (v[u][2].x * 3) ^ v[u][3].y ^ v[u][0].y ^ v[u][1].x ^ v[u][2].y ^ v[u][3].x;
            r[u] = (uint32_t)x ^ (uint32_t)(x >> 32);
        }
        if (STORE == 0) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) acc ^= r[u];
        } else if (STORE == 1 || STORE == 2) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                const int64_t i = t * tile + (int64_t)u * BLOCK + threadIdx.x;
                if (i < n_vec) {
                    if (STORE == 1) out[i] = r[u];
                    else __builtin_nontemporal_store(r[u], out + i);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < UNITS; u += 4) {
                const int64_t i = (t * tile + (int64_t)u * BLOCK) / 4 + threadIdx.x;  // 16-byte units
                if (i * 4 < n_vec) {
                    uint4 w = make_uint4(r[u], r[(u + 1) % UNITS], r[(u + 2) % UNITS], r[(u + 3) % UNITS]);
                    if (STORE == 3) reinterpret_cast<uint4*>(out)[i] = w;
                    else {
                        u64* o = reinterpret_cast<u64*>(out) + 2 * i;
                        __builtin_nontemporal_store((u64)w.x | ((u64)w.y << 32), o);
                        __builtin_nontemporal_store((u64)w.z | ((u64)w.w << 32), o + 1);
                    }
                }
            }
        }
    }
    if (STORE == 0) out[(int64_t)blockIdx.x * BLOCK + threadIdx.x] = (uint32_t)acc;
}

extern "C" int probe3(int store, int units, int block, int grid, const void* a, const void* b, const void* c,
                      const void* d, void* out, int64_t n_vec, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(S, U, B)                                                                                     \
    if (store == S && units == U && block == B) {                                                       \
        hipLaunchKernelGGL((k<S, U, B>), dim3(grid), dim3(B), 0, st, (const ulong2*)a, (const ulong2*)b, \
                           (const ulong2*)c, (const ulong2*)d, (uint32_t*)out, n_vec);                  \
        return (int)hipGetLastError();                                                                  \
    }
    GO(0, 1, 256) GO(0, 2, 256) GO(0, 4, 256) GO(1, 1, 256) GO(1, 2, 256) GO(1, 4, 256) GO(2, 1, 256) GO(2, 2, 256) GO(2, 4, 256)
    GO(3, 4, 256) GO(4, 4, 256) GO(0, 2, 512) GO(1, 2, 512) GO(2, 2, 512) GO(3, 4, 512) GO(4, 4, 512) GO(0, 1, 512) GO(1, 1, 512) GO(2, 1, 512)
    return -1;
}
