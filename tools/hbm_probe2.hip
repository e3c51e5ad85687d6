// HBM ceiling probes, second set (measurement tool): copy vs read-only vs the 4-stream demux
// pattern, with plain and non-temporal loads, persistent vs one-tile-per-workgroup grids.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

template <bool NT>
__device__ __forceinline__ ulong2 ld(const ulong2* p) {
    if (NT) {
        ulong2 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    }
    return *p;
}

// mode 0: copy (read 16 B, write 16 B per lane-item); mode 1: read-only (one store per thread at the
// end); mode 2: read 4 streams, write 4 B per lane-item (the demux shape)
template <int MODE, bool NT, int UNITS>
__global__ __launch_bounds__(256) void k(const ulong2* a, const ulong2* b, const ulong2* c, const ulong2* d,
                                         ulong2* out, int64_t n_vec) {
    const int64_t tile = 256 * UNITS;
    const int64_t ntiles = (n_vec + tile - 1) / tile;
    u64 acc = 0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        ulong2 v[UNITS][MODE == 2 ? 4 : 1];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t i = t * tile + u * 256 + threadIdx.x;
            if (i < n_vec) {
                v[u][0] = ld<NT>(a + i);
                if (MODE == 2) {
                    v[u][1] = ld<NT>(b + i);
                    v[u][2] = ld<NT>(c + i);
                    v[u][3] = ld<NT>(d + i);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t i = t * tile + u * 256 + threadIdx.x;
            if (i < n_vec) {
                if (MODE == 0) out[i] = v[u][0];
                if (MODE == 1) acc ^= v[u][0].x + v[u][0].y;
                if (MODE == 2) {
                    u64 x = v[u][0].x ^ v[u][1].y ^ (v[u][2].x * 3) ^ v[u][3].y ^ v[u][0].y ^ v[u][1].x ^ v[u][2].y ^ v[u][3].x;
                    reinterpret_cast<uint32_t*>(out)[i] = (uint32_t)x ^ (uint32_t)(x >> 32);
                }
            }
        }
    }
    if (MODE == 1) reinterpret_cast<u64*>(out)[blockIdx.x * 256 + threadIdx.x] = acc;
}

extern "C" int probe2(int mode, int nt, int units, int grid, const void* a, const void* b, const void* c,
                      const void* d, void* out, int64_t n_vec, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(M, N, U)                                                                                     \
    if (mode == M && nt == N && units == U) {                                                           \
        hipLaunchKernelGGL((k<M, (bool)N, U>), dim3(grid), dim3(256), 0, st, (const ulong2*)a,          \
                           (const ulong2*)b, (const ulong2*)c, (const ulong2*)d, (ulong2*)out, n_vec);  \
        return (int)hipGetLastError();                                                                  \
    }
    GO(0, 0, 1) GO(0, 0, 4) GO(0, 1, 4) GO(1, 0, 1) GO(1, 0, 4) GO(1, 1, 4) GO(1, 0, 8) GO(2, 0, 1) GO(2, 0, 2) GO(2, 1, 2) GO(2, 1, 1)
    return -1;
}
