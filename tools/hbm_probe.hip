// HBM ceiling probe for the demux access pattern (measurement tool, not product code):
// reads the same four row arrays with the same 16-B-per-lane loads and writes one dword per two
// pairs, but does no matching.  Gives (a) the bandwidth ceiling of this access pattern and (b) a
// known byte count to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE against.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NSTREAM, int UNITS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe(const uint8_t* a, const uint8_t* b, const uint8_t* c,
                                               const uint8_t* d, uint32_t* out, int64_t n_vec) {
    // n_vec = number of 16-byte vectors per stream (= pairs / 2)
    const int64_t tile = (int64_t)BLOCK * UNITS;
    const int64_t ntiles = (n_vec + tile - 1) / tile;
    const uint8_t* s[4] = {a, b, c, d};
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        ulong2 v[UNITS][NSTREAM];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t i = t * tile + (int64_t)u * BLOCK + threadIdx.x;
            if (i < n_vec) {
#pragma unroll
                for (int k = 0; k < NSTREAM; ++k) v[u][k] = *reinterpret_cast<const ulong2*>(s[k] + i * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t i = t * tile + (int64_t)u * BLOCK + threadIdx.x;
            if (i < n_vec) {
                uint64_t x = 0;
#pragma unroll
                for (int k = 0; k < NSTREAM; ++k) x ^= v[u][k].x ^ (v[u][k].y * 3);
                out[i] = (uint32_t)x ^ (uint32_t)(x >> 32);
            }
        }
    }
}

extern "C" int probe_run(int nstream, int units, int block, int grid, const void* a, const void* b,
                         const void* c, const void* d, void* out, int64_t n_vec, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(NS, U, B)                                                                                  \
    if (nstream == NS && units == U && block == B) {                                                  \
        hipLaunchKernelGGL((probe<NS, U, B>), dim3(grid), dim3(B), 0, st, (const uint8_t*)a,          \
                           (const uint8_t*)b, (const uint8_t*)c, (const uint8_t*)d, (uint32_t*)out, n_vec); \
        return (int)hipGetLastError();                                                                \
    }
    GO(4, 1, 256) GO(4, 2, 256) GO(4, 4, 256) GO(4, 1, 512) GO(4, 2, 512) GO(4, 4, 512) GO(4, 1, 1024) GO(4, 2, 1024)
    GO(1, 1, 256) GO(1, 2, 256) GO(1, 4, 256) GO(1, 4, 512) GO(1, 8, 256) GO(2, 2, 256) GO(2, 4, 256)
    return -1;
}
