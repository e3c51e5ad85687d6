// HBM probe 9 (measurement tool; VERDICT r02 #2: "probe an interleaved codes|mol 14-B output record -- one write stream
// instead of two -- with an hbm_probe variant before touching the ABI").  Bytes of the cfg4 kernel (dual 8 bp + 6 bp molecular
// index: per lane = 2 pairs two 14-byte seq rows as two 16-byte loads, the second end-aligned, and two 8-byte qual rows as one,
// for both index reads; out 2 x 2 B of codes + 2 x 12 B of molecular bytes), wave runs of 4 steps, no matching.
//  mode 0: two output streams, as shipped: the run's codes (1 KiB per wave) and its molecular bytes (6 KiB per wave) leave
//          through the wave's LDS strips as 16-byte write-through stores
//  mode 1: ONE output stream of 14-byte records (codes | molecular bytes per pair): 7 KiB per wave and run, same stores
//  mode 2: no output (what the reads alone cost)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u32a __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ v4u32 ldnt(const uint8_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4u32a*>(p)); }
__device__ __forceinline__ void st16(void* p, v4u32 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t* s1, const uint8_t* q1, const uint8_t* s2, const uint8_t* q2, uint8_t* codes,
                                           uint8_t* mol, uint8_t* rec, int64_t n_units) {
    constexpr int RUN = 4;
    __shared__ __attribute__((aligned(16))) uint32_t strips[BLOCK / 64][RUN * 64 * 7];  // 28 bytes per lane and step
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int64_t nsuper = n_units / (BLOCK * RUN);
    uint32_t* mine = strips[wave];
    for (int64_t s = blockIdx.x; s < nsuper; s += gridDim.x) {
        const int64_t w0 = (s * (BLOCK / 64) + wave) * (int64_t)(RUN * 64);  // first unit (= 2 pairs) of the wave's run
#pragma unroll
        for (int step = 0; step < RUN; ++step) {
            const int64_t i = w0 + step * 64 + lane;
            const v4u32 x = ldnt(s1 + i * 28) ^ ldnt(s1 + i * 28 + 12) ^ ldnt(q1 + i * 16) ^ ldnt(s2 + i * 28) ^ ldnt(s2 + i * 28 + 12) ^
                            ldnt(q2 + i * 16);
            const uint32_t r = x.x ^ x.y ^ x.z ^ x.w;
            if (MODE == 0) {  // codes: dword [step*64 + lane]; molecular bytes: 6 dwords behind the codes' 256
                mine[step * 64 + lane] = r;
#pragma unroll
                for (int j = 0; j < 6; ++j) mine[RUN * 64 + (step * 64 + lane) * 6 + j] = x[j & 3] + j;
            } else if (MODE == 1) {  // 7 dwords per lane: 2 records of 14 bytes
                mine[(step * 64 + lane) * 7] = r;
#pragma unroll
                for (int j = 0; j < 6; ++j) mine[(step * 64 + lane) * 7 + 1 + j] = x[j & 3] + j;
            } else if (r == 0x12345678u) {
                codes[i] = 1;
            }
        }
        if (MODE == 2) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (MODE == 0) {
            st16(codes + (w0 + 4 * lane) * 4, *reinterpret_cast<const v4u32*>(&mine[4 * lane]));
#pragma unroll
            for (int r = 0; r < 6; ++r)  // 6 KiB = 384 pieces of 16 bytes
                st16(mol + w0 * 24 + (r * 64 + lane) * 16, *reinterpret_cast<const v4u32*>(&mine[RUN * 64 + (r * 64 + lane) * 4]));
        } else {
#pragma unroll
            for (int r = 0; r < 7; ++r)  // 7 KiB = 448 pieces
                st16(rec + w0 * 28 + (r * 64 + lane) * 16, *reinterpret_cast<const v4u32*>(&mine[(r * 64 + lane) * 4]));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int probe9(int mode, int block, int grid, const void* s1, const void* q1, const void* s2, const void* q2, void* codes, void* mol,
                      void* rec, int64_t n_units, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(M, B)                                                                                                                  \
    if (mode == M && block == B) {                                                                                                \
        hipLaunchKernelGGL((k<M, B>), dim3(grid), dim3(B), 0, st, (const uint8_t*)s1, (const uint8_t*)q1, (const uint8_t*)s2,     \
                           (const uint8_t*)q2, (uint8_t*)codes, (uint8_t*)mol, (uint8_t*)rec, n_units);                          \
        return (int)hipGetLastError();                                                                                            \
    }
    GO(0, 512) GO(1, 512) GO(2, 512) GO(0, 256) GO(1, 256) GO(2, 256)
    return -1;
}
