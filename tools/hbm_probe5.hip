// HBM probe 5 (measurement tool): cfg4-shaped traffic with 16-byte padded rows (32 B per lane per
// seq stream, aligned dwordx4 x2) versus exact 14-byte rows (28 B per lane, 4-byte aligned
// dwordx4 + dwordx3).  Output: 4 B codes + 24 B molecular per lane in both cases.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned long v2u64 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned int v3u32 __attribute__((ext_vector_type(3)));

template <bool EXACT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t* s1, const uint8_t* q1, const uint8_t* s2, const uint8_t* q2,
                                           uint32_t* codes, uint8_t* mol, int64_t n_units) {
    // unit = 2 pairs handled by one lane
    const int64_t ntiles = n_units / BLOCK;  // full tiles only
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t i = t * BLOCK + threadIdx.x;
        v4u32 a0, a1, b0, b1;
        if (EXACT) {
            const uint8_t* pa = s1 + i * 28;
            const uint8_t* pb = s2 + i * 28;
            a0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(pa));
            v3u32 t3 = __builtin_nontemporal_load(reinterpret_cast<const v3u32*>(pa + 16));
            a1 = v4u32{t3.x, t3.y, t3.z, 0};
            b0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(pb));
            v3u32 u3 = __builtin_nontemporal_load(reinterpret_cast<const v3u32*>(pb + 16));
            b1 = v4u32{u3.x, u3.y, u3.z, 0};
        } else {
            a0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(s1 + i * 32));
            a1 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(s1 + i * 32 + 16));
            b0 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(s2 + i * 32));
            b1 = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(s2 + i * 32 + 16));
        }
        v4u32 c = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(q1 + i * 16));
        v4u32 d = __builtin_nontemporal_load(reinterpret_cast<const v4u32*>(q2 + i * 16));
        v4u32 x = a0 ^ a1 ^ b0 ^ b1 ^ c ^ d;
        codes[i] = x.x ^ x.y ^ x.z ^ x.w;
        unsigned long* m = reinterpret_cast<unsigned long*>(mol + i * 24);
        m[0] = ((unsigned long)a0.x << 32) | b0.y;
        m[1] = ((unsigned long)a1.x << 32) | b1.y;
        m[2] = ((unsigned long)c.x << 32) | d.y;
    }
}

extern "C" int probe5(int exact, int block, int grid, const void* s1, const void* q1, const void* s2, const void* q2,
                      void* codes, void* mol, int64_t n_units, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define GO(E, B)                                                                                              \
    if (exact == E && block == B) {                                                                           \
        hipLaunchKernelGGL((k<(bool)E, B>), dim3(grid), dim3(B), 0, st, (const uint8_t*)s1, (const uint8_t*)q1, \
                           (const uint8_t*)s2, (const uint8_t*)q2, (uint32_t*)codes, (uint8_t*)mol, n_units); \
        return (int)hipGetLastError();                                                                        \
    }
    GO(0, 256) GO(1, 256) GO(0, 512) GO(1, 512)
    return -1;
}
