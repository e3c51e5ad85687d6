// CRC-32 (gzip's) of text that a workgroup holds in LDS: used where the text is staged anyway -- the BGZF inflater checks every
// block it made against the block's trailer before the text leaves (quade_inflate.hip), the coder takes the CRC of a sub-block
// while it stages it (quade_deflate.hip) -- so the bytes are not read from HBM again for it.
//
// Shape: the text is cut from its END into slices of SW words, one per lane (SW odd: lane l's k-th word lies in bank
// (SW * l + k) mod 32, so a wave's reads spread over the banks); a lane runs slice-by-4 over its slice out of four 256-entry
// tables in LDS, multiplies its CRC by x^(8 * bytes behind the slice) mod P -- two table factors, the slices being equal -- and
// the lanes' terms are XOR-ed: crc(A || B) = crc(A) * x^(8 |B|) + crc(B) over GF(2).
#pragma once
#include <stdint.h>

#include "text_rules.h"

namespace qdcrc {

struct Tables {
    uint32_t t[4][256];
};
constexpr Tables make_tables() {
    Tables T{};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        T.t[0][i] = c;
    }
    for (int k = 1; k < 4; ++k)
        for (uint32_t i = 0; i < 256; ++i) T.t[k][i] = (T.t[k - 1][i] >> 8) ^ T.t[0][T.t[k - 1][i] & 0xFFu];
    return T;
}
constexpr uint32_t mulmod_c(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
constexpr uint32_t xpow8_c(uint32_t n) {  // x^(8 n) mod P
    uint32_t p = 0x80000000u, v = 0x00800000u;
    for (; n; n >>= 1) {
        if (n & 1u) p = mulmod_c(v, p);
        v = mulmod_c(v, v);
    }
    return p;
}
// x^(8 * SW * 4 * j) = lo[j & 31] * hi[j >> 5]; tail[k] = x^(8 k), k < 4
template <int SW>
struct Shifts {
    uint32_t lo[32], hi[32], tail[4];
};
template <int SW>
constexpr Shifts<SW> make_shifts() {
    Shifts<SW> S{};
    for (uint32_t j = 0; j < 32; ++j) {
        S.lo[j] = xpow8_c((uint32_t)SW * 4u * j);
        S.hi[j] = xpow8_c((uint32_t)SW * 4u * 32u * j);
    }
    for (uint32_t k = 0; k < 4; ++k) S.tail[k] = xpow8_c(k);
    return S;
}

}  // namespace qdcrc

#if defined(__HIPCC__)
namespace qdcrc {

__device__ const Tables g_tables = make_tables();
template <int SW>
__device__ const Shifts<SW> g_shifts = make_shifts<SW>();

// the four tables -> LDS (1 024 words); a barrier before the first use is the caller's
template <int NT>
__device__ __forceinline__ void stage_tables(uint32_t* lds_t) {
    const uint32_t* src = &g_tables.t[0][0];
    for (uint32_t i = threadIdx.x; i < 1024; i += NT) lds_t[i] = src[i];
}

// CRC-32 of the len bytes at `words` (LDS, 4-byte aligned); NT lanes, slices of SW words: NT * SW * 4 >= len.  `part`: NT / 64
// words of LDS.  Every lane of the workgroup calls it (two barriers inside); the result is the same in every lane.
template <int NT, int SW>
__device__ __forceinline__ uint32_t crc32_lds(const uint32_t* words, uint32_t len, const uint32_t* lds_t, uint32_t* part) {
    static_assert(SW % 2 == 1, "odd slices spread a wave's reads over the LDS banks");
    const Shifts<SW>& shifts = g_shifts<SW>;
    const uint32_t tid = threadIdx.x;
    const uint32_t main_words = len >> 2, tail = len & 3u;
    // slice j = the SW words that end j slices before the end of the whole words; lane tid takes slice tid
    const uint32_t end_w = main_words > tid * (uint32_t)SW ? main_words - tid * (uint32_t)SW : 0u;
    const uint32_t beg_w = end_w > (uint32_t)SW ? end_w - (uint32_t)SW : 0u;
    uint32_t term = 0;
    if (end_w > beg_w || (tid == 0 && tail)) {
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t w = beg_w; w < end_w; ++w) {
            c ^= words[w];
            c = lds_t[768 + (c & 0xFFu)] ^ lds_t[512 + ((c >> 8) & 0xFFu)] ^ lds_t[256 + ((c >> 16) & 0xFFu)] ^ lds_t[c >> 24];
        }
        if (tid == 0 && tail) {  // the bytes behind the last whole word belong to the last slice
            const uint32_t w = words[main_words];
            for (uint32_t k = 0; k < tail; ++k) c = (c >> 8) ^ lds_t[(c ^ (w >> (8 * k))) & 0xFFu];
        }
        c = ~c;
        if (tid == 0) {
            term = c;
        } else {
            term = qd_crc_mulmod(qd_crc_mulmod(c, shifts.lo[tid & 31u]), shifts.hi[tid >> 5]);
            if (tail) term = qd_crc_mulmod(term, shifts.tail[tail]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) term ^= __shfl_xor(term, d, 64);
    if ((tid & 63u) == 0) part[tid >> 6] = term;
    __syncthreads();
    uint32_t crc = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) crc ^= part[i];
    __syncthreads();
    return crc;
}

}  // namespace qdcrc
#endif
