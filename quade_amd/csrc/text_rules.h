// Per-record text rules shared by the device kernels (quade_text.hip) and their host-side unit tests: what the path consumes
// of pyFastq's records, restated once.  Plain functions over bytes, no I/O.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define QD_HD __host__ __device__ __forceinline__
#else
#define QD_HD inline
#endif

// Python's bytes.split() blanks: space, \t \n \v \f \r
QD_HD bool qd_is_blank(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// name of a record = header line without its first byte, first blank-delimited token (src/FastqWriter.py:61-66 appends the
// tag to it; the bundled goldens pin the cut).  head .. line_end: the header line without its newline.
QD_HD void qd_name_of(const uint8_t* text, uint32_t head, uint32_t line_end, uint32_t* name_off, uint32_t* name_len) {
    uint32_t h = head + (line_end > head ? 1u : 0u);
    while (h < line_end && qd_is_blank(text[h])) ++h;
    uint32_t e = h;
    while (e < line_end && !qd_is_blank(text[e])) ++e;
    *name_off = h;
    *name_len = e - h;
}

// bytes of read[start:end] for a read of `len` bases (Python slice clamping, src/Quade.py:217-218)
QD_HD uint32_t qd_slice_len(int32_t start, int32_t end, uint32_t len) {
    const uint32_t e = (uint32_t)end < len ? (uint32_t)end : len;
    return e > (uint32_t)start ? e - (uint32_t)start : 0u;
}

// CRC-32 of the gzip trailer, reflected polynomial 0xEDB88320: a * b mod P with x^0 = 0x80000000
QD_HD uint32_t qd_crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// x^(8 * 2^k) mod P for k = 0 .. 31 into t[]
QD_HD void qd_crc_pow_table(uint32_t* t) {
    uint32_t v = 0x00800000u;  // x^8
    for (int k = 0; k < 32; ++k) {
        t[k] = v;
        v = qd_crc_mulmod(v, v);
    }
}
// x^(8 n) mod P
QD_HD uint32_t qd_crc_xpow8(const uint32_t* t, uint32_t n) {
    uint32_t p = 0x80000000u;
    for (int k = 0; n; ++k, n >>= 1)
        if (n & 1u) p = qd_crc_mulmod(t[k], p);
    return p;
}
