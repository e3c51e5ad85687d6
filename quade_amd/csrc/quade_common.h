// Shared host/device definitions of the demux library: canonical key form, case fold, hash.
// The host builds the barcode table with exactly the functions the kernels probe it with.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define QD_HD __host__ __device__ inline
#else
#define QD_HD inline
#endif

#define QD_EMPTY_SLOT 0xFFFFFFFFu
#define QD_KEY_WORDS 4 /* QD_MAX_KEY / 8 */

// ASCII case fold of 8 packed bytes: a-z -> A-Z, every other byte (incl. >= 0x80) untouched.
// Restates `index.seq.upper()` (src/Sample.py:65,67) for byte strings: only bytes that fold to
// A,C,G,T,N can ever equal a registered barcode (alphabet: src/Sample.py:40,141).
QD_HD uint64_t qd_fold8(uint64_t x) {
    const uint64_t L = 0x0101010101010101ull;
    uint64_t x7 = x & (0x7Full * L);
    uint64_t ge_a = x7 + (0x80 - 0x61) * L;   // bit7 set where (b & 0x7f) >= 'a'
    uint64_t gt_z = x7 + (0x80 - 0x7B) * L;   // bit7 set where (b & 0x7f) >  'z'
    uint64_t lower = ge_a & ~gt_z & ~x & (0x80ull * L);
    return x ^ (lower >> 2);                  // 0x80 >> 2 == 0x20
}

// 1 when every byte of x is >= thr (1 <= thr <= 0x7F); bytes >= 0x80 (the 0xFF padding) pass.
// Restates `min(index.qual) >= MIN_QUAL` (src/Sample.py:70) on Phred+33 text: thr = MIN_QUAL + 33.
QD_HD uint32_t qd_all_ge8(uint64_t x, uint32_t thr) {
    const uint64_t L = 0x0101010101010101ull;
    uint64_t t = (x & (0x7Full * L)) + (uint64_t)(0x80 - thr) * L;  // bit7 where (b & 0x7f) >= thr
    return (((x | t) & (0x80ull * L)) == (0x80ull * L)) ? 1u : 0u;
}

// Hash of a canonical key: `len` bytes, little-endian packed into ceil(len/8) words, zero padded.
QD_HD uint32_t qd_hash_step(uint32_t h, uint64_t w) {
    h = (h ^ (uint32_t)w) * 0x85EBCA77u;
    h = (h ^ (h >> 13) ^ (uint32_t)(w >> 32)) * 0xC2B2AE3Du;
    return h;
}
QD_HD uint32_t qd_hash_init(uint32_t len, uint32_t seed) { return (len + seed) * 0x9E3779B1u + 0x7F4A7C15u; }
QD_HD uint32_t qd_hash_fini(uint32_t h) {
    h ^= h >> 16;
    h *= 0x27D4EB2Fu;
    h ^= h >> 15;
    return h;
}
QD_HD uint32_t qd_hash_key(const uint64_t* w, uint32_t len, uint32_t seed) {
    uint32_t h = qd_hash_init(len, seed);
    uint32_t nw = (len + 7u) >> 3;
    for (uint32_t i = 0; i < nw; ++i) h = qd_hash_step(h, w[i]);
    return qd_hash_fini(h);
}
// slot entry = (fingerprint16 << 16) | sample ordinal; the fingerprint is the hash's top half.
QD_HD uint32_t qd_slot_entry(uint32_t h, uint32_t ordinal) { return (h & 0xFFFF0000u) | ordinal; }
