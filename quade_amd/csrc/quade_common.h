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

// ---- wide keys (16 < K <= 32 bytes) on the 16-byte machinery ---------------------------------------------
// Barcodes are words over {A,C,G,T,N} (src/Sample.py:40,141) and those five letters differ in their low
// nibble (1, 3, 7, 4, 0xE): a K-byte barcode packs into K nibbles = at most 16 bytes, injectively.  A read's
// folded slice is packed the same way and looked up in the table of packed barcodes; a hit names the only
// barcode the read can equal, and a byte compare against that one barcode decides (a read byte outside the
// alphabet shares its nibble with some letter -- 'Q' with 'A' -- so the packed hit alone proves nothing).
// low nibbles of 8 packed bytes -> 32 bits (byte i -> bits 4i .. 4i+3)
QD_HD uint32_t qd_pack_nib8(uint64_t x) {
    x &= 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = x | (x >> 16);
    return (uint32_t)x;
}
// packed key of the two index-read slices (each <= 16 folded bytes as lo/hi words, zero beyond its width;
// w1 = bytes of the first slice): the first slice's nibbles, then the second's
QD_HD void qd_wide_key(uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, int w1, uint64_t* lo, uint64_t* hi) {
    const uint64_t n1 = (uint64_t)qd_pack_nib8(a_lo) | ((uint64_t)qd_pack_nib8(a_hi) << 32);
    const uint64_t n2 = (uint64_t)qd_pack_nib8(b_lo) | ((uint64_t)qd_pack_nib8(b_hi) << 32);
    if (w1 <= 0) {
        *lo = n2;
        *hi = 0;
    } else if (w1 >= 16) {
        *lo = n1;
        *hi = n2;
    } else {
        *lo = n1 | (n2 << (4 * w1));
        *hi = n2 >> (64 - 4 * w1);
    }
}
// hash of a packed key as probe_lds() computes it for K > 8: both words, the byte length K in the seed
QD_HD uint32_t qd_hash_wide(uint64_t lo, uint64_t hi, uint32_t K, uint32_t seed) {
    return qd_hash_fini(qd_hash_step(qd_hash_step(qd_hash_init(K, seed), lo), hi));
}
