/*
 * STRONG CPU BASELINE ("baseline B" of BASELINE.md) -- MEASUREMENT / TEST INFRASTRUCTURE ONLY.
 * Never linked into the product and never a fallback: only bench.py's cpu_baseline leg and the tests
 * load it.  A tuned host version of the hot path for the 8-byte-row configs (single 8 bp / dual 8+8 bp
 * index, barcode slices = the whole rows): SWAR case fold and quality gate on 64-bit words, an
 * open-addressing hash table over the barcodes, pthreads over contiguous shares of the batch.
 *
 * Restates (reference file:line), on the packed rows the HIP library takes (include/quade_hip.h):
 *   fused key = I1 row (+ I2 row) ......................... src/Quade.py:217, 246
 *   case fold for the lookup only ......................... src/Sample.py:65
 *   exact whole-string match, ordinal = section order ..... src/Sample.py:65-67, src/Quade.py:133
 *   min phred >= MIN_QUAL over the barcode slice .......... src/Sample.py:70
 *   counters ............................................... src/Sample.py:62,71-72,79-80,88
 * Pinned to the scalar C oracle and, through it, to the Python oracle (tests/test_oracle_c.py).
 * Deliberately self-contained (own fold / gate / hash): it shares no code with the GPU library.
 *
 * Build: gcc -O3 -pthread -shared -fPIC -o oracle/libstrong_demux.so oracle/strong_demux.c
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define L8 0x0101010101010101ull

static inline uint64_t fold8(uint64_t x) { /* a-z -> A-Z on 8 packed bytes, bytes >= 0x80 untouched */
    const uint64_t x7 = x & (0x7Full * L8);
    const uint64_t ge_a = x7 + (0x80 - 0x61) * L8, gt_z = x7 + (0x80 - 0x7B) * L8;
    return x ^ ((ge_a & ~gt_z & ~x & (0x80ull * L8)) >> 2);
}
static inline int all_ge8(uint64_t x, uint32_t thr) { /* every byte >= thr; bytes >= 0x80 pass */
    const uint64_t t = (x & (0x7Full * L8)) + (uint64_t)(0x80 - thr) * L8;
    return ((x | t) & (0x80ull * L8)) == (0x80ull * L8);
}
static inline uint32_t mix(uint64_t a, uint64_t b) {
    uint64_t h = (a ^ (b * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    return (uint32_t)(h ^ (h >> 29));
}

typedef struct {
    const uint64_t *s1, *s2, *q1, *q2; /* rows as 64-bit words (stride 8) */
    const uint64_t* key;               /* [S][2] */
    const uint32_t* slots;
    uint32_t mask, thr, S;
    int dual;
    int64_t lo, hi;
    uint16_t* codes;
    uint64_t* hist; /* [2S+1] of this thread */
} job_t;

static void* work(void* arg) {
    job_t* j = (job_t*)arg;
    for (int64_t i = j->lo; i < j->hi; ++i) {
        const uint64_t klo = fold8(j->s1[i]), khi = j->dual ? fold8(j->s2[i]) : 0;
        uint32_t s = mix(klo, khi) & j->mask, code = 0xFFFF;
        for (;;) {
            const uint32_t e = j->slots[s];
            if (e == 0xFFFFFFFFu) break;
            if (j->key[2 * e] == klo && j->key[2 * e + 1] == khi) {
                const int pass = all_ge8(j->q1[i], j->thr) & (j->dual ? all_ge8(j->q2[i], j->thr) : 1);
                code = e * 2 + (uint32_t)(pass ^ 1);
                break;
            }
            s = (s + 1) & j->mask;
        }
        j->codes[i] = (uint16_t)code;
        j->hist[code == 0xFFFF ? 2 * j->S : code] += 1;
    }
    return 0;
}

/* rows: seq/qual of stride 8, barcode slice = columns 0..7 of each index read; barcodes: S keys of 8
 * (single) or 16 (dual) upper-case bytes, concatenated.  counts: uint64[2S+4] as include/quade_hip.h.
 * Returns 0, or -1 for a shape it does not cover. */
int strong_demux_rows8(int dual, int32_t min_qual, int32_t S, const uint8_t* barcodes, int64_t n, const uint8_t* seq1,
                       const uint8_t* qual1, const uint8_t* seq2, const uint8_t* qual2, int32_t threads, uint16_t* codes,
                       uint64_t* counts) {
    if (S < 0 || S > 32767 || n < 0 || threads < 1 || threads > 1024) return -1;
    const int K = dual ? 16 : 8;
    uint32_t m = 16;
    while (m < 4u * (uint32_t)S) m <<= 1;
    uint64_t* key = (uint64_t*)calloc((size_t)(S > 0 ? S : 1) * 2, 8);
    uint32_t* slots = (uint32_t*)malloc((size_t)m * 4);
    memset(slots, 0xFF, (size_t)m * 4);
    for (int32_t i = 0; i < S; ++i) {
        memcpy(&key[2 * i], barcodes + (size_t)i * K, (size_t)K);
        uint32_t s = mix(key[2 * i], key[2 * i + 1]) & (m - 1);
        while (slots[s] != 0xFFFFFFFFu) s = (s + 1) & (m - 1);
        slots[s] = (uint32_t)i;
    }
    const size_t nb = (size_t)2 * S + 1;
    uint64_t* hist = (uint64_t*)calloc(nb * (size_t)threads, 8);
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) {
        job_t j = {(const uint64_t*)seq1, (const uint64_t*)seq2, (const uint64_t*)qual1, (const uint64_t*)qual2, key, slots,
                   m - 1, (uint32_t)(min_qual + 33), (uint32_t)S, dual, n * t / threads, n * (t + 1) / threads, codes,
                   hist + nb * (size_t)t};
        jobs[t] = j;
        if (threads == 1)
            work(&jobs[t]);
        else
            pthread_create(&th[t], 0, work, &jobs[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
    memset(counts, 0, ((size_t)2 * S + 4) * 8);
    for (int t = 0; t < threads; ++t) {
        const uint64_t* h = hist + nb * (size_t)t;
        for (int32_t i = 0; i < S; ++i) {
            counts[4 + 2 * i] += h[2 * i];
            counts[5 + 2 * i] += h[2 * i + 1];
            counts[1] += h[2 * i];
            counts[2] += h[2 * i + 1];
        }
        counts[3] += h[2 * S];
    }
    counts[0] = (uint64_t)n;
    free(key); free(slots); free(hist); free(jobs); free(th);
    return 0;
}
