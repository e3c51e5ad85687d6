/*
 * CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Scalar, byte-at-a-time restatement of the demultiplexing hot path of a-slide/Quade 0.3.2 on the
 * packed-row inputs the HIP library takes (include/quade_hip.h: qd_layout).  It exists so that
 * parity can be checked at sizes the pure-Python oracle (oracle/quade_oracle.py, the pinned one)
 * cannot reach in seconds; tests/test_oracle_c.py pins this file to the Python oracle.
 *
 * Restates (reference file:line):
 *   slice + fuse, Python slice clamping of short reads ... src/Quade.py:217-218, 246-247
 *   case fold for the lookup only ......................... src/Sample.py:65
 *   exact whole-string match, ordinal = section order ..... src/Sample.py:65-67, src/Quade.py:133
 *   min phred >= MIN_QUAL over the barcode slice .......... src/Sample.py:70
 *   counters ............................................... src/Sample.py:62,71-72,79-80,88
 * Deliberately different in structure from the GPU kernels: no hashing (linear scan over the
 * barcodes), no SWAR, no vector loads.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/liboracle_demux.so oracle/demux_oracle.c
 */
#include <stdint.h>
#include <string.h>

typedef struct {
    int32_t n_streams;
    int32_t seq_off[2], seq_width[2], seq_stride[2];
    int32_t qual_off[2], qual_width[2], qual_stride[2];
    int32_t key_width, mol_width;
} layout_t; /* same fields as qd_layout */

typedef struct {
    int32_t dual, min_qual;
    int32_t idx1_start, idx1_end, idx2_start, idx2_end;
    int32_t mol1_start, mol1_end, mol2_start, mol2_end;
} plan_t; /* same fields as qd_plan */

static int clamp_end(int end, int len) { return end < len ? end : len; }

/* counts: uint64[2S+4] as include/quade_hip.h; codes: uint16 per pair; mol: n x mol_width, zero padded */
int oracle_demux_rows(const layout_t* L, const plan_t* P, int32_t S, const uint8_t* barcodes,
                      const int32_t* bc_off, int64_t n, const uint8_t* const seq[2],
                      const uint8_t* const qual[2], const uint8_t* const len[2], uint16_t* codes,
                      uint8_t* mol, uint64_t* counts) {
    const int is[2] = {P->idx1_start, P->idx2_start}, ie[2] = {P->idx1_end, P->idx2_end};
    const int ms[2] = {P->mol1_start, P->mol2_start}, me[2] = {P->mol1_end, P->mol2_end};
    memset(counts, 0, sizeof(uint64_t) * (size_t)(2 * S + 4));
    for (int64_t r = 0; r < n; ++r) {
        uint8_t key[128], q[128], m[128];
        int klen = 0, mlen = 0;
        for (int k = 0; k < L->n_streams; ++k) {
            const int rlen = (len && len[k]) ? len[k][r] : 1 << 30;
            const uint8_t* srow = seq[k] + r * L->seq_stride[k];
            const uint8_t* qrow = qual[k] + r * L->qual_stride[k];
            for (int c = is[k]; c < clamp_end(ie[k], rlen); ++c) { /* index.seq / index.qual */
                key[klen] = srow[c - L->seq_off[k]];
                q[klen] = qrow[c - L->qual_off[k]];
                ++klen;
            }
        }
        for (int k = 0; k < L->n_streams; ++k) {
            const int rlen = (len && len[k]) ? len[k][r] : 1 << 30;
            const uint8_t* srow = seq[k] + r * L->seq_stride[k];
            for (int c = ms[k]; c < clamp_end(me[k], rlen); ++c) m[mlen++] = srow[c - L->seq_off[k]];
        }
        if (mol && L->mol_width) {
            memset(mol + r * L->mol_width, 0, (size_t)L->mol_width);
            memcpy(mol + r * L->mol_width, m, (size_t)mlen);
        }
        counts[0] += 1; /* TOTAL */
        for (int i = 0; i < klen; ++i)
            if (key[i] >= 'a' && key[i] <= 'z') key[i] -= 32; /* .upper() */
        int hit = -1;
        for (int s = 0; s < S && hit < 0; ++s) {
            const int blen = bc_off[s + 1] - bc_off[s];
            if (blen == klen && klen > 0 && memcmp(barcodes + bc_off[s], key, (size_t)klen) == 0) hit = s;
        }
        if (hit < 0) {
            counts[3] += 1;
            codes[r] = 0xFFFF;
            continue;
        }
        int minq = 255;
        for (int i = 0; i < klen; ++i)
            if (q[i] < minq) minq = q[i];
        if (minq - 33 >= P->min_qual) {
            counts[1] += 1;
            counts[4 + 2 * hit] += 1;
            codes[r] = (uint16_t)(2 * hit);
        } else {
            counts[2] += 1;
            counts[5 + 2 * hit] += 1;
            codes[r] = (uint16_t)(2 * hit + 1);
        }
    }
    return 0;
}
