/* A plain C99 client of libquade_hip.so: what a non-Python binding does with the C ABI (include/quade_hip.h).
 * No HIP header, no device pointer: index-read fastq text is packed straight into a pinned slot, the slot is
 * submitted, the routing codes come back in the slot, the counters from qd_get_counts.
 *   gcc -std=c99 -I include examples/abi_client.c -L quade_amd/lib -lquade_hip -o abi_client
 * exit codes: 0 = ran on the GPU and every result is as expected, 77 = no MI355X here (the library never falls
 * back to the CPU: qd_create says QD_ERR_NO_DEVICE), anything else = a failure. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "quade_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != QD_OK) {                                                              \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, qd_strerror(rc_), qd_last_error(ctx)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(void) {
    qd_ctx* ctx = NULL;
    if (qd_version() != QD_ABI_VERSION) return 2;
    int rc = qd_create(0, &ctx);
    if (rc == QD_ERR_NO_DEVICE) {
        printf("no usable device: %s\n", qd_last_error(NULL));
        return 77;
    }
    if (rc != QD_OK) return 3;

    /* [index] index2 : True, index1 1..8, index2 1..8, molecular1 9..12; minimal_qual 25 (src/Quade.py:96-116) */
    qd_plan plan;
    memset(&plan, 0, sizeof plan);
    plan.dual = 1;
    plan.min_qual = 25;
    plan.idx1_start = 0, plan.idx1_end = 8;
    plan.idx2_start = 0, plan.idx2_end = 8;
    plan.mol1_start = 8, plan.mol1_end = 12;
    CHECK(qd_set_plan(ctx, &plan));
    qd_layout lay;
    CHECK(qd_get_layout(ctx, &lay));

    /* two samples, in Sample.SAMPLE_LIST order: index1_seq + index2_seq */
    const char* barcodes = "ACGTACGTTTTTCCCC" "GGGGGGGGAAAAAAAA";
    const int32_t offsets[3] = {0, 16, 32};
    CHECK(qd_set_barcodes(ctx, 2, (const uint8_t*)barcodes, offsets));

    /* index reads as fastq text: pass, fail (one base below phred 25), undetermined, lower case (folds), and a
     * record whose quality line is one short (skipped inside its own stream, as pyFastq does) */
    const char* i1 =
        "@r1\nACGTACGTAAAA\n+\nIIIIIIIIIIII\n"
        "@r2\nGGGGGGGGCCCC\n+\nIIII5IIIIIII\n"
        "@r3\nNNNNNNNNGGGG\n+\nIIIIIIIIIIII\n"
        "@bad\nACGTACGTAAAA\n+\nIIIIIIIIIII\n"
        "@r4\nacgtacgtTTTT\n+\nIIIIIIIIIIII\n";
    const char* i2 =
        "@r1\nTTTTCCCC\n+\nIIIIIIII\n"
        "@r2\nAAAAAAAA\n+\nIIIIIIII\n"
        "@r3\nTTTTCCCC\n+\nIIIIIIII\n"
        "@r4\nTTTTCCCC\n+\nIIIIIIII\n";
    CHECK(qd_slots_create(ctx, 1, 64));
    qd_slot_buffers sb;
    CHECK(qd_slot_get(ctx, 0, &sb));
    const char* text[2] = {i1, i2};
    int64_t n_short[2] = {0, 0}, n = -1;
    for (int k = 0; k < 2; ++k) {
        int32_t full = 0;
        int64_t consumed = 0;
        const int64_t got = qd_pack_index_fastq(&lay, k, (const uint8_t*)text[k], (int64_t)strlen(text[k]), sb.max_pairs,
                                                sb.seq[k], sb.qual[k], sb.len[k], &full, &consumed, sb.short_idx[k],
                                                sb.short_cap, &n_short[k]);
        if (got != 4 || !full || consumed != (int64_t)strlen(text[k])) {
            fprintf(stderr, "stream %d: %lld records, full %d, consumed %lld\n", k, (long long)got, full, (long long)consumed);
            return 4;
        }
        n = (n < 0 || got < n) ? got : n; /* the chunk ends with the first exhausted stream (src/Quade.py:223-224) */
    }
    CHECK(qd_submit(ctx, 0, n, 0));
    CHECK(qd_wait(ctx, 0));
    const uint16_t want[4] = {0, 3, QD_CODE_UNDETERMINED, 0}; /* 2*i = sample i pass, 2*i+1 = fail */
    for (int r = 0; r < 4; ++r)
        if (sb.codes[r] != want[r]) {
            fprintf(stderr, "pair %d: code %u, expected %u\n", r, sb.codes[r], want[r]);
            return 5;
        }
    if (memcmp(sb.mol, "AAAA" "CCCC" "GGGG" "TTTT", 16) != 0) return 6; /* molecular bytes, raw case */
    uint64_t counts[2 * 2 + 4];
    CHECK(qd_get_counts(ctx, counts, 8));
    const uint64_t wantc[8] = {4, 2, 1, 1, 2, 0, 0, 1}; /* total, pass, fail, undetermined, then pass/fail per sample */
    if (memcmp(counts, wantc, sizeof wantc) != 0) return 7;

    /* name tags of the routed records: ":IDX:MOL" (src/FastqWriter.py:61-66) */
    uint8_t tags[4][64], tag_len[4];
    const uint8_t* seq_rows[2] = {sb.seq[0], sb.seq[1]};
    const uint8_t* len_rows[2] = {NULL, NULL};
    CHECK(qd_build_tags(&lay, &plan, 4, seq_rows, len_rows, sb.mol, &tags[0][0], 64, tag_len));
    if (tag_len[3] != 22 || memcmp(tags[3], ":acgtacgtTTTTCCCC:TTTT", 22) != 0) {
        fprintf(stderr, "tag of pair 3: %.*s\n", tag_len[3], (const char*)tags[3]);
        return 8;
    }
    char name[128];
    int32_t cus = 0;
    int64_t mem = 0;
    CHECK(qd_device_info(ctx, name, (int32_t)sizeof name, &cus, &mem));
    printf("ok: 4 pairs on %s (%d CUs): codes %u %u %u %u, kernel kind %d\n", name, cus, sb.codes[0], sb.codes[1], sb.codes[2],
           sb.codes[3], qd_kernel_kind(ctx, 0));
    CHECK(qd_slots_destroy(ctx));
    CHECK(qd_destroy(ctx));
    return 0;
}
