// Device-side BGZF inflate: descriptors and launch (quade_inflate.hip), used by quade_api.cpp's qd_inflater.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// status codes of one block (0 = inflated to exactly out_len bytes)
#define QD_INFLATE_TRUNCATED 1     /* the compressed payload ended inside a code                      */
#define QD_INFLATE_BAD_TYPE 2      /* reserved block type, or no final block                          */
#define QD_INFLATE_BAD_STORED 3    /* stored block: LEN / NLEN mismatch                               */
#define QD_INFLATE_BAD_TABLE 4     /* dynamic block: invalid code lengths                             */
#define QD_INFLATE_BAD_CODE 5      /* a code that is not in its table                                 */
#define QD_INFLATE_BAD_DISTANCE 6  /* a match reaching behind the start of the block                  */
#define QD_INFLATE_OVERRUN 7       /* more output than the block's ISIZE                              */
#define QD_INFLATE_LENGTH 8        /* less output than the block's ISIZE                              */
#define QD_INFLATE_CRC 9           /* the text's CRC-32 is not the one in the block's trailer (checked on the device when asked) */
#define QD_INFLATE_TABLE_SPACE 10  /* third form: more symbols with long codes than a lane's table holds -- another form takes the block */
#define QD_INFLATE_TOKEN_SPACE 11  /* third form: more tokens than the unit's slot region holds                                          */
#define QD_INFLATE_CHAIN 12        /* third form, a stretch of a gzip member: the decode passed its stop position, no block starts there   */

struct qd_inflate_block {
    uint32_t in_off, in_len;    // raw deflate payload of the block inside the compressed buffer
    uint32_t out_off, out_len;  // where its text goes, and how much there must be (ISIZE)
};

hipError_t qd_launch_inflate(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out,
                             int32_t* status, hipStream_t st);

// Second form: 256 lanes per block (speculative spans that synchronise; matches resolved from a list).  matches: scratch of
// n_blocks x matches_per_block 64-bit entries; max_in_len: the longest payload of the launch (sizes the workgroups' LDS).
#define QD_INFLATE_MATCHES_PER_BLOCK 12288
size_t qd_inflate2_lds(uint32_t max_in_len);  // LDS a workgroup of the second form needs; above 160 KB (payloads beyond ~52 KB: stored blocks) use the first
hipError_t qd_launch_inflate2(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out, int32_t* status,
                              unsigned long long* matches, uint32_t matches_per_block, uint32_t max_in_len, hipStream_t st,
                              uint32_t* rounds_out = nullptr,           // rounds_out (measurement): per block, rounds | deflate blocks << 16
                              const uint32_t* expect_crc = nullptr);    // per block, the CRC-32 of its trailer: checked while the text is in LDS (QD_INFLATE_CRC)
