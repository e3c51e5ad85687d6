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

struct qd_inflate_block {
    uint32_t in_off, in_len;    // raw deflate payload of the block inside the compressed buffer
    uint32_t out_off, out_len;  // where its text goes, and how much there must be (ISIZE)
};

hipError_t qd_launch_inflate(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out,
                             int32_t* status, hipStream_t st);
