// Third inflater: launches (quade_inflate3.hip).  Same status codes as quade_inflate.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quade_inflate.h"

// token slots a BGZF block's unit may fill: every byte of 64 KiB a literal, plus the last store's padding
#define QD_INFLATE3_TOK_STRIDE (65536u + 8u)

// One BGZF block of a launch, by address: blocks of several input streams (their compressed bytes and their text in different
// allocations) go down in ONE launch -- the token kernel's throughput is the number of blocks in flight.
struct qd_inflate3_job {
    const uint8_t* payload;  // raw deflate; readable up to 512 bytes behind payload + in_len (a lane's input ring runs ahead of its position)
    uint8_t* out;            // out_len bytes of text
    uint32_t in_len, out_len;
    uint32_t expect_crc, check_crc;  // check_crc != 0: the text's CRC-32 must be expect_crc (QD_INFLATE_CRC)
};

// device scratch of a launch of n_blocks BGZF blocks: token slots | results | code-length scratch | jobs
size_t qd_inflate3_scratch_bytes(uint32_t n_blocks);
// jobs (device memory) -> text; status[i] = 0 or QD_INFLATE_* (QD_INFLATE_TABLE_SPACE: the block's Huffman codes have more long
// symbols than a lane's table holds -- another form takes it).  scratch: 256-byte aligned, qd_inflate3_scratch_bytes(n_blocks).
hipError_t qd_launch_inflate3_jobs(const qd_inflate3_job* d_jobs, uint32_t n_blocks, int32_t* status, void* scratch, hipStream_t st);
// the same for blocks of one compressed buffer into one text buffer (qd_inflate_block offsets); comp readable up to 512 bytes behind
// comp_bytes; expect_crc (device, per block) may be null
hipError_t qd_launch_inflate3(const uint8_t* comp, size_t comp_bytes, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out, int32_t* status, void* scratch,
                              hipStream_t st, const uint32_t* expect_crc = nullptr);
