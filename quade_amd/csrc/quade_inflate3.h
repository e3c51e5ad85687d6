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

// ---- ordinary gzip members on the device (quade_inflate3.hip, second half) -----------------------------------------------------------
// One step of one stream: the compressed bytes the caller holds on the device, decoding from a known block header on.
struct qd_gz_step {
    // in
    const uint8_t* comp;     // device; 16-byte aligned; readable up to 512 bytes behind comp_bytes
    uint64_t comp_bytes;
    uint64_t bit_start;      // a deflate block header starts here (bits from comp)
    int32_t at_end;          // no more input exists behind comp_bytes
    int32_t starved;         // out of decode(): not all of this step's stretches went into the launch (its token slots are indexed with
                             // 32 bits: the streams behind take their turn in the next step) -- "no progress" then says nothing about the stream
    uint8_t* carried;        // device, 32 KiB: the text in front of bit_start's block (in), in front of the next step's (out of qd_gz::resolve)
    uint32_t carried_valid;  // how much of it exists (0 at a member's start); updated by resolve
    uint32_t pad1;
    // out of decode()
    uint64_t text_len;       // the text this step makes
    uint64_t bit_next;       // where the next step starts: a block boundary behind bit_start (== bit_start: nothing was decoded)
    int32_t member_end;      // the member's last block is inside: bit_next lies behind it (the trailer follows at the next byte boundary)
    int32_t failed;          // != 0: the device gives this stream up (a QD_INFLATE_* code): nothing of this step counts
    // out of finish(): CRC-32 of this step's text (combined over its units)
    uint32_t crc32;
    uint32_t pad2;
};
struct qd_gz_stats {
    int64_t stretches, probed_bits_est, units, chain_retries, partial_last;
    int64_t plain_probes;  // decodes done again without the probe's text filter (a stream that is not text)
};
class qd_gz_impl;
// Several streams advance together (one token launch for all of them: its throughput is the number of stretches in flight).
class qd_gz {
  public:
    qd_gz();
    ~qd_gz();
    qd_gz(const qd_gz&) = delete;
    qd_gz& operator=(const qd_gz&) = delete;
    // probe + tokens + chain check; synchronises `st` twice.  Fills text_len / bit_next / member_end / failed of every step.
    hipError_t decode(qd_gz_step* steps, int n, hipStream_t st, int slots_per_byte = 4, bool text_filter = true);
    // the decoded steps' text -> out[i][0 .. text_len) (device); asynchronous on `st`; carried windows updated on the device
    hipError_t resolve(qd_gz_step* steps, int n, uint8_t* const* out, hipStream_t st);
    // after `st` has run resolve(): the steps' CRC-32s, and failed != 0 for a stream whose tokens did not resolve (a damaged stream)
    hipError_t finish(qd_gz_step* steps, int n);
    qd_gz_stats stats() const;
    // device buffers for steps of about this size in one go (comp_bytes, text_bytes: summed over the streams of a step): growing them
    // step by step drains the device every time
    hipError_t reserve(uint64_t comp_bytes, uint64_t text_bytes);
    // knobs (tests: small values make many stretches and units out of small inputs)
    uint64_t stretch_bytes = 16u << 10;  // compressed bytes per stretch: a lane decodes one
    bool unit_text_given = false;        // unit_text is an order (tests, A/B runs), not an upper bound
    uint64_t unit_text = 2u << 20;       // text per resolve unit, about (the windows kernel walks a stream's units one after the other)

  private:
    qd_gz_impl* p_;
};
