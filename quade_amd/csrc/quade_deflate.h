// Device-side Huffman-only gzip members: descriptors and launch (quade_deflate.hip), used by quade_api.cpp's qd_deflater.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct qd_deflate_piece {
    uint64_t text_off;  // where the piece's text starts in the text buffer, 16-byte aligned
    uint32_t text_len;
    uint32_t crc32;     // of the piece's text, made on the host
};

// member i -> out[i * out_stride ...], its length -> out_bytes[i] (0: it did not fit out_stride bytes; out_stride % 4 == 0)
hipError_t qd_launch_huffman(const uint8_t* text, const qd_deflate_piece* pieces, uint32_t n_pieces, uint8_t* out, int64_t out_stride,
                             uint32_t* out_bytes, hipStream_t st);

// ---- LZ77 + dynamic Huffman members (`gzip_level : 1` on the device) ----
#ifndef QD_LZ_SUB
#define QD_LZ_SUB 65536 /* text per sub-block = per workgroup = per dynamic-Huffman block */
#endif
struct qd_lz_sub {
    uint64_t text_off;  // where the sub-block's text starts in the text buffer, 16-byte aligned
    uint32_t text_len;  // 1 .. QD_LZ_SUB
    uint32_t piece;
};
// piece i = sub-blocks first_sub[i] .. first_sub[i + 1] (first_sub has n_pieces + 1 entries).  Scratch: tokens = n_subs x QD_LZ_SUB
// words, sub_out = n_subs x sub_stride bytes (sub_stride % 4 == 0), sub_bytes = n_subs words.  Member i -> out[i * out_stride ...],
// its length -> out_bytes[i] (0: a sub-block or the member did not fit its slot).
hipError_t qd_launch_lz(const uint8_t* text, const qd_deflate_piece* pieces, uint32_t n_pieces, const qd_lz_sub* subs, const uint32_t* first_sub,
                        uint32_t n_subs, uint32_t* tokens, uint8_t* sub_out, int64_t sub_stride, uint32_t* sub_bytes, uint8_t* out,
                        int64_t out_stride, uint32_t* out_bytes, hipStream_t st);
// The two stages on their own, for a caller that has the members' CRC-32s made on the device in between: the sub-block kernel
// leaves every sub-block's CRC-32 in sub_crc (taken while the text is staged; NULL: not wanted), the caller combines them per
// piece into qd_deflate_piece::crc32 (quade_text.h, qd_text_crc32_combine) before the members are strung together.
hipError_t qd_launch_lz_subblocks(const uint8_t* text, const qd_lz_sub* subs, uint32_t n_subs, uint32_t* tokens, uint8_t* sub_out, int64_t sub_stride,
                                  uint32_t* sub_bytes, uint32_t* sub_crc, hipStream_t st);
hipError_t qd_launch_lz_members(const qd_deflate_piece* pieces, uint32_t n_pieces, const qd_lz_sub* subs, const uint32_t* first_sub, uint32_t n_subs,
                                const uint8_t* sub_out, int64_t sub_stride, const uint32_t* sub_bytes, uint8_t* out, int64_t out_stride, uint32_t* out_bytes,
                                hipStream_t st);
