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
