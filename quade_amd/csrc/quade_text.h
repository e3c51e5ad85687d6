// Device-side fastq text stages (quade_text.hip), used by the chunk pipeline (quade_pipe.cpp): record scan of inflated text,
// index-row packing, stable scatter by routing code, record formatting, CRC-32 and member packing.  Everything here takes
// DEVICE pointers and a stream and returns after the launch; nothing synchronises.
//
// What each stage replaces in the reference (file:line under /root/reference):
//   record scan ........ pyFastq.FastqReader as used at src/Quade.py:203-214 (4-line records; a record whose sequence and
//                        quality lengths differ is dropped inside its own stream; name = header without its
//                        first byte, first whitespace-delimited token)
//   row packing ........ the operands of index1[s:e] + index2[s:e] (src/Quade.py:217-218, 246-247), Python slice clamping
//   scatter + format ... Sample.FINDER's routing tail (src/Sample.py:74-91) and FastqWriter.__call__ (src/FastqWriter.py:61-69)
//   CRC-32 ............. the gzip trailer FastqWriter.flush_buffers' gzip.open(..., "ab") writes (src/FastqWriter.py:83-90)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/quade_hip.h"

// one kept record of a text window: offsets into the window's text
struct qd_rec {
    uint32_t head;      // first byte of the header line
    uint32_t name_off;  // first byte of the name (behind the header's first byte and any blanks)
    uint32_t name_len;
    uint32_t seq;       // first byte of the sequence line
    uint32_t seq_len;   // without a trailing '\r'; the quality line of a kept record has the same length
    uint32_t qual;      // first byte of the quality line
};

// what a scan leaves behind for the host (one per stream window; device memory, copied to the host by the caller)
struct qd_scan_result {
    uint32_t n_lines;     // newline-terminated lines of the window (a last line without newline counts at the end of a file)
    uint32_t n_records;   // complete records = n_lines / 4
    uint32_t n_kept;      // of them kept (sequence and quality of one length)
    uint32_t n_short;     // kept records whose sequence is shorter than `need` (index streams: Python slice clamping applies)
    uint32_t tail_start;  // first byte behind the last complete record
    uint32_t overflow;    // != 0: the line table was too small (n_lines is still right): grow it and scan again
    uint32_t first_bad;   // BGZF verification: index of the first block that did not inflate / check, 0xFFFFFFFF = none
    uint32_t carry_start; // filled by qd_text_carry_info: where the text the next batch keeps starts
};

#define QD_TEXT_TILE 16384u  // bytes of text per workgroup of the line kernels

// Scratch of one window's scan: tile_counts / tile_base hold ceil((len + 1) / QD_TEXT_TILE) + 1 words each, lines line_cap words,
// rec_tile (ceil(line_cap / 4 / 1024) + 2) words, recs line_cap / 4 entries.
struct qd_scan_scratch {
    uint32_t* tile_counts;
    uint32_t* tile_base;
    uint32_t* lines;     // position of every newline, in text order
    uint32_t line_cap;
    uint32_t* rec_tile;  // kept records per tile of 1024 records, then their exclusive scan
    qd_rec* recs;        // kept records, compacted, in text order
};

// text[0 .. len): whole lines from a record start on.  at_eof: the stream ends with this window (a last line without newline
// counts).  want_names: fill name_off / name_len (insert reads).  need: sequences shorter than this count as short.
hipError_t qd_text_scan(const uint8_t* text, uint32_t len, int at_eof, int want_names, uint32_t need, const qd_scan_scratch& s,
                        qd_scan_result* result, hipStream_t st);

// ... of up to four windows at once (a batch's streams): same results; the stages of all windows back to back, the small serial kernels
// once for all of them
struct qd_scan_job {
    const uint8_t* text;
    uint32_t len;
    int at_eof, want_names;
    uint32_t need;
    qd_scan_scratch s;
    qd_scan_result* result;
};
hipError_t qd_text_scan_many(int n_windows, const qd_scan_job* jobs, hipStream_t st);

// carry_start of every stream for a batch that consumes the first n[s] kept records of stream s: the head of kept record n[s], or
// tail_start when the window holds no more kept records
hipError_t qd_text_carry_info(const qd_rec* const recs[4], qd_scan_result* const results[4], int n_streams, const uint32_t n[4], hipStream_t st);

// Grains (a chunk shared by several ranks): the window's text is cut at grain_start[0 .. n_grains] (text offsets, ascending; the
// last entry closes the last grain).  A record belongs to the grain its header line starts in.  For each grain and each residue
// phase = (lines of the file before the grain) mod 4: how many of its records are kept, where the first kept one starts, and whether
// a record of the grain reaches beyond the window (then the caller needs more text behind it).  Needs the scan's line table.
struct qd_grain_index {
    uint32_t first_line;  // newlines of the window before the grain
    uint32_t n_lines;     // newlines inside the grain
    uint32_t kept[4], first_head[4], incomplete[4];
};
hipError_t qd_text_grain_index(const uint8_t* text, const uint32_t* lines, uint32_t line_cap, const qd_scan_result* res, const uint32_t* grain_start,
                               uint32_t n_grains, int at_eof, qd_grain_index* out, hipStream_t st);

// Index rows of pairs [0, n): stream k's rows from text[k] / recs[k] as qd_layout says (seq window zero padded, barcode
// qualities 0xFF padded, len = min(255, read length)).  short_idx / n_short: the pairs with a read shorter than its window
// (unique, any order; *n_short must be zero before the launch).
struct qd_pack_args {
    const uint8_t* text[2];
    const qd_rec* recs[2];
    uint8_t* seq[2];
    uint8_t* qual[2];
    uint8_t* len[2];
    uint32_t* short_idx;
    uint32_t* n_short;
    uint32_t short_cap;
};
hipError_t qd_text_pack_rows(const qd_layout& L, uint32_t n, const qd_pack_args& a, hipStream_t st);

// Destination and output lengths of every pair: dest = routing code (0xFFFF -> 2 * S), len1 / len2 = bytes of the pair's two
// output records "@name:IDX[:MOL]\nseq\n+\nqual\n" (0 when the destination's write flag is off).
struct qd_route_args {
    const uint16_t* codes;
    const qd_rec* r1;
    const qd_rec* r2;
    const qd_rec* idx[2];
    uint16_t* dest;
    uint32_t* len1;
    uint32_t* len2;
};
hipError_t qd_text_dest_lens(const qd_plan& P, uint32_t n_samples, int write_pass, int write_fail, int write_undet, uint32_t n,
                             const qd_route_args& a, hipStream_t st);

// Stable sort of 0 .. n-1 by dest (LSD radix, 8 bits per pass; one pass when n_dest <= 256).  Scratch: hist H + H / 4096 + 4 words
// with H = 256 * ceil(n / 1024), tmp n words.  perm[k] = the pair at sorted position k.
hipError_t qd_text_sort_by_dest(const uint16_t* dest, uint32_t n, uint32_t n_dest, uint32_t* hist, uint32_t* tmp, uint32_t* perm,
                                hipStream_t st);

// out[k] = sum of in[perm[i]] for i < k (exclusive), out[n] = the total; sdest[k] = dest[perm[k]] when sdest != NULL.
// Scratch: tiles ceil(n / 4096) + 2 words.  The total must stay below 2^32 (the caller bounds its windows).
hipError_t qd_text_scan_gathered(const uint32_t* in, const uint32_t* perm, uint32_t n, uint32_t* tiles, uint32_t* out,
                                 const uint16_t* dest, uint16_t* sdest, hipStream_t st);

// first[d] = the first sorted position of destination d (0xFFFFFFFF: none), g1_first[d] / g2_first[d] = G1 / G2 there
hipError_t qd_text_dest_bounds(const uint16_t* sdest, const uint32_t* g1, const uint32_t* g2, uint32_t n, uint32_t n_dest,
                               uint32_t* first, uint32_t* g1_first, uint32_t* g2_first, hipStream_t st);

// The records themselves: pair perm[k]'s two records to out1 + base1[sdest[k]] + g1[k] and out2 + base2[sdest[k]] + g2[k]
// (a destination's base is what the host made of its first position: where its text starts minus g at that position).
struct qd_format_args {
    const uint32_t* perm;
    const uint16_t* sdest;
    const uint32_t* g1;
    const uint32_t* g2;
    const int64_t* base1;
    const int64_t* base2;
    const uint8_t* text1;
    const uint8_t* text2;
    const qd_rec* r1;
    const qd_rec* r2;
    const uint8_t* itext[2];
    const qd_rec* idx[2];
    uint8_t* out1;
    uint8_t* out2;
};
hipError_t qd_text_format(const qd_plan& P, uint32_t n_samples, int write_pass, int write_fail, int write_undet, uint32_t n,
                          const qd_format_args& a, hipStream_t st);

// CRC-32 (the gzip / zlib polynomial) of n ranges of text, each at most 64 KiB: crc[i] of text[off[i] .. off[i] + len[i])
struct qd_crc_range {
    uint64_t off;
    uint32_t len;
    uint32_t pad;
};
hipError_t qd_text_crc32(const uint8_t* text, const qd_crc_range* ranges, uint32_t n, uint32_t* crc, hipStream_t st);
// the same for the text of inflated BGZF blocks: range i = out[blocks[i].out_off .. + blocks[i].out_len)
struct qd_inflate_block;
hipError_t qd_text_crc32_blocks(const uint8_t* out, const qd_inflate_block* blocks, uint32_t n, uint32_t* crc, hipStream_t st);
// crc of piece i = the CRCs of its ranges first[i] .. first[i + 1] combined, written to piece_crc[i * stride_words] (the
// pipeline points this at qd_deflate_piece::crc32)
hipError_t qd_text_crc32_combine(const qd_crc_range* ranges, const uint32_t* crc, const uint32_t* first, uint32_t n_pieces,
                                 uint32_t* piece_crc, uint32_t stride_words, hipStream_t st);
// BGZF verification: status[i] != 0 or crc[i] != expect[i] -> atomicMin(first_bad, i)
hipError_t qd_text_check_blocks(const int32_t* status, const uint32_t* crc, const uint32_t* expect, uint32_t n, uint32_t base_index,
                                uint32_t* first_bad, hipStream_t st);

// Members made in slots of `stride` bytes -> one packed byte stream: offsets[i] = sum of len[j], j < i (offsets[n] = total),
// packed[offsets[i] ..] = slots[i * stride .. + len[i]).
hipError_t qd_text_pack_members(const uint8_t* slots, int64_t stride, const uint32_t* len, uint32_t n, uint64_t* offsets, uint8_t* packed,
                                hipStream_t st);
