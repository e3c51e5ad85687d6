// BGZF block inflate on the device (gfx950): row N2's "ship the raw fastq.gz to the GPU" follow-up.
//
// What it replaces: the gunzip inside pyFastq.FastqReader that the reference's chunk loops draw their records
// from (src/Quade.py:203-214) -- for BGZF files (bgzip layout: independent gzip members of <= 64 KiB, each
// with its compressed size in the header and its inflated size in the trailer), whose blocks the native
// reader otherwise inflates in parallel on host threads (quade_io.cpp).  End to end the host cores are what
// bounds the rate and inflate is 30 % of their work, while the GPU idles.
//
// One WAVE inflates one block.  DEFLATE is a serial bit stream: the symbol decode is one dependent chain, run
// redundantly (wave-uniform, kept in scalar registers) by all 64 lanes out of tables in LDS -- a 10-bit lookup for
// literal/length codes, an 8-bit one for distances, the canonical bit-by-bit decode (RFC 1951; count[] / symbol[]
// per code) behind them for longer codes; the payload is read one dword ahead.  What the lanes share is the work
// that is not serial: a match of up to 258 bytes is copied by all lanes at once.  The text is written in place
// (global memory; a match reads back what this same wave stored earlier, through the same L1, in program order),
// so a workgroup needs 3.6 KB of LDS and thousands of blocks are in flight: a launch takes ~16 ms whether it
// holds 500 blocks or 8 000 (13 GB/s of text then).  History of the design, measured on 512 MB of fastq text
// (profiles/r02_device_inflate.txt): one LANE per block with tables in scratch memory 0.1-1 GB/s (every table
// step and copied byte a dependent trip to memory); one wave per block with the 64 KiB window in LDS 2.4 GB/s
// (two waves per CU: nothing hides the chain's latency); this form 13 GB/s.
//
// Written for a machine where a fault can take the whole node down: every input read, window access, table
// index and loop is bounded; any inconsistency ends the block with a status code and the host inflates that run
// itself.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "crc_lds.h"
#include "quade_inflate.h"

namespace {

#ifndef QD_INFLATE_LBITS
#define QD_INFLATE_LBITS 10
#endif
#ifndef QD_INFLATE_DBITS
#define QD_INFLATE_DBITS 8
#endif
constexpr int LBITS = QD_INFLATE_LBITS, DBITS = QD_INFLATE_DBITS;  // first-level lookup widths
constexpr uint32_t WINDOW = 65536;           // BGZF: ISIZE <= 64 KiB
struct Lds {                                  // dynamic LDS of one workgroup (= one wave): the tables only
    uint16_t llut[1 << LBITS];                // literal/length: symbol | code length << 9 (0 = longer code or none)
    uint16_t dlut[1 << DBITS];                // distance: symbol | code length << 5
    uint16_t lsym[288], dsym[32];             // symbols ordered by code (canonical decode of the longer codes)
    uint16_t lcount[16], dcount[16];
    uint8_t lengths[320];
};

// Every value of the decode chain is wave-uniform; readfirstlane says so to the compiler, which then keeps the bit
// buffer and the counters in scalar registers (the chain is one dependent instruction after the other: scalar ones
// issue without the vector pipeline's latency).
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct Bits {
    const uint8_t* p;
    uint32_t pos, end;  // next byte to fetch into `ahead`, one past the payload
    uint64_t buf;       // LSB first
    int cnt;            // valid bits in buf; < 0 after reading past the end (the caller checks)
    uint32_t ahead, ahead_bytes;  // the next <= 4 payload bytes, loaded one refill early (their latency is hidden)
};

__device__ __forceinline__ void fetch_ahead(Bits& b) {
    const uint32_t left = b.end - b.pos, take = left < 4 ? left : 4;
    uint32_t w = 0;
    if (take > 0) w |= b.p[b.pos];
    if (take > 1) w |= (uint32_t)b.p[b.pos + 1] << 8;
    if (take > 2) w |= (uint32_t)b.p[b.pos + 2] << 16;
    if (take > 3) w |= (uint32_t)b.p[b.pos + 3] << 24;
    b.ahead = w;  // (made uniform where it is consumed: the load stays in flight until then)
    b.ahead_bytes = take;
    b.pos += take;
}

// at least 32 valid bits unless the payload is exhausted (then zeros are shifted in and cnt runs negative on use)
__device__ __forceinline__ void refill(Bits& b) {
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)uni(b.ahead) << b.cnt;
        b.cnt += 8 * (int)b.ahead_bytes;
        fetch_ahead(b);
    }
}
__device__ __forceinline__ uint32_t take(Bits& b, int n) {  // n <= 16; the caller has refilled
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// canonical decode, one bit per step; -1 = no such code
__device__ int slow_decode(Bits& b, const uint16_t* count, const uint16_t* symbol, int nsym) {
    int code = 0, first = 0, index = 0;
#pragma unroll 1
    for (int len = 1; len <= 15; ++len) {
        code |= (int)take(b, 1);
        const int c = (int)uni(count[len]);
        if (code - c < first) {
            const int at = index + (code - first);
            return at < nsym ? (int)uni(symbol[at]) : -1;
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Tables of one Huffman code from n code lengths, by all threads of the workgroup: the lengths are counted with LDS atomics, one
// thread derives the first code and the first slot of every length, then every thread places its symbols (a symbol's place among
// those of its length = how many earlier symbols have that length) and fills their look-up entries.  (Lane 0 alone -- a chain
// of dependent LDS read-modify-writes per symbol -- took ~90 us per block of the second form's 580.)  Returns 0 for a complete
// code, > 0 incomplete, < 0 over-subscribed.  lut entries: symbol | length << shift for codes of at most `bits` bits.
__device__ int build(const uint8_t* length, int n, uint16_t* count, uint16_t* symbol, uint16_t* lut, int bits, int shift,
                     uint32_t lane) {
    const uint32_t nthr = blockDim.x;
    __shared__ uint32_t cnt32[16];
    __shared__ uint16_t offs[16], next[16];
    __shared__ int verdict, usable;
    for (uint32_t i = lane; i < (1u << bits); i += nthr) lut[i] = 0;
    if (lane < 16) cnt32[lane] = 0;
    __syncthreads();
    for (uint32_t s = lane; s < (uint32_t)n; s += nthr) atomicAdd(&cnt32[length[s] & 15], 1u);
    __syncthreads();
    if (lane == 0) {
        int left = 1;
        for (int l = 0; l <= 15; ++l) count[l] = (uint16_t)cnt32[l];
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) break;
        }
        usable = left >= 0 && count[0] != n;
        if (usable) {
            offs[1] = 0;
            int code = 0;
            next[0] = 0;
            for (int l = 1; l <= 15; ++l) {
                code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
                next[l] = (uint16_t)code;
                if (l < 15) offs[l + 1] = offs[l] + count[l];
            }
        }
        count[0] = (uint16_t)(count[0] == n ? 0xFFFF : count[0]);  // no codes at all: marked
        verdict = left;
    }
    __syncthreads();
    if (usable) {
        for (uint32_t s = lane; s < (uint32_t)n; s += nthr) {
            const uint32_t l = length[s] & 15;
            if (!l) continue;
            // how many earlier symbols have this length: four lengths per LDS word (every length is < 16; byte by byte this loop was
            // most of a table's time)
            uint32_t rank = 0;
            {
                const uint32_t* lw = reinterpret_cast<const uint32_t*>(length);  // (the tables' lengths start on word boundaries)
                const uint32_t pat = l * 0x01010101u, whole = s >> 2;
                auto equal_bytes = [&](uint32_t w) {  // 0x80 in every byte of w that equals l
                    const uint32_t x = w ^ pat;
                    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
                };
                for (uint32_t q = 0; q < whole; ++q) rank += (uint32_t)__popc(equal_bytes(lw[q]));
                if (s & 3u) rank += (uint32_t)__popc(equal_bytes(lw[whole]) & ((1u << (8u * (s & 3u))) - 1u));
            }
            symbol[offs[l] + rank] = (uint16_t)s;
            const uint32_t c = next[l] + rank;
            if ((int)l <= bits) {  // the stream carries codes MSB first inside an LSB-first bit order: index by the reversed code
                const uint32_t rev = __brev(c) >> (32 - l);
                const uint16_t e = (uint16_t)(s | (l << shift));
                for (uint32_t k = rev; k < (1u << bits); k += 1u << l) lut[k] = e;
            }
        }
    }
    __syncthreads();
    return verdict;
}

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// literal/length + distance codes -> the window; 0 at the end-of-block symbol.  Wave-uniform control flow.
__device__ int codes(Bits& b, Lds& L, uint8_t* o, uint32_t& opos, uint32_t olen, uint32_t lane, int nlsym, int ndsym) {
#pragma unroll 1
    for (;;) {
        refill(b);
        int sym;
        const uint32_t e = uni(L.llut[(uint32_t)b.buf & ((1u << LBITS) - 1u)]);
        if (e >> 9) {
            sym = e & 511;
            b.buf >>= (e >> 9);
            b.cnt -= (e >> 9);
        } else {
            sym = slow_decode(b, L.lcount, L.lsym, nlsym);
        }
        if (b.cnt < 0) return QD_INFLATE_TRUNCATED;
        if (sym < 0) return QD_INFLATE_BAD_CODE;
        if (sym < 256) {
            if (opos >= olen) return QD_INFLATE_OVERRUN;
            if (lane == 0) o[opos] = (uint8_t)sym;
            ++opos;
            continue;
        }
        if (sym == 256) return 0;
        sym -= 257;
        if (sym >= 29) return QD_INFLATE_BAD_CODE;
        const uint32_t len = (uint32_t)LBASE[sym] + take(b, LEXT[sym]);
        refill(b);
        int ds;
        const uint32_t d = uni(L.dlut[(uint32_t)b.buf & ((1u << DBITS) - 1u)]);
        if (d >> 5) {
            ds = d & 31;
            b.buf >>= (d >> 5);
            b.cnt -= (d >> 5);
        } else {
            ds = slow_decode(b, L.dcount, L.dsym, ndsym);
        }
        if (ds < 0 || ds >= 30) return b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_CODE;
        const uint32_t dist = (uint32_t)DBASE[ds] + take(b, DEXT[ds]);
        if (b.cnt < 0) return QD_INFLATE_TRUNCATED;
        if (dist > opos) return QD_INFLATE_BAD_DISTANCE;  // a BGZF block never reaches behind its own start
        if (len > olen - opos) return QD_INFLATE_OVERRUN;
        // the copy: all lanes, one byte each per round; a match longer than its distance repeats the last `dist` bytes.
        // Source bytes were stored earlier by this same wave (lane 0's literals, earlier copies): its loads come
        // after those stores in the one instruction stream all lanes share, through the same L1.
        const uint32_t from = opos - dist;
        for (uint32_t i = lane; i < len; i += 64) {
            const uint32_t s = dist >= len ? i : i % dist;
            o[opos + i] = o[from + s];
        }
        opos += len;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Second form (late r03): the block's symbols decoded by 256 lanes at once.
// The first form's chain -- one symbol after the other, run redundantly by a whole wave -- makes a launch take ~16 ms
// whatever it holds and fills the GPU at 13 GB/s; end to end its lanes starve the coder's launches.  Here one
// workgroup of 256 threads inflates a block:
//   * header and tables of a deflate block as before (wave-uniform, cheap);
//   * the block's bit range is cut into 256 spans.  Lane i decodes tokens from a GUESSED bit position near the start of its span
//     until one ends behind the span, and reports where (exit position), how many bytes and matches it saw.  Lane 0's
//     start is true; in the next round every lane starts where its predecessor ended.  Huffman codes resynchronise after
//     a few symbols, so after 2-4 rounds every lane starts at its predecessor's exit: the chain 0, 1, ... up to the lane
//     that meets the end-of-block symbol is then exactly the sequential decode (each round confirms at least one more
//     lane, so 256 rounds bound it);
//   * a scan of the byte and match counts places every span's output; the lanes decode their spans once more, writing
//     literals into the block's text in LDS and matches (destination, length, distance) into a list in text order;
//   * one wave walks the list, every match copied by its 64 lanes inside LDS; the text leaves with coalesced stores.
// The payload is staged in LDS too (a lane reads bits at arbitrary positions).  Same status codes and bounds as the first form.
#ifndef QD_INFLATE2_MAX_ROUNDS
#define QD_INFLATE2_MAX_ROUNDS (1024 + 1) /* every round confirms at least one more lane (A/B: a small bound shows how many rounds real blocks need) */
#endif
#ifndef QD_INFLATE2_THREADS
#define QD_INFLATE2_THREADS 0 /* lanes (= spans) per block: 0 = 1 024 where the payload leaves the LDS for it, else 512 (512 against 256: 0.58 vs 0.83 ms per block;
                                  1 024 against 512: the end-to-end job 9 % faster, profiles/r04_ab_inflate_threads.txt); 256 / 512 / 1024 fix it */
#endif
namespace v2 {
#ifndef QD_INFLATE2_Q
#define QD_INFLATE2_Q 16384 /* positions per window of the match stage (2 bytes of LDS each) */
#endif
static_assert(2 * QD_INFLATE2_Q >= 4096 + 64, "the CRC stage's tables lie where the match stage keeps its parents");
#if defined(QD_INFLATE_TIMING)
__device__ unsigned long long g_inflate_ticks[11];
#endif
template <int NT>
struct Lds2 {
    Lds t;
    uint32_t start[NT], exitp[NT], nout[NT], nmat[NT], flag[NT];
    uint32_t start_sum[2 * (NT / 64)];  // the waves' totals of the span scan
    uint32_t ctl[8];
    uint32_t win_first[65536 / QD_INFLATE2_Q + 2];  // per window of the match stage: the first match that starts in it
    uint32_t ll[1 << LBITS], dl[1 << DBITS];        // the span decode's tables (lit_entry / dist_entry)
};
enum { F_EOB = 1, F_ERR_CODE = 2, F_ERR_TRUNC = 4 };

__device__ __forceinline__ uint32_t peek(const uint32_t* pw, uint32_t pos) {  // 32 bits of the payload from bit `pos` on
    const uint32_t i = pos >> 5;
    return __builtin_amdgcn_alignbit(pw[i + 1], pw[i], pos & 31u);
}
// canonical decode of a code longer than the first-level table; bits = the stream from the code's first bit; -1 = none
__device__ __forceinline__ int slow_sym(uint32_t bits, const uint16_t* count, const uint16_t* symbol, int nsym, uint32_t& used) {
    int code = 0, first = 0, index = 0;
#pragma unroll 1
    for (int len = 1; len <= 15; ++len) {
        code |= (int)((bits >> (len - 1)) & 1u);
        const int c = (int)count[len];
        if (code - c < first) {
            used = (uint32_t)len;
            const int at = index + (code - first);
            return at < nsym ? (int)symbol[at] : -1;
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// The span decode's tables: what build() made (symbol | code bits), widened so that a token costs one look-up and no arithmetic on
// the symbol.  Literal/length entry: code bits (4; 0 = a longer code, or none) | extra bits << 4 | value << 8 (the byte, or
// the length's base) | kind << 17 (0 literal, 1 length, 2 end of block, 3 no such symbol).  Distance entry: code bits | extra
// bits << 4 | base << 8 (0 = no such symbol).  RFC 1951's base/extra tables are regular: computed, not looked up.
__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t nbits) {
    if (sym < 256) return nbits | (sym << 8);
    if (sym == 256) return nbits | (2u << 17);
    const uint32_t s = sym - 257;
    if (s >= 29) return nbits | (3u << 17);
    const uint32_t le = (s < 8 || s == 28) ? 0u : (s >> 2) - 1u;
    const uint32_t base = s < 8 ? 3u + s : (s == 28 ? 258u : 3u + ((4u + (s & 3u)) << le));
    return nbits | (le << 4) | (base << 8) | (1u << 17);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t ds, uint32_t nbits) {
    if (ds >= 30) return nbits;
    const uint32_t de = ds < 4 ? 0u : (ds >> 1) - 1u;
    const uint32_t base = ds < 4 ? 1u + ds : 1u + ((2u + (ds & 1u)) << de);
    return nbits | (de << 4) | (base << 8);
}

// build() for the literal/length and the distance code of a deflate block at once, straight into the span decode's widened tables:
// the two codes share the four barriers, their one-lane parts (Kraft check, first codes, offsets per length) run on two
// waves side by side, and the 16-bit tables + their conversion fall away.  verdict[k]: 0 complete, > 0 incomplete, < 0
// over-subscribed (k = 0 literal/length, 1 distance); L.lcount / lsym / dcount / dsym as build() leaves them (the slow path's).
template <int NT>
__device__ void build2(Lds& L, int nlen, int ndist, uint32_t* ll, uint32_t* dl, int (&verdict_out)[2]) {
    static_assert(NT >= 352, "literal lanes [0, 288), distance lanes [320, 352)");
    __shared__ uint32_t cnt32[2][16];
    __shared__ uint16_t offs[2][16], next[2][16];
    __shared__ int verdict[2], usable[2];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (1u << LBITS); i += NT) ll[i] = 0;
    for (uint32_t i = tid; i < (1u << DBITS); i += NT) dl[i] = 0;
    if (tid < 32) cnt32[tid >> 4][tid & 15u] = 0;
    __syncthreads();
    const bool is_lit = tid < (uint32_t)nlen, is_dist = tid >= 320 && tid < 320u + (uint32_t)ndist;
    const uint8_t* length = L.lengths + (is_dist ? 288 : 0);
    const uint32_t sym = is_dist ? tid - 320u : tid, which = is_dist ? 1u : 0u;
    const uint32_t l = (is_lit || is_dist) ? (uint32_t)(length[sym] & 15) : 0u;
    if (is_lit || is_dist) atomicAdd(&cnt32[which][l], 1u);
    __syncthreads();
    if (tid == 0 || tid == 64) {  // (two waves)
        const uint32_t k = tid ? 1u : 0u;
        const int n = k ? ndist : nlen;
        uint16_t* count = k ? L.dcount : L.lcount;
        int left = 1;
        for (int q = 0; q <= 15; ++q) count[q] = (uint16_t)cnt32[k][q];
        for (int q = 1; q <= 15; ++q) {
            left <<= 1;
            left -= count[q];
            if (left < 0) break;
        }
        usable[k] = left >= 0 && count[0] != n;
        if (usable[k]) {
            offs[k][1] = 0;
            int code = 0;
            next[k][0] = 0;
            for (int q = 1; q <= 15; ++q) {
                code = (code + (q > 1 ? count[q - 1] : 0)) << 1;
                next[k][q] = (uint16_t)code;
                if (q < 15) offs[k][q + 1] = offs[k][q] + count[q];
            }
        }
        count[0] = (uint16_t)(count[0] == n ? 0xFFFF : count[0]);  // no codes at all: marked
        verdict[k] = left;
    }
    __syncthreads();
    if ((is_lit || is_dist) && l && usable[which]) {
        uint32_t rank = 0;
        {
            const uint32_t* lw = reinterpret_cast<const uint32_t*>(length);
            const uint32_t pat = l * 0x01010101u, whole = sym >> 2;
            auto equal_bytes = [&](uint32_t w) {
                const uint32_t x = w ^ pat;
                return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
            };
            for (uint32_t q = 0; q < whole; ++q) rank += (uint32_t)__popc(equal_bytes(lw[q]));
            if (sym & 3u) rank += (uint32_t)__popc(equal_bytes(lw[whole]) & ((1u << (8u * (sym & 3u))) - 1u));
        }
        (which ? L.dsym : L.lsym)[offs[which][l] + rank] = (uint16_t)sym;
        const uint32_t c = next[which][l] + rank;
        const uint32_t bits = which ? (uint32_t)DBITS : (uint32_t)LBITS;
        if (l <= bits) {
            const uint32_t rev = __brev(c) >> (32 - l);
            const uint32_t e = which ? dist_entry(sym, l) : lit_entry(sym, l);
            uint32_t* tab = which ? dl : ll;
            for (uint32_t k = rev; k < (1u << bits); k += 1u << l) tab[k] = e;
        }
    }
    __syncthreads();
    verdict_out[0] = verdict[0];
    verdict_out[1] = verdict[1];
}

// Tokens from bit `pos` on while they start before `limit`.  WRITE: literals -> ob[o...], matches -> list[m...], and win_first[w]
// = the lowest list index of a match whose first byte lies in window w of the text (the match stage's windows).  ll / dl: the
// widened tables above; L: the canonical tables behind them, for codes longer than the first-level look-up.
// (Tried on top of this and dropped, both measured with the timing build on the end-to-end job's blocks: remembering three token
//  boundaries per lane so that a re-decode which meets one takes the old result from there -- 192 us of rounds per block with it,
//  180 without: the bookkeeping ran on every token; and letting every lane decode 1 / 3 / 6 spans ahead of its own first, for a
//  start that is already in step -- 16.5 -> 14.6 / 11.0 / 7.4 rounds per block, 191 -> 197 / 233 / 306 us: a round's cost is
//  its decode, not its barriers.  One CODE per turn instead of one token -- a uniform short turn, the table chosen by a state flag,
//  so that a wave with literals and matches side by side does not run the long match path on every turn: 158 -> 182 us; four of
//  ten tokens of these blocks are matches, and each then costs two turns.)
template <bool WRITE>
__device__ __forceinline__ void decode_span(const uint32_t* pw, uint32_t total_bits, const Lds& L, const uint32_t* ll, const uint32_t* dl, int nlsym,
                                            int ndsym, uint32_t pos, uint32_t limit, uint8_t* ob, uint32_t o, uint32_t olen, unsigned long long* list,
                                            uint32_t m, uint32_t mcap, uint32_t& exit_pos, uint32_t& n_out, uint32_t& n_mat, uint32_t& flag,
                                            uint32_t* win_first = nullptr) {
    uint32_t out = 0, mat = 0, fl = 0;
    uint32_t last_win = 0xFFFFFFFFu;
    // The stream from `pos` on in a register: `have` bits of it in `buf`, the word behind them (`ahead`) fetched one refill early.
    // A token then costs its table look-ups and nothing else on the dependent path (a peek at the payload per code -- two
    // more LDS trips per literal, four per match -- was half of a decode).
    uint64_t buf = 0;
    uint32_t have = 0, wi = 0, ahead = 0;
    if (pos < total_bits) {
        wi = pos >> 5;
        const uint32_t sh = pos & 31u;
        buf = ((uint64_t)pw[wi] | ((uint64_t)pw[wi + 1] << 32)) >> sh;
        have = 64 - sh;
        wi += 2;
        ahead = pw[wi];
    }
    auto top_up = [&]() {  // at least 33 bits in hand
        if (have <= 32) {  // (a branch-free form that fetches the word behind on every turn: the same 158 us of rounds per block)
            buf |= (uint64_t)ahead << have;
            have += 32;
            ahead = pw[++wi];
        }
    };
    auto drop = [&](uint32_t n) {
        buf >>= n;
        have -= n;
        pos += n;
    };
    const uint32_t stop = min(limit, total_bits);  // (a token that starts at or behind the payload's end: below)
#pragma unroll 1
    while (pos < stop) {  // (every turn takes at least one bit)
        top_up();
        uint32_t e = ll[(uint32_t)buf & ((1u << LBITS) - 1u)];
        if ((e & 15u) == 0) {
            uint32_t used;
            const int sym = slow_sym((uint32_t)buf, L.lcount, L.lsym, nlsym, used);
            if (sym < 0) {
                fl = F_ERR_CODE;
                break;
            }
            e = lit_entry((uint32_t)sym, used);
        }
        const uint32_t nb = e & 15u, kind = e >> 17;
        if (kind == 0) {
            if (WRITE) {
                if (o + out < olen) ob[o + out] = (uint8_t)(e >> 8);
                else fl = F_ERR_CODE;
            }
            ++out;
            drop(nb);
            continue;
        }
        if (kind != 1) {
            if (kind == 2) {
                fl |= F_EOB;
                drop(nb);
            } else {
                fl = F_ERR_CODE;
            }
            break;
        }
        // (code <= 15 bits + <= 5 extra bits: inside the 33 in hand)
        const uint32_t le = (e >> 4) & 15u;
        const uint32_t len = ((e >> 8) & 511u) + ((uint32_t)(buf >> nb) & ((1u << le) - 1u));
        drop(nb + le);
        top_up();
        uint32_t d = dl[(uint32_t)buf & ((1u << DBITS) - 1u)];
        if ((d & 15u) == 0) {
            uint32_t used;
            const int ds = slow_sym((uint32_t)buf, L.dcount, L.dsym, ndsym, used);
            if (ds < 0) {
                fl = F_ERR_CODE;
                break;
            }
            d = dist_entry((uint32_t)ds, used);
        }
        if ((d >> 8) == 0) {
            fl = F_ERR_CODE;
            break;
        }
        const uint32_t dn = d & 15u, de = (d >> 4) & 15u;
        const uint32_t dist = (d >> 8) + ((uint32_t)(buf >> dn) & ((1u << de) - 1u));
        drop(dn + de);
        if (pos > total_bits) {
            fl = F_ERR_TRUNC;
            break;
        }
        if (WRITE) {
            if (m + mat < mcap) {
                list[m + mat] = (unsigned long long)(o + out) | ((unsigned long long)len << 20) | ((unsigned long long)dist << 32);
#if !defined(QD_INFLATE2_FULL_LIST)
                const uint32_t win = min((o + out) / (uint32_t)QD_INFLATE2_Q, 65536u / QD_INFLATE2_Q);
                if (win != last_win) {
                    atomicMin(&win_first[win], m + mat);
                    last_win = win;
                }
#endif
            } else {
                fl = F_ERR_CODE;
            }
        }
        out += len;
        ++mat;
    }
    if (fl == 0 && pos < limit) fl = F_ERR_TRUNC;  // (pos >= total_bits: the payload ended inside the span)
    exit_pos = pos;
    n_out = out;
    n_mat = mat;
    flag = fl;
}

template <int NT>
__global__ __launch_bounds__(NT) void inflate_bgzf_blocks2(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out,
                                                           int32_t* status, unsigned long long* matches, uint32_t mcap, uint32_t pay_words,
                                                           uint32_t* rounds_out, const uint32_t* expect_crc) {
    uint32_t rounds_used = 0, dblocks = 0;
    uint64_t tm[10] = {(uint64_t)wall_clock64(), 0, 0, 0, 0, 0, 0, 0, 0, 0};  // measurement: where a block's time goes (100 MHz ticks), summed over its deflate blocks
    auto stamp = [&](int k, uint64_t& since) {
        const uint64_t now = (uint64_t)wall_clock64();
        tm[k] += now - since;
        since = now;
    };
    uint64_t since = tm[0];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    constexpr int CRC_SW = NT >= 1024 ? 17 : (NT >= 512 ? 33 : 65);  // words per lane of the CRC stage: NT slices cover a 64 KiB block
    static_assert((size_t)NT * CRC_SW * 4 >= 65536, "the CRC stage's slices");
    Lds2<NT>& S = *reinterpret_cast<Lds2<NT>*>(lds_raw);
    Lds& L = S.t;
    uint32_t* pw = reinterpret_cast<uint32_t*>(lds_raw + ((sizeof(Lds2<NT>) + 15) & ~(size_t)15));  // the payload, pay_words + 4 words
    uint32_t* ow = pw + pay_words + 4;                                                          // the text, 64 KiB + 16 bytes
    uint8_t* ob = reinterpret_cast<uint8_t*>(ow);
    const uint32_t i = blockIdx.x, tid = threadIdx.x;
    if (i >= n_blocks) return;
    const qd_inflate_block blk = blocks[i];
    const uint32_t olen = blk.out_len, total_bits = blk.in_len * 8u;
    unsigned long long* list = matches + (size_t)i * mcap;
    int err = 0, last = 0;
    uint32_t opos = 0;
    if (olen > WINDOW || ((blk.in_len + 3) >> 2) > pay_words) err = QD_INFLATE_OVERRUN;
    // the payload -> LDS (zero behind it)
    for (uint32_t k = tid; k < pay_words + 4; k += NT) {
        uint32_t w = 0;
        const uint32_t at = 4 * k;
        if (!err && at < blk.in_len) {
            const uint8_t* p = comp + blk.in_off + at;
            const uint32_t left = blk.in_len - at;
            w = p[0];
            if (left > 1) w |= (uint32_t)p[1] << 8;
            if (left > 2) w |= (uint32_t)p[2] << 16;
            if (left > 3) w |= (uint32_t)p[3] << 24;
        }
        pw[k] = w;
    }
    __syncthreads();
    stamp(1, since);  // payload staged
    // (the headers are read from the staged copy as well: the wave-uniform reader's byte loads then cost an LDS trip, not a global one)
    Bits b{reinterpret_cast<const uint8_t*>(pw), 0, blk.in_len, 0, 0, 0, 0};
    fetch_ahead(b);
#pragma unroll 1
    for (int guard = 0; guard < 4096 && !last && !err; ++guard) {
        refill(b);
        last = (int)take(b, 1);
        const int type = (int)take(b, 2);
        if (b.cnt < 0) {
            err = QD_INFLATE_TRUNCATED;
            break;
        }
        if (type == 0) {  // stored: byte aligned LEN, NLEN, then LEN bytes
            const int drop = b.cnt & 7;
            b.buf >>= drop;
            b.cnt -= drop;
            refill(b);
            if (b.cnt < 32) {
                err = QD_INFLATE_TRUNCATED;
                break;
            }
            const uint32_t len = take(b, 16), nlen = take(b, 16);
            const uint32_t back = ((uint32_t)b.cnt >> 3) + b.ahead_bytes;
            b.pos -= back;
            b.buf = 0;
            b.cnt = 0;
            b.ahead = b.ahead_bytes = 0;
            if ((len ^ 0xFFFFu) != nlen) err = QD_INFLATE_BAD_STORED;
            else if (len > b.end - b.pos) err = QD_INFLATE_TRUNCATED;
            else if (len > olen - opos) err = QD_INFLATE_OVERRUN;
            else {
                for (uint32_t k = tid; k < len; k += NT) ob[opos + k] = b.p[b.pos + k];
                opos += len;
                b.pos += len;
                fetch_ahead(b);
                __syncthreads();
            }
            continue;
        }
        if (type != 1 && type != 2) {
            err = QD_INFLATE_BAD_TYPE;
            break;
        }
        int nlen = 288, ndist = 30;
        if (type == 1) {  // fixed codes
            for (uint32_t s = tid; s < 288; s += NT) L.lengths[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
            for (uint32_t s = tid; s < 30; s += NT) L.lengths[288 + s] = 5;
            __syncthreads();
        } else {  // dynamic codes: the code-length code first
            nlen = (int)take(b, 5) + 257;
            ndist = (int)take(b, 5) + 1;
            const int ncode = (int)take(b, 4) + 4;
            if (b.cnt < 0 || nlen > 286 || ndist > 30) {
                err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                break;
            }
            if (tid < 19) L.lengths[tid] = 0;
            __syncthreads();
            for (int k = 0; k < ncode; ++k) {
                refill(b);
                const uint32_t v = take(b, 3);
                if (tid == 0) L.lengths[CLORDER[k]] = (uint8_t)v;
            }
            __syncthreads();
            // (the code-length code's 7-bit table goes where the distance table will be; the span decode's literal table holds a table
            //  of GROUPS meanwhile: 10 bits of the stream -> up to four plain lengths at once, as many codes as fit -- the walk over the
            //  ~300 code lengths is one dependent look-up after the other, by every wave alike, and was a sixth of a block's time)
            static_assert(DBITS >= 7 && LBITS >= 10, "the header's two tables borrow the code tables' space");
            // (the code-length code: 19 symbols of at most 7 bits -- one wave builds its table from ballots, no barrier inside; build()'s four
            //  barriers and one lane's pass over the lengths were a fifth of the tables' time.  The code must be complete: Kraft sum 128.)
            if (tid < 64) {
                const uint32_t l = tid < 19 ? (uint32_t)L.lengths[tid] : 0u;
                uint32_t start = 0, mine = 0, kraft = 0, rank = 0;
#pragma unroll
                for (uint32_t len = 1; len <= 7; ++len) {
                    const uint64_t m = __ballot(l == len);
                    const uint32_t c = (uint32_t)__popcll(m);
                    if (l == len) {
                        mine = start;
                        rank = (uint32_t)__popcll(m & ((1ull << tid) - 1ull));
                    }
                    kraft += c << (7u - len);
                    start = (start + c) << 1;
                }
                if (kraft == 128u && l) {
                    const uint32_t rev = __brev(mine + rank) >> (32u - l);
                    const uint16_t e = (uint16_t)(tid | (l << 9));
                    for (uint32_t k = rev; k < 128u; k += 1u << l) L.dlut[k] = e;
                }
                if (tid == 0) S.ctl[0] = kraft;
            }
            __syncthreads();
            if (b.cnt < 0 || S.ctl[0] != 128u) {
                err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                break;
            }
            // entry: first symbol (5 bits) | its code's bits (3: 1..7, 0 = no code) | second, third, fourth symbol (4 bits each) | how many
            // plain lengths (3 bits) | their codes' bits together (4)
            for (uint32_t i = tid; i < 1024u; i += NT) {
                const uint32_t e1 = L.dlut[i & 127u];
                uint32_t e = 0;
                if (e1 >> 9) {
                    const uint32_t s1 = e1 & 511u, n1 = e1 >> 9;
                    e = s1 | (n1 << 5);
                    if (s1 < 16) {
                        uint32_t cnt = 1, used = n1;
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            const uint32_t ej = L.dlut[(i >> used) & 127u];
                            if (cnt != (uint32_t)j || !(ej >> 9) || (ej & 511u) >= 16 || used + (ej >> 9) > 10) continue;
                            e |= (ej & 511u) << (4 + 4 * j);
                            used += ej >> 9;
                            cnt = (uint32_t)j + 1;
                        }
                        e |= (cnt << 20) | (used << 23);
                    }
                }
                S.ll[i] = e;
            }
            __syncthreads();
            int idx = 0, prev = 0;
#pragma unroll 1
            while (idx < nlen + ndist && !err) {
                refill(b);
                int sym;
                const uint32_t e = uni(S.ll[(uint32_t)b.buf & 1023u]);
                const int cnt = (int)((e >> 20) & 7u);
                if (cnt >= 2 && idx + cnt <= nlen + ndist) {  // several plain lengths
                    const uint32_t used = (e >> 23) & 15u;
                    b.buf >>= used;
                    b.cnt -= (int)used;
                    if (b.cnt < 0) {
                        err = QD_INFLATE_TRUNCATED;
                        break;
                    }
                    if ((int)tid < cnt) {
                        const int at = idx + (int)tid;
                        L.lengths[at < nlen ? at : 288 + (at - nlen)] = (uint8_t)(tid ? (e >> (4 + 4 * tid)) & 15u : e & 31u);
                    }
                    idx += cnt;
                    prev = (int)((e >> (4 + 4 * (cnt - 1))) & 15u);
                    continue;
                }
                if ((e >> 5) & 7u) {
                    sym = (int)(e & 31u);
                    b.buf >>= ((e >> 5) & 7u);
                    b.cnt -= (int)((e >> 5) & 7u);
                } else {
                    sym = -1;
                }
                if (b.cnt < 0 || sym < 0 || sym > 18) {
                    err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_CODE;
                    break;
                }
                int rep = 1, v = sym;
                if (sym == 16) {
                    if (idx == 0) {
                        err = QD_INFLATE_BAD_TABLE;
                        break;
                    }
                    v = prev;
                    rep = 3 + (int)take(b, 2);
                } else if (sym == 17) {
                    v = 0;
                    rep = 3 + (int)take(b, 3);
                } else if (sym == 18) {
                    v = 0;
                    rep = 11 + (int)take(b, 7);
                }
                if (b.cnt < 0 || idx + rep > nlen + ndist) {
                    err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                    break;
                }
                for (int k = (int)tid; k < rep; k += NT) {
                    const int at = idx + k;
                    L.lengths[at < nlen ? at : 288 + (at - nlen)] = (uint8_t)v;
                }
                idx += rep;
                prev = v;
            }
            if (err) break;
            __syncthreads();
            if (L.lengths[256] == 0) {
                err = QD_INFLATE_BAD_TABLE;
                break;
            }
        }
        stamp(8, since);  // block header, code lengths
        int r2[2];
        build2<NT>(L, nlen, ndist, S.ll, S.dl, r2);
        const int lit_codes = L.lcount[0] == 0xFFFF ? 0 : nlen - (int)L.lcount[0];
        const int dist_codes = L.dcount[0] == 0xFFFF ? 0 : ndist - (int)L.dcount[0];
        if (type == 2 && (r2[0] < 0 || (r2[0] > 0 && lit_codes != 1) || r2[1] < 0 || (r2[1] > 0 && dist_codes > 1))) {
            err = QD_INFLATE_BAD_TABLE;
            break;
        }
        stamp(2, since);  // the two tables
        // ---- the block's symbols: 256 spans, guessed starts, rounds until the chain from lane 0 is confirmed up to the end-of-block symbol
        const uint32_t bitpos = 8u * (b.pos - b.ahead_bytes) - (uint32_t)b.cnt;  // (cnt >= 0: checked above)
        const uint32_t rest = total_bits > bitpos ? total_bits - bitpos : 0;
        uint32_t span = (rest + NT - 1) / NT;
        if (span < 64) span = 64;
        const uint32_t my_limit = tid == NT - 1 ? 0xFFFFFFF0u : bitpos + (tid + 1) * span;
        S.start[tid] = bitpos + tid * span;
        __syncthreads();
        uint32_t eob_lane = NT;
        uint32_t ex = 0, no = 0, nm = 0, fl = 0, decoded_from = 0xFFFFFFFFu;
        // A round: the lanes whose start moved decode their spans again; then every lane compares its start with its predecessor's
        // exit (a shuffle inside the wave, LDS across waves), the waves post their first lane that is out of step and their first
        // that met the end, and every lane takes the minimum -- two barriers a round (r04 first had four, and atomics).
        uint32_t st = S.start[tid];
        uint32_t* const wave_first = S.start_sum;  // [0, NT/64): out of step, [NT/64, 2 NT/64): ended (the scan below uses the space afterwards)
#pragma unroll 1
        for (int round = 0; round <= QD_INFLATE2_MAX_ROUNDS; ++round) {
            // (a lane whose start did not move keeps what it found: the confirming rounds decode only the spans that still change)
            if (st != decoded_from) {
                decode_span<false>(pw, total_bits, L, S.ll, S.dl, lit_codes, dist_codes, st, my_limit, ob, 0, olen, list, 0, mcap, ex, no, nm, fl);
                decoded_from = st;
            }
            const uint32_t lane = tid & 63u, wave = tid >> 6;
            if (lane == 63) S.exitp[tid] = ex;
            __syncthreads();
            uint32_t prev_exit = (uint32_t)__shfl_up((int)ex, 1, 64);
            if (lane == 0) prev_exit = tid ? S.exitp[tid - 1] : st;
            const uint64_t off = __ballot(st != prev_exit), ended = __ballot(fl != 0);
            if (lane == 0) {
                wave_first[wave] = off ? wave * 64u + (uint32_t)__builtin_ctzll(off) : (uint32_t)NT;
                wave_first[NT / 64 + wave] = ended ? wave * 64u + (uint32_t)__builtin_ctzll(ended) : (uint32_t)NT;
            }
            __syncthreads();
            uint32_t confirmed = NT, ender = NT;  // lanes [0, confirmed) are the sequential decode; ender: the first lane that ends the block or fails
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                confirmed = min(confirmed, wave_first[w]);
                ender = min(ender, wave_first[NT / 64 + w]);
            }
            if (ender < confirmed) {
                eob_lane = ender;
                break;
            }
            if (confirmed >= NT) break;  // every lane confirmed and none met the end: the payload ended first
            st = prev_exit;
            ++rounds_used;
        }
        S.start[tid] = st;
        S.exitp[tid] = ex;
        S.nout[tid] = no;
        S.nmat[tid] = nm;
        S.flag[tid] = fl;
        __syncthreads();
        ++rounds_used;
        ++dblocks;
        stamp(3, since);  // rounds
        if (eob_lane >= NT) {
            err = QD_INFLATE_TRUNCATED;
            break;
        }
        if (!(S.flag[eob_lane] & F_EOB) || (S.flag[eob_lane] & (F_ERR_CODE | F_ERR_TRUNC))) {
            err = (S.flag[eob_lane] & F_ERR_TRUNC) ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_CODE;
            break;
        }
        // where every span's bytes and matches go: exclusive sums over the spans up to the one that ends the block (wave scans, then
        // the waves' totals; one lane summing NT counts took ~20 us of a block's ~500)
        {
            const uint32_t lane = tid & 63u, wave = tid >> 6;
            uint32_t vo = tid <= eob_lane ? S.nout[tid] : 0u, vm = tid <= eob_lane ? S.nmat[tid] : 0u;
            const uint32_t o0 = vo, m0 = vm;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t yo = __shfl_up(vo, d, 64), ym = __shfl_up(vm, d, 64);
                if (lane >= (uint32_t)d) {
                    vo += yo;
                    vm += ym;
                }
            }
            __syncthreads();
            if (lane == 63) {
                S.start_sum[wave] = vo;
                S.start_sum[NT / 64 + wave] = vm;
            }
            __syncthreads();
            uint32_t bo = opos, bm = 0, to = opos, tmm = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                const uint32_t so = S.start_sum[w], sm = S.start_sum[NT / 64 + w];
                if ((uint32_t)w < wave) {
                    bo += so;
                    bm += sm;
                }
                to += so;
                tmm += sm;
            }
            S.nout[tid] = bo + vo - o0;
            S.nmat[tid] = bm + vm - m0;
            if (tid == 0) {
                S.ctl[2] = to;
                S.ctl[3] = tmm;
            }
        }
        __syncthreads();
        const uint32_t new_opos = S.ctl[2], n_matches = S.ctl[3];
        tm[7] += n_matches;
        if (new_opos > olen) {
            err = QD_INFLATE_OVERRUN;
            break;
        }
        if (n_matches > mcap) {
            err = QD_INFLATE_OVERRUN;  // (more matches than the list holds: the host inflates this run)
            break;
        }
        constexpr uint32_t N_WIN = 65536 / QD_INFLATE2_Q + 2;
        if (tid < N_WIN) S.win_first[tid] = n_matches;
        __syncthreads();
        if (tid <= eob_lane) {
            uint32_t ex, no, nm, fl;
            decode_span<true>(pw, total_bits, L, S.ll, S.dl, lit_codes, dist_codes, S.start[tid], my_limit, ob, S.nout[tid], olen, list, S.nmat[tid], mcap,
                              ex, no, nm, fl, S.win_first);
        }
        __threadfence_block();
        __syncthreads();
        stamp(4, since);  // scan + write pass
        // The matches.  Walking them in text order, one dependent LDS read -> write trip each, took 1.2 of a block's 1.8 ms (6 800
        // matches per block of libdeflate-1 fastq), whatever was batched.  Instead every byte that a match produces gets a
        // parent -- the byte `distance` before it (true for overlapping matches too) -- and the chains are shortened by
        // pointer jumping: a byte whose parent is final takes its value and becomes final, another one adopts its parent's
        // parent; every round halves the chains (a name copied from record to record through a whole block: 8 rounds).
        // A window of the text (QD_INFLATE2_Q positions: a quarter by default) at a time: that many 16-bit parents lie in LDS beside the text.
        {
            constexpr uint32_t Q = QD_INFLATE2_Q;
            static_assert(Q <= 65536, "the parents are 16-bit positions inside the window");
            constexpr int MATCH_ROUNDS = 20;  // >= log2(Q) + 3: a chain of Q links is resolved in log2(Q) + 1 rounds
            // (volatile: a lane reads a parent's state, then -- only if that is final -- its byte; the two reads stay in that order,
            //  as the writes "byte, then final" of the lane that owns the parent do)
            volatile uint16_t* par = reinterpret_cast<volatile uint16_t*>(ow + 16384 + 4);
            volatile uint8_t* vob = ob;
            if (tid == 0) S.ctl[4] = S.ctl[5] = S.ctl[6] = S.ctl[7] = 0;
            __syncthreads();
#pragma unroll 1
            for (uint32_t qb = (opos / Q) * Q; qb < new_opos; qb += Q) {
                const uint32_t lo = max(qb, opos), hi = min(qb + Q, new_opos);  // (positions below `lo` are final)
                // the matches that can reach into this window: those that start in it, and the last one before them.  (The
                // list is in text order; every match starts in exactly one window, so each is also checked once.)
                uint32_t m_lo = n_matches, m_hi = n_matches;
                for (uint32_t w = qb / Q; w < N_WIN; ++w) {
                    const uint32_t f = S.win_first[w];
                    if (f < m_lo) m_lo = f;
                    if (w > qb / Q && f < m_hi) m_hi = f;
                }
                if (m_lo > 0) --m_lo;
#if defined(QD_INFLATE2_FULL_LIST) /* A/B: every window walks the whole list */
                m_lo = 0, m_hi = n_matches;
#endif
                // (fetching a lane's first match ahead of this and setting the parents four per store changed nothing: 86 -> 87 us per block)
                for (uint32_t p = lo + tid; p < hi; p += NT) par[p - qb] = (uint16_t)p;
                __syncthreads();
                for (uint32_t m = m_lo + tid; m < m_hi; m += NT) {
                    const unsigned long long e = list[m];
                    const uint32_t d = (uint32_t)e & 0xFFFFFu, len = ((uint32_t)e >> 20), dist = (uint32_t)(e >> 32);
                    if (dist == 0 || dist > d || len > olen - d) {
                        S.ctl[4] = 1;
                        continue;
                    }
                    const uint32_t a = max(d, lo), z = min(d + len, hi);
                    for (uint32_t p = a; p < z; ++p) par[p - qb] = (uint16_t)(p - dist);
                }
                __syncthreads();
                // This lane's positions of the window: qb + tid + k * NT; `pend` = those not final yet, kept over the rounds, so a
                // round touches only them (most bytes are literals or final after the first round; reading every position's parent to
                // find that out was most of a round).  A lane takes its pending positions in rising order, as the whole workgroup
                // does: a parent made final earlier in the round already counts, which settles most chains within two rounds (all
                // reads of a batch of positions before their writes -- fewer dependent trips -- needed six and was slower).
                constexpr int K = (int)(Q / NT);
                static_assert(Q % NT == 0 && K <= 32, "a lane's share of a window");
                uint32_t pend = 0;
#pragma unroll 4
                for (int k = 0; k < K; ++k) {
                    const uint32_t p = qb + tid + (uint32_t)k * NT;
                    if (p >= lo && p < hi && par[p - qb] != p) pend |= 1u << k;
                }
#pragma unroll 1
                for (int round = 0; round < MATCH_ROUNDS; ++round) {
#pragma unroll 1
                    for (uint32_t left = pend; left; left &= left - 1u) {
                        const uint32_t k = (uint32_t)__builtin_ctz(left), p = qb + tid + k * NT;
                        const uint32_t q = par[p - qb];
                        if (q < lo || par[q - qb] == q) {
                            vob[p] = vob[q];
                            par[p - qb] = (uint16_t)p;
                            pend &= ~(1u << k);
                        } else {
                            par[p - qb] = par[q - qb];
                        }
                    }
                    uint32_t* flag = &S.ctl[5];  // three flags in turn: the one of round r + 2 is cleared behind the barrier of round r
                    if (pend) flag[round % 3] = 1;
                    __syncthreads();
                    const uint32_t more = flag[round % 3];
                    if (tid == 0) flag[(round + 2) % 3] = 0;
                    if (!more) break;
                    if (round == MATCH_ROUNDS - 1 && tid == 0) S.ctl[4] = 1;  // (chains halve every round: cannot happen; a block that did is not shipped)
                }
            }
        }
        __syncthreads();
        stamp(5, since);  // matches
        if (S.ctl[4]) {
            err = QD_INFLATE_BAD_DISTANCE;
            break;
        }
        opos = new_opos;
        // the next deflate block's header starts where the end-of-block symbol ended
        {
            const uint32_t np = S.exitp[eob_lane];
            __syncthreads();
            b.pos = np >> 3;
            b.buf = 0;
            b.cnt = 0;
            b.ahead = b.ahead_bytes = 0;
            fetch_ahead(b);
            refill(b);
            refill(b);
            const int drop = (int)(np & 7u);
            b.buf >>= drop;
            b.cnt -= drop;
        }
    }
    if (!err && !last) err = QD_INFLATE_BAD_TYPE;
    if (!err && opos != olen) err = QD_INFLATE_LENGTH;
    __syncthreads();
    // the block's CRC-32 against its trailer while the text is still in LDS (the tables go where the match stage kept its parents)
    if (expect_crc) {
        uint32_t* crc_t = ow + 16384 + 4;
        qdcrc::stage_tables<NT>(crc_t);
        __syncthreads();
        const uint32_t crc = qdcrc::crc32_lds<NT, CRC_SW>(ow, err ? 0u : olen, crc_t, crc_t + 1024);
        if (!err && crc != expect_crc[i]) err = QD_INFLATE_CRC;
    }
    if (!err) {  // the text leaves: bytes up to the first 4-byte boundary of the destination, dwords, the tail
        uint8_t* const o = out + blk.out_off;
        const uint32_t head = min(olen, (uint32_t)((4u - ((uintptr_t)o & 3u)) & 3u));
        if (tid < head) o[tid] = ob[tid];
        const uint32_t nd = (olen - head) >> 2;
        for (uint32_t k = tid; k < nd; k += NT) {
            const uint32_t at = head + 4 * k, wi = at >> 2;
            *reinterpret_cast<uint32_t*>(o + at) = __builtin_amdgcn_alignbyte(ow[wi + 1], ow[wi], at & 3u);
        }
        for (uint32_t k = head + 4 * nd + tid; k < olen; k += NT) o[k] = ob[k];
    }
    stamp(6, since);  // flush
    if (tid == 0) status[i] = err;
    if (tid == 0 && rounds_out) {
        rounds_out[8 * i] = rounds_used | (dblocks << 16);
        for (int k = 1; k < 8; ++k) rounds_out[8 * i + k] = (uint32_t)(k == 2 ? tm[2] + tm[8] : tm[k]);  // (2: headers + tables)
    }
#if defined(QD_INFLATE_TIMING) /* measurement build: where the blocks' time goes, summed over a process's launches */
    if (tid == 0) {
        atomicAdd(&g_inflate_ticks[0], (unsigned long long)rounds_used);
        for (int k = 1; k < 9; ++k) atomicAdd(&g_inflate_ticks[k], (unsigned long long)tm[k]);
        atomicAdd(&g_inflate_ticks[9], 1ull);
        atomicAdd(&g_inflate_ticks[10], (unsigned long long)dblocks);
    }
#endif
}
}  // namespace v2

}  // namespace

// One workgroup of 64 lanes (one wave) per block.  blocks[i]: payload (raw deflate) at comp + in_off, in_len bytes
// -> out + out_off, exactly out_len bytes.  status[i] = 0 or a QD_INFLATE_* code (output of a failed block: unspecified).
__global__ __launch_bounds__(64) void inflate_bgzf_blocks(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks,
                                                          uint8_t* out, int32_t* status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    Lds& L = *reinterpret_cast<Lds*>(lds_raw);
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_blocks) return;
    const qd_inflate_block blk = blocks[i];
    const uint32_t olen = blk.out_len;
    int err = 0, last = 0;
    uint32_t opos = 0;
    if (olen > WINDOW) err = QD_INFLATE_OVERRUN;
    uint8_t* const o = out + blk.out_off;  // the block's text is written in place: matches read it back from there
    Bits b{comp + blk.in_off, 0, blk.in_len, 0, 0, 0, 0};
    fetch_ahead(b);
#pragma unroll 1
    for (int guard = 0; guard < 4096 && !last && !err; ++guard) {  // a 64 KiB block holds far fewer deflate blocks
        refill(b);
        last = (int)take(b, 1);
        const int type = (int)take(b, 2);
        if (b.cnt < 0) {
            err = QD_INFLATE_TRUNCATED;
            break;
        }
        if (type == 0) {  // stored: byte aligned LEN, NLEN, then LEN bytes
            const int drop = b.cnt & 7;
            b.buf >>= drop;
            b.cnt -= drop;
            refill(b);
            if (b.cnt < 32) {
                err = QD_INFLATE_TRUNCATED;
                break;
            }
            const uint32_t len = take(b, 16), nlen = take(b, 16);
            // whole bytes still in the bit buffer belong to the stored data: give them back
            const uint32_t back = ((uint32_t)b.cnt >> 3) + b.ahead_bytes;  // ... and the look-ahead word too
            b.pos -= back;
            b.buf = 0;
            b.cnt = 0;
            b.ahead = b.ahead_bytes = 0;
            if ((len ^ 0xFFFFu) != nlen) err = QD_INFLATE_BAD_STORED;
            else if (len > b.end - b.pos) err = QD_INFLATE_TRUNCATED;
            else if (len > olen - opos) err = QD_INFLATE_OVERRUN;
            else {
                for (uint32_t k = lane; k < len; k += 64) o[opos + k] = b.p[b.pos + k];
                opos += len;
                b.pos += len;
                fetch_ahead(b);
            }
        } else if (type == 1 || type == 2) {
            int nlen = 288, ndist = 30;
            if (type == 1) {  // fixed codes
                for (uint32_t s = lane; s < 288; s += 64) L.lengths[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
                for (uint32_t s = lane; s < 30; s += 64) L.lengths[288 + s] = 5;
                __syncthreads();
            } else {  // dynamic codes: the code-length code first
                nlen = (int)take(b, 5) + 257;
                ndist = (int)take(b, 5) + 1;
                const int ncode = (int)take(b, 4) + 4;
                if (b.cnt < 0 || nlen > 286 || ndist > 30) {
                    err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                    break;
                }
                if (lane < 19) L.lengths[lane] = 0;
                __syncthreads();
                for (int k = 0; k < ncode; ++k) {
                    refill(b);
                    const uint32_t v = take(b, 3);
                    if (lane == 0) L.lengths[CLORDER[k]] = (uint8_t)v;
                }
                __syncthreads();
                if (b.cnt < 0 || build(L.lengths, 19, L.lcount, L.lsym, L.llut, 7, 9, lane) != 0) {  // must be complete
                    err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                    break;
                }
                // the code lengths of both codes, run-length coded
                int idx = 0, prev = 0;
#pragma unroll 1
                while (idx < nlen + ndist && !err) {
                    refill(b);
                    int sym;
                    const uint32_t e = uni(L.llut[(uint32_t)b.buf & 127u]);
                    if (e >> 9) {
                        sym = e & 511;
                        b.buf >>= (e >> 9);
                        b.cnt -= (e >> 9);
                    } else {
                        sym = -1;  // code-length codes are at most 7 bits: a miss is an invalid code
                    }
                    if (b.cnt < 0 || sym < 0 || sym > 18) {
                        err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_CODE;
                        break;
                    }
                    int rep = 1, v = sym;
                    if (sym == 16) {
                        if (idx == 0) {
                            err = QD_INFLATE_BAD_TABLE;
                            break;
                        }
                        v = prev;
                        rep = 3 + (int)take(b, 2);
                    } else if (sym == 17) {
                        v = 0;
                        rep = 3 + (int)take(b, 3);
                    } else if (sym == 18) {
                        v = 0;
                        rep = 11 + (int)take(b, 7);
                    }
                    if (b.cnt < 0 || idx + rep > nlen + ndist) {
                        err = b.cnt < 0 ? QD_INFLATE_TRUNCATED : QD_INFLATE_BAD_TABLE;
                        break;
                    }
                    // lengths of the literal/length code at [320-array 0..nlen), of the distance code behind index 288
                    for (int k = (int)lane; k < rep; k += 64) {
                        const int at = idx + k;
                        L.lengths[at < nlen ? at : 288 + (at - nlen)] = (uint8_t)v;
                    }
                    idx += rep;
                    prev = v;
                }
                if (err) break;
                __syncthreads();
                if (L.lengths[256] == 0) {  // no end-of-block code
                    err = QD_INFLATE_BAD_TABLE;
                    break;
                }
            }
            // the two codes of this deflate block (the code-length code's tables are overwritten here)
            int r = build(L.lengths, nlen, L.lcount, L.lsym, L.llut, LBITS, 9, lane);
            const int lit_codes = L.lcount[0] == 0xFFFF ? 0 : nlen - (int)L.lcount[0];
            if (type == 2 && (r < 0 || (r > 0 && lit_codes != 1))) {  // incomplete only for a single code
                err = QD_INFLATE_BAD_TABLE;
                break;
            }
            r = build(L.lengths + 288, ndist, L.dcount, L.dsym, L.dlut, DBITS, 5, lane);
            const int dist_codes = L.dcount[0] == 0xFFFF ? 0 : ndist - (int)L.dcount[0];
            if (type == 2 && (r < 0 || (r > 0 && dist_codes > 1))) {  // (the fixed distance code has 30 of 32 codes)
                err = QD_INFLATE_BAD_TABLE;
                break;
            }
            err = codes(b, L, o, opos, olen, lane, lit_codes, dist_codes);
        } else {
            err = QD_INFLATE_BAD_TYPE;
        }
    }
    if (!err && !last) err = QD_INFLATE_BAD_TYPE;
    if (!err && opos != olen) err = QD_INFLATE_LENGTH;
    if (lane == 0) status[i] = err;
}

// dynamic LDS of the second form's workgroups for a launch whose longest payload is max_in_len bytes: tables | payload | text | parents
template <int NT>
static size_t inflate2_lds(uint32_t max_in_len) {
    return ((sizeof(v2::Lds2<NT>) + 15) & ~(size_t)15) + ((size_t)((max_in_len + 3) / 4) + 4) * 4 + 65536 + 16 + 2 * QD_INFLATE2_Q;
}
// The kernel's statically declared LDS (build()'s counters) comes on top of the dynamic part: a launch 60 bytes under the CU's
// 160 KB by its dynamic size alone was refused by the runtime.
constexpr size_t STATIC_LDS = 512;
// (what the narrowest instantiation needs: a launch that fits no form is the one-wave kernel's)
size_t qd_inflate2_lds(uint32_t max_in_len) { return inflate2_lds<512>(max_in_len) + STATIC_LDS; }

template <int NT>
static hipError_t launch_inflate2(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out, int32_t* status,
                                  unsigned long long* matches, uint32_t matches_per_block, uint32_t max_in_len, hipStream_t st, uint32_t* rounds_out,
                                  const uint32_t* expect_crc) {
    const uint32_t pay_words = (max_in_len + 3) / 4;
    const size_t lds = inflate2_lds<NT>(max_in_len);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v2::inflate_bgzf_blocks2<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(v2::inflate_bgzf_blocks2<NT>, dim3(n_blocks), dim3(NT), lds, st, comp, blocks, n_blocks, out, status, matches,
                       matches_per_block, pay_words, rounds_out, expect_crc);
#if defined(QD_INFLATE_TIMING)
    {
        unsigned long long t[11];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(v2::g_inflate_ticks), sizeof t);
        static const char* names[9] = {"", "stage", "tables", "rounds", "scan+write", "matches", "flush", "", "headers"};
        const double nb = t[9] ? (double)t[9] : 1.0;
        fprintf(stderr, "inflate_bgzf_blocks2<%d> per block over %llu blocks so far: %.2f deflate blocks, %.2f rounds, %.0f matches;", NT, t[9], (double)t[10] / nb,
                (double)t[0] / nb, (double)t[7] / nb);
        for (int k = 1; k < 9; ++k)
            if (k != 7) fprintf(stderr, " %s %.1f us", names[k], (double)t[k] / nb / 100.0);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

// 1 024 lanes per block while the launch's longest payload leaves room for their bookkeeping in LDS (up to ~40 KB: every block of fastq
// text bgzip makes), 512 beyond that (up to ~52 KB); QD_INFLATE2_THREADS fixes the count (measurement builds).
hipError_t qd_launch_inflate2(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out, int32_t* status,
                              unsigned long long* matches, uint32_t matches_per_block, uint32_t max_in_len, hipStream_t st, uint32_t* rounds_out,
                              const uint32_t* expect_crc) {
    if (n_blocks == 0) return hipSuccess;
#if QD_INFLATE2_THREADS == 1024 || QD_INFLATE2_THREADS == 0
    if (inflate2_lds<1024>(max_in_len) + STATIC_LDS <= 160 * 1024)
        return launch_inflate2<1024>(comp, blocks, n_blocks, out, status, matches, matches_per_block, max_in_len, st, rounds_out, expect_crc);
#endif
#if QD_INFLATE2_THREADS == 256
    if (inflate2_lds<256>(max_in_len) + STATIC_LDS <= 160 * 1024)
        return launch_inflate2<256>(comp, blocks, n_blocks, out, status, matches, matches_per_block, max_in_len, st, rounds_out, expect_crc);
#endif
    if (inflate2_lds<512>(max_in_len) + STATIC_LDS > 160 * 1024) return hipErrorInvalidValue;
    return launch_inflate2<512>(comp, blocks, n_blocks, out, status, matches, matches_per_block, max_in_len, st, rounds_out, expect_crc);
}

hipError_t qd_launch_inflate(const uint8_t* comp, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out,
                             int32_t* status, hipStream_t st) {
    if (n_blocks == 0) return hipSuccess;
    // per launch: the attribute belongs to the current device, and a process may run inflaters on several
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(inflate_bgzf_blocks),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(inflate_bgzf_blocks, dim3(n_blocks), dim3(64), sizeof(Lds), st, comp, blocks, n_blocks, out, status);
    return hipGetLastError();
}
