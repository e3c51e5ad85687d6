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

#include "quade_inflate.h"

namespace {

constexpr int LBITS = 10, DBITS = 8;         // first-level lookup widths
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

// Tables of one Huffman code from n code lengths (lane 0 writes; the wave waits).  Returns 0 for a complete code,
// > 0 incomplete, < 0 over-subscribed.  lut entries: symbol | length << shift for codes of at most `bits` bits.
__device__ int build(const uint8_t* length, int n, uint16_t* count, uint16_t* symbol, uint16_t* lut, int bits, int shift,
                     uint32_t lane) {
    for (uint32_t i = lane; i < (1u << bits); i += 64) lut[i] = 0;
    __syncthreads();
    int left = 1;
    if (lane == 0) {
        uint16_t offs[16], next[16];
        for (int l = 0; l <= 15; ++l) count[l] = 0;
        for (int s = 0; s < n; ++s) ++count[length[s] & 15];
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) break;
        }
        if (left >= 0 && count[0] != n) {
            offs[1] = 0;
            int code = 0;
            next[0] = 0;
            for (int l = 1; l <= 15; ++l) {
                code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
                next[l] = (uint16_t)code;
                if (l < 15) offs[l + 1] = offs[l] + count[l];
            }
#pragma unroll 1
            for (int s = 0; s < n; ++s) {
                const int l = length[s] & 15;
                if (!l) continue;
                symbol[offs[l]++] = (uint16_t)s;
                const uint32_t c = next[l]++;
                if (l <= bits) {  // the stream carries codes MSB first inside an LSB-first bit order: index by the reversed code
                    const uint32_t rev = __brev(c) >> (32 - l);
                    const uint16_t e = (uint16_t)(s | (l << shift));
                    for (uint32_t k = rev; k < (1u << bits); k += 1u << l) lut[k] = e;
                }
            }
        }
        count[0] = (uint16_t)(count[0] == n ? 0xFFFF : count[0]);  // no codes at all: marked
    }
    // every lane needs the verdict: through LDS (lengths[] is free for that after the build; use lut-independent slot)
    __shared__ int verdict;
    if (lane == 0) verdict = left;
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
