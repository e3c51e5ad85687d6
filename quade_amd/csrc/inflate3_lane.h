// The third inflater's token decoder: ONE LANE decodes one DEFLATE stream, symbol after symbol, 64 streams per wave in lock step.
//
// What it replaces: the gunzip inside pyFastq.FastqReader (src/Quade.py:203-206, 234-236) -- as quade_inflate.hip does, but
// without decoding anything twice.  The second form cuts a block's bits into spans and lets 1 024 lanes decode from guessed
// positions until their chain synchronises: every span is decoded ~10 times (~340 lane-instructions per byte
// of text).  Here a stream's Huffman decode stays the serial chain it is -- one table look-up per code on the dependent path --
// and the parallelism is ACROSS streams: a launch holds thousands of BGZF blocks (or thousands of stretches of one gzip member
// that start at a deflate block), each lane owns one, and a wave instruction advances 64 of them.  What a lane produces is not
// text but TOKENS (16-bit slots: a literal byte, or a match as length slot + distance slot), because copying matches is the
// part that a whole workgroup does well and a lane does badly; the resolve kernels (quade_inflate3.hip) turn tokens into text.
//
// A lane's tables live in LDS (its slice of the workgroup's dynamic LDS, an odd number of dwords apart so that equal indices of
// different lanes fall into different banks):
//     lit[2^LB]   literal/length code, indexed by the next LB bits of the stream      entry: code bits | kind << 4 | value << 6
//     dst[2^DB]   distance code, indexed by the next DB bits                          (kind 0 literal, 1 length / distance symbol,
//     lng[NLONG]  the symbols of both codes whose codes are longer, in code order            2 end of block, 3 + zero bits: longer code or none)
// A code longer than the first-level index is found the canonical way: the next 15 bits, MSB first, are compared with the
// left-aligned upper limits of the code lengths LB+1 .. 15 (kept in registers), the symbol is lng[base[len] + code].
// One CODE per turn: a match takes two turns (length code + extra bits, then distance code + extra bits), so that the lanes of
// a wave run one short body whatever they hold (profiles/r05_inflate3_*: what a turn costs).
//
// This header compiles for the host too (plain C++): tests/native/inflate3_lane_test.cpp runs the lane against zlib.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define QD3_HD __host__ __device__ __forceinline__
#else
#define QD3_HD inline
#endif

#ifndef QD_INFLATE_TRUNCATED
#define QD_INFLATE_TRUNCATED 1
#define QD_INFLATE_BAD_TYPE 2
#define QD_INFLATE_BAD_STORED 3
#define QD_INFLATE_BAD_TABLE 4
#define QD_INFLATE_BAD_CODE 5
#define QD_INFLATE_BAD_DISTANCE 6
#define QD_INFLATE_OVERRUN 7
#define QD_INFLATE_LENGTH 8
#define QD_INFLATE_CRC 9
#define QD_INFLATE_TABLE_SPACE 10
#define QD_INFLATE_TOKEN_SPACE 11
#define QD_INFLATE_CHAIN 12
#endif

namespace qd3 {

enum : uint32_t { ST_LIT = 0, ST_DIST = 1, ST_STORED = 2, ST_HEADER = 3, ST_DONE = 4 };
enum : uint32_t { TOK_MATCH = 0x8000u };  // slot: literal byte | TOK_MATCH + (length - 3), followed by a slot distance - 1 (< 0x8000)

// LB / DB: bits of the two first-level tables; NLONG: symbols with longer codes (both codes together); DBYTE: the distance table's
// entries are bytes (bits | symbol << 3) instead of 16-bit ones; CHUNKS: 16-byte chunks of its stream a lane keeps in the wave's
// ring; TURNS: turns between two top-ups of the ring.
template <int LB_, int DB_, int NLONG_, bool DBYTE_ = false, int CHUNKS_ = 16, int TURNS_ = 24>
struct Cfg {
    static constexpr int LB = LB_, DB = DB_, NLONG = NLONG_;
    static constexpr bool DBYTE = DBYTE_;
    static constexpr int LIT_N = 1 << LB, DST_N = 1 << DB;
    static constexpr int DST_HW = DBYTE ? DST_N / 2 : DST_N;                  // 16-bit units the distance table takes
    static constexpr int LANE_DW = ((LIT_N + DST_HW + NLONG + 1) / 2) | 1;    // dwords of LDS per lane (odd)
    static constexpr int NL = 15 - LB, ND = 15 - DB;                          // code lengths behind the first level
    static constexpr int NP = 15 - DB;  // ... of either code (DB <= LB): lengths DB + 1 .. 15, the two codes' limits and bases packed in pairs
    static constexpr int RING_CHUNKS = CHUNKS_;
    static constexpr int RING_DW = RING_CHUNKS * 256;  // dwords of a wave's ring: slot j = [256 j, 256 (j + 1)), lane l's chunk at + 4 l (what one LDS-DMA writes)
    // A turn moves at most one dword into buf and looks at the one behind it: when a round starts, at least 4 CHUNKS - 3 - TURNS
    // dwords in front of the lane have landed (everything the top-up before the last round asked for), and it takes TURNS + 1.
    static constexpr int ROUND_TURNS = TURNS_;
    static_assert((CHUNKS_ & (CHUNKS_ - 1)) == 0 && 2 * TURNS_ + 4 <= 4 * CHUNKS_, "ring size against the turns of a round");
    static_assert(DB >= 6 && (DST_HW + NLONG) * 2 >= 128, "the code-length code's 128-byte table borrows the space of the distance table (and of the long symbols behind it)");
    static_assert(LB >= DB && LB <= 11, "first-level widths");
    static_assert(!DBYTE || DB <= 7, "a byte entry holds three bits of code length");
};
constexpr int LENS_DW = 40;  // a lane's scratch of code lengths, a nibble each: literal/length [0, 288), distance [288, 320)

typedef uint32_t u32x4 __attribute__((vector_size(16)));

QD3_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// what the host says about a unit, and what the lane says back
struct Unit {
    const uint32_t* base;  // 16-byte aligned; the positions below are bits from here; wend dwords from here may be read
    uint64_t bit_start;    // first bit of a deflate block header
    uint64_t bit_stop;     // a block header AT this position ends the unit (the next unit starts there); ~0: none
    uint64_t bit_end;      // the input ends here (decoding beyond it: truncated)
    uint64_t tok_off;      // the unit's slot region in the token buffer: [tok_off, tok_off + tok_cap), tok_off % 4 == 0
    uint32_t tok_cap;      // % 4 == 0
    uint32_t wend;         // at least 80 dwords behind bit_end (a lane's ring holds the 256 bytes behind its position)
    // A unit whose decode passes bit_stop without a block starting there (the candidate was not a block start) goes on until a block
    // header lies on one of these positions -- the stream's other candidates, ascending -- or its input ends: no second launch.
    const uint64_t* cands;
    uint32_t n_cands, pad;
};
struct Result {
    uint32_t status;     // 0, or QD_INFLATE_*
    uint32_t final_seen; // the stream's last block (BFINAL) was decoded: the unit ends behind its end-of-block symbol
    uint32_t n_slots, text_len;          // tokens made and the text they stand for
    uint64_t bit_next;                   // where the unit stopped: == bit_stop, or behind the final block
    // the last block boundary the decode passed (a unit that ran out of input or of space ends here for the caller)
    uint64_t blk_bit;
    uint32_t blk_slots, blk_text;
};

template <class C>
struct Lane {
    const uint32_t* comp;
    // The stream reaches the lane through a ring of 16-byte chunks in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no register
    // destination, so nothing waits for a load where it is issued) and topped up for all lanes of the wave together every
    // ROUND_TURNS turns, behind ONE wait.  (The first form fetched the next word into a register at every refill: the compiler
    // copies a loaded loop-carried value at once, behind s_waitcnt vmcnt(0) -- which on gfx9 also waits for every store before
    // it -- and with 64 lanes refilling at their own times the wave took that wait on nearly every turn: 1 650 cycles a turn,
    // profiles/r05_inflate3_first_form.txt.)
    uint32_t rd;       // dwords of the stream moved into buf so far (the stream position is 32 rd - have bits)
    uint32_t have;     // valid bits in buf
    uint32_t fetched;  // chunks [.., fetched) of the stream have been requested (chunk = dword index / 4)
    uint32_t landed;   // chunks [.., landed) are in the ring
    uint32_t clast;    // the last chunk that may be loaded (requests beyond are clamped to it: garbage, never a fault)
    uint64_t buf;
    uint64_t bit_stop, bit_end;
    const uint64_t* cands;
    uint32_t n_cands;
    uint32_t* lens;  // LENS_DW dwords of scratch (global memory)
    // Codes longer than the first level, the canonical way: entry j is for length DB + 1 + j -- the left-aligned upper limit of that
    // length's codes (15 bits; 0: none) and the place of its first symbol among the long symbols minus its first code, 16 bits each,
    // the literal/length code's in the low half, the distance code's in the high half: ONE chain of compares serves a turn whatever
    // its lane decodes (two chains, one per code, ran on nearly every turn: some lane of 64 always holds the other kind).
    uint32_t plim[C::NP], pbas[C::NP];
    uint32_t on, on0, on_end;   // next slot, first slot, end of the region in the launch's token buffer (a wave-uniform base: turn()'s `tok`)
    uint32_t text_len, pend, stored_left;
    uint32_t state, status, final_seen;
    uint64_t blk_bit;
    uint32_t blk_slots, blk_text;
};

// ---- the stream ------------------------------------------------------------------------------------------------------------------
template <class C>
QD3_HD uint32_t ring_at(uint32_t w, uint32_t lane) {
    return (((w >> 2) & (uint32_t)(C::RING_CHUNKS - 1)) << 8) + (lane << 2) + (w & 3u);
}

// chunk `src` of the stream -> slot `slot` of the ring, for the lanes that `want` it; lands some time later (ring_wait)
QD3_HD void ring_dma(const uint32_t* comp, uint32_t src, uint32_t* ring, uint32_t slot, uint32_t lane, bool want) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (want) {
        const uint32_t* g = comp + 4ull * src;
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(ring + 256u * slot);  // wave-uniform; the hardware adds 16 x lane
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
#else
    if (want) memcpy(ring + 256u * slot + 4u * lane, comp + 4ull * src, 16);
#endif
}
// every chunk requested so far is in the ring (the wave's lanes call it together)
QD3_HD void ring_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// requests the chunks behind the lane's position whose slots are free: afterwards the ring holds (or awaits) chunks [rd / 4, rd / 4 + CHUNKS)
template <class C>
QD3_HD void topup(Lane<C>& L, uint32_t* ring, uint32_t lane, bool active) {
    const uint32_t rc = L.rd >> 2;
#pragma unroll 1
    for (uint32_t j = 0; j < (uint32_t)C::RING_CHUNKS; ++j) {  // (the slot is wave-uniform: one LDS-DMA instruction serves the lanes that need this slot)
        const uint32_t c = rc + ((j - rc) & (uint32_t)(C::RING_CHUNKS - 1));
        ring_dma(L.comp, c < L.clast ? c : L.clast, ring, j, lane, active && c >= L.fetched);
    }
    if (active) L.fetched = rc + (uint32_t)C::RING_CHUNKS;
}
// (after ring_wait)
template <class C>
QD3_HD void landed_all(Lane<C>& L) {
    L.landed = L.fetched;
}
template <class C>
QD3_HD void rd_init(Lane<C>& L, uint64_t bit) {  // position only: prime() once the ring holds the first chunks
    L.rd = (uint32_t)(bit >> 5);
    L.fetched = L.landed = L.rd >> 2;
    L.buf = 0;
    L.have = (uint32_t)bit & 31u;  // (kept here until prime(): the bits of the first dword in front of the position)
}
template <class C>
QD3_HD void prime(Lane<C>& L, const uint32_t* ring, uint32_t lane) {
    const uint32_t sh = L.have;
    const uint64_t lo = ring[ring_at<C>(L.rd, lane)], hi = ring[ring_at<C>(L.rd + 1, lane)];
    L.buf = (lo | (hi << 32)) >> sh;
    L.have = 64 - sh;
    L.rd += 2;
}
// A header's way of taking bits: at least 33 in hand afterwards.  The lanes that parse headers call it in step; one that has used
// up what landed makes the wave top up and wait (rare: the ring is full when a header starts, and few headers are longer).
template <class C>
QD3_HD void refill(Lane<C>& L, uint32_t* ring, uint32_t lane) {
    if (L.have <= 32) {
        if (L.rd >= 4u * L.landed) {
            topup(L, ring, lane, true);
            ring_wait();
            landed_all(L);
        }
        L.buf |= (uint64_t)ring[ring_at<C>(L.rd, lane)] << L.have;
        L.have += 32;
        ++L.rd;
    }
}
template <class C>
QD3_HD uint64_t bitpos(const Lane<C>& L) {
    return (uint64_t)L.rd * 32u - L.have;
}
template <class C>
QD3_HD void drop(Lane<C>& L, uint32_t n) {
    L.buf >>= n;
    L.have -= n;
}
template <class C>
QD3_HD void fail(Lane<C>& L, uint32_t code) {
    if (!L.status) L.status = code;
    L.state = ST_DONE;
}

// ---- tokens out ------------------------------------------------------------------------------------------------------------------
template <class C>
QD3_HD uint32_t slots_made(const Lane<C>& L) {
    return L.on - L.on0;
}

// ---- fifteen 16-bit fields in four registers (a lane cannot index an array by a variable without it going to scratch memory) -----
struct Pk {
    uint64_t a, b, c, d;
};
QD3_HD uint32_t pk_get(const Pk& p, uint32_t l) {
    const uint64_t w = l < 4 ? p.a : (l < 8 ? p.b : (l < 12 ? p.c : p.d));
    return (uint32_t)(w >> (16u * (l & 3u))) & 0xFFFFu;
}
QD3_HD void pk_add(Pk& p, uint32_t l, uint32_t v) {
    const uint64_t x = (uint64_t)v << (16u * (l & 3u));
    if (l < 4) p.a += x;
    else if (l < 8) p.b += x;
    else if (l < 12) p.c += x;
    else p.d += x;
}

// table entries
constexpr uint32_t E_NONE = 0x30u;  // kind 3, zero bits: a longer code, or no code
QD3_HD uint32_t lit_entry(uint32_t sym, uint32_t nbits) {
    if (sym < 256) return nbits | (sym << 6);
    if (sym == 256) return nbits | (2u << 4);
    if (sym >= 286) return E_NONE;  // (the fixed code's two unused symbols: refused when met, like a code that does not exist)
    return nbits | (1u << 4) | ((sym - 257u) << 6);
}
QD3_HD uint32_t dist_entry(uint32_t sym, uint32_t nbits) { return sym >= 30 ? E_NONE : nbits | (1u << 4) | (sym << 6); }

// One Huffman code from n code lengths (nibbles in lens[]): first-level table t[2^XB], the longer codes' symbols appended to
// lng[] from *long_used on, their limits / bases to lim[] / bas[].  0, or a QD_INFLATE_* code.
template <class C, bool DIST>
QD3_HD uint32_t build(const uint32_t* lens, uint32_t n, uint16_t* t, uint16_t* lng, uint32_t* plim, uint32_t* pbas, uint32_t* long_used) {
    constexpr int XB = DIST ? C::DB : C::LB;
    constexpr uint32_t N = 1u << XB;
    constexpr int NX = 15 - XB;
    constexpr int J0 = XB - C::DB;           // the packed entry of length XB + 1
    constexpr uint32_t HS = DIST ? 16u : 0u;  // this code's half of a packed entry
    Pk cnt{0, 0, 0, 0};
    for (uint32_t w = 0; 8u * w < n; ++w) {
        const uint32_t d = lens[w];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t l = (d >> (4u * j)) & 15u;
            if (8u * w + j < n && l) pk_add(cnt, l, 1);
        }
    }
    int left = 1;
    uint32_t total = 0, maxl = 0;
    Pk first{0, 0, 0, 0};
    {
        uint32_t code = 0, prev = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15; ++l) {
            const uint32_t c = pk_get(cnt, l);
            left = (left << 1) - (int)c;
            if (left < 0) return QD_INFLATE_BAD_TABLE;  // over-subscribed
            total += c;
            if (c) maxl = l;
            code = (code + prev) << 1;
            prev = c;
            pk_add(first, l, code);
        }
    }
    constexpr bool BYTES = DIST && C::DBYTE;  // (a byte entry: bits | symbol << 3; 0: a longer code, or none)
    uint8_t* const t8 = reinterpret_cast<uint8_t*>(t);
    if (BYTES) {
        for (uint32_t i = 0; i < N / 4; ++i) reinterpret_cast<uint32_t*>(t)[i] = 0;
    } else {
        for (uint32_t i = 0; i < N / 2; ++i) reinterpret_cast<uint32_t*>(t)[i] = E_NONE | (E_NONE << 16);
    }
#pragma unroll
    for (int j = 0; j < C::NP; ++j) {  // (this code's halves: the shorter lengths of the literal/length code stay 0 -- never "below the limit")
        plim[j] &= ~(0xFFFFu << HS);
        pbas[j] &= ~(0xFFFFu << HS);
    }
    if (total == 0) return 0;                                // no codes at all: every look-up fails (zlib: as long as none is used)
    if (left > 0 && maxl != 1) return QD_INFLATE_BAD_TABLE;  // incomplete: only a single one-bit code may be (zlib's rule)
    uint32_t loff = *long_used;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const uint32_t l = (uint32_t)(XB + 1 + k), c = pk_get(cnt, l), f = pk_get(first, l);
        plim[J0 + k] |= ((f + c) << (15u - l)) << HS;  // (<= 2^15)
        pbas[J0 + k] |= ((loff - f) & 0xFFFFu) << HS;
        loff += c;
    }
    if (loff > (uint32_t)C::NLONG) return QD_INFLATE_TABLE_SPACE;
    *long_used = loff;
    Pk next = first;
    for (uint32_t w = 0; 8u * w < n; ++w) {
        const uint32_t d = lens[w];
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t s = 8u * w + j, l = (d >> (4u * j)) & 15u;
            if (s >= n || !l) continue;
            const uint32_t c = pk_get(next, l);
            pk_add(next, l, 1);
            const uint32_t e = DIST ? dist_entry(s, l) : lit_entry(s, l);
            if (l <= (uint32_t)XB) {  // the stream carries a code MSB first inside an LSB-first bit order: indexed by the reversed code
                const uint32_t rev = brev32(c) >> (32u - l);
                if (BYTES) {
                    const uint32_t e8 = s >= 30 ? 0u : l | (s << 3);
                    for (uint32_t k = rev; k < N; k += 1u << l) t8[k] = (uint8_t)e8;
                } else {
                    for (uint32_t k = rev; k < N; k += 1u << l) t[k] = (uint16_t)e;
                }
            } else {
                uint32_t at = 0;
#pragma unroll
                for (int k = 0; k < NX; ++k)
                    if ((uint32_t)(XB + 1 + k) == l) at = (((pbas[J0 + k] >> HS) & 0xFFFFu) + c) & 0xFFFFu;
                if (at < (uint32_t)C::NLONG) lng[at] = (uint16_t)e;
            }
        }
    }
    return 0;
}

// order of the code-length code's lengths in a block header (RFC 1951), five bits each
constexpr uint64_t CLORDER_LO = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) |
                                (5ull << 45) | (11ull << 50) | (4ull << 55);
constexpr uint64_t CLORDER_HI = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);

// A block header at the lane's position: tables built, state LIT / STORED -- or DONE (the unit's stop position, an error), or
// HEADER again (an empty stored block).  tab: the lane's LDS.
template <class C>
QD3_HD void header(Lane<C>& L, uint16_t* tab, uint32_t* ring, uint32_t lane) {
    uint16_t* const lit = tab;
    uint16_t* const dst = tab + C::LIT_N;
    uint16_t* const lng = tab + C::LIT_N + C::DST_HW;
    const uint64_t at = bitpos(L);
    if (at <= L.bit_end) {  // (a block that "ended" behind the input's end was decoded from bytes that are not the stream's)
        L.blk_bit = at;
        L.blk_slots = slots_made(L);
        L.blk_text = L.text_len;
    }
    if (at == L.bit_stop) {  // the next unit starts here
        L.state = ST_DONE;
        return;
    }
    if (at > L.bit_stop) {  // the position this unit was to stop at is no block boundary: on to the next candidate that is one
        uint32_t lo = 0, hi = L.n_cands;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (L.cands[mid] < at) lo = mid + 1;
            else hi = mid;
        }
        if (lo < L.n_cands && L.cands[lo] == at) {
            L.state = ST_DONE;
            return;
        }
    }
    if (at + 3 > L.bit_end) return fail(L, QD_INFLATE_TRUNCATED);
    refill(L, ring, lane);
    L.final_seen = (uint32_t)L.buf & 1u;
    const uint32_t type = ((uint32_t)L.buf >> 1) & 3u;
    drop(L, 3);
    if (type == 0) {  // stored: byte aligned LEN, NLEN, then LEN bytes -- they leave as literals, a byte a turn
        drop(L, L.have & 7u);
        refill(L, ring, lane);
        const uint32_t len = (uint32_t)L.buf & 0xFFFFu, nlen = (uint32_t)(L.buf >> 16) & 0xFFFFu;
        drop(L, 32);
        if ((len ^ 0xFFFFu) != nlen) return fail(L, QD_INFLATE_BAD_STORED);
        if (bitpos(L) + 8ull * len > L.bit_end) return fail(L, QD_INFLATE_TRUNCATED);
        L.stored_left = len;
        L.state = len ? (uint32_t)ST_STORED : (L.final_seen ? (uint32_t)ST_DONE : (uint32_t)ST_HEADER);
        refill(L, ring, lane);  // (a turn starts with at least 33 bits in hand)
        return;
    }
    if (type == 3) return fail(L, QD_INFLATE_BAD_TYPE);
    uint32_t nlit = 288, ndist = 30;
    if (type == 1) {  // fixed codes: 8 bits for 0..143, 9 for 144..255, 7 for 256..279, 8 for 280..287; distances 5 bits
        for (uint32_t w = 0; w < 36; ++w) L.lens[w] = w < 18 ? 0x88888888u : (w < 32 ? 0x99999999u : (w < 35 ? 0x77777777u : 0x88888888u));
        for (uint32_t w = 36; w < 40; ++w) L.lens[w] = 0x55555555u;
        ndist = 32;  // (the fixed distance code is complete with its two unused symbols, which are refused when met)
    } else {
        refill(L, ring, lane);
        nlit = ((uint32_t)L.buf & 31u) + 257u;
        ndist = (((uint32_t)L.buf >> 5) & 31u) + 1u;
        const uint32_t ncl = (((uint32_t)L.buf >> 10) & 15u) + 4u;
        drop(L, 14);
        if (nlit > 286 || ndist > 30) return fail(L, QD_INFLATE_BAD_TABLE);
        uint64_t cl = 0;  // the code-length code's 19 lengths, three bits each, by symbol
        for (uint32_t k = 0; k < ncl; ++k) {
            refill(L, ring, lane);
            const uint32_t sym = (uint32_t)((k < 12 ? CLORDER_LO >> (5u * k) : CLORDER_HI >> (5u * (k - 12u))) & 31u);
            cl |= (uint64_t)((uint32_t)L.buf & 7u) << (3u * sym);
            drop(L, 3);
        }
        // its table: 128 bytes where the distance table (and, behind a byte-wide one, the long symbols) will be: symbol | bits << 5; the code must be complete
        uint8_t* const clt = reinterpret_cast<uint8_t*>(dst);
        uint64_t cc = 0;  // count per length, eight bits each
        for (uint32_t s = 0; s < 19; ++s) {
            const uint32_t l = (uint32_t)(cl >> (3u * s)) & 7u;
            if (l) cc += 1ull << (8u * l);
        }
        uint64_t nx = 0;
        {
            uint32_t code = 0, kraft = 0;
#pragma unroll
            for (uint32_t l = 1; l <= 7; ++l) {
                code = (code + (l > 1 ? (uint32_t)(cc >> (8u * (l - 1u))) & 255u : 0u)) << 1;
                nx |= (uint64_t)code << (8u * l);
                kraft += ((uint32_t)(cc >> (8u * l)) & 255u) << (7u - l);
            }
            if (kraft != 128u) return fail(L, QD_INFLATE_BAD_TABLE);
        }
        for (uint32_t s = 0; s < 19; ++s) {
            const uint32_t l = (uint32_t)(cl >> (3u * s)) & 7u;
            if (!l) continue;
            const uint32_t c = (uint32_t)(nx >> (8u * l)) & 255u;
            nx += 1ull << (8u * l);
            const uint32_t rev = brev32(c) >> (32u - l);
            for (uint32_t k = rev; k < 128u; k += 1u << l) clt[k] = (uint8_t)(s | (l << 5));
        }
        // the two codes' lengths, run-length coded; a nibble each into the scratch
        uint32_t idx = 0, prev = 0, cw = 0, accw = 0, eob_len = 0;
        const uint32_t total = nlit + ndist;
        auto put = [&](uint32_t v) {
            const uint32_t pos = idx < nlit ? idx : 288u + (idx - nlit), w = pos >> 3;
            if (w != cw) {
                L.lens[cw] = accw;
                cw = w;
                accw = 0;
            }
            accw |= v << (4u * (pos & 7u));
            if (pos == 256u) eob_len = v;
            ++idx;
        };
        while (idx < total) {
            refill(L, ring, lane);
            const uint32_t e = clt[(uint32_t)L.buf & 127u];
            drop(L, e >> 5);
            const uint32_t s = e & 31u;
            if (s < 16) {
                put(s);
                prev = s;
                continue;
            }
            uint32_t rep, v = 0;
            if (s == 16) {
                if (idx == 0) return fail(L, QD_INFLATE_BAD_TABLE);
                v = prev;
                rep = 3u + ((uint32_t)L.buf & 3u);
                drop(L, 2);
            } else if (s == 17) {
                rep = 3u + ((uint32_t)L.buf & 7u);
                drop(L, 3);
            } else {
                rep = 11u + ((uint32_t)L.buf & 127u);
                drop(L, 7);
            }
            if (idx + rep > total) return fail(L, QD_INFLATE_BAD_TABLE);
            for (uint32_t r = 0; r < rep; ++r) put(v);
            prev = v;
        }
        L.lens[cw] = accw;
        if (bitpos(L) > L.bit_end) return fail(L, QD_INFLATE_TRUNCATED);
        if (eob_len == 0) return fail(L, QD_INFLATE_BAD_TABLE);  // no end-of-block code
    }
    uint32_t long_used = 0;
    uint32_t rc = build<C, false>(L.lens, nlit, lit, lng, L.plim, L.pbas, &long_used);
    if (!rc) rc = build<C, true>(L.lens + 36, ndist, dst, lng, L.plim, L.pbas, &long_used);
    if (rc) return fail(L, rc);
    L.state = ST_LIT;
    refill(L, ring, lane);  // (a turn starts with at least 33 bits in hand)
}

// One turn of a decoding lane: a literal/length code or a distance code.  At least 33 bits are in hand when it starts; it takes at
// most 28 and moves the stream's next dword in behind them when fewer than 33 are left.  That dword is read from the ring at the
// start of the turn, needed or not: its trip to LDS runs beside the table look-up's.
// Straight-line code: what a lane holds decides by selects, not branches -- a wave's lanes hold everything at once, so every
// branch of a turn written with if / else was executed on every turn (510 instructions per turn, 1 800 cycles:
// profiles/r05_inflate3_first_form.txt).  The one branch left: a code longer than the first level.  Tokens leave as they are made
// (a 2-byte store per literal, two per match, at the lane's own place: nothing is gathered in registers first).
template <class C>
QD3_HD void turn(Lane<C>& L, const uint16_t* tab, const uint32_t* ring, uint32_t lane, uint16_t* tok) {
    const uint32_t next_word = ring[ring_at<C>(L.rd, lane)];
    const uint32_t lo32 = (uint32_t)L.buf;
    const bool dist = L.state == ST_DIST;
    uint32_t e;
    if (C::DBYTE) {  // the distance table's entries are bytes: the 16-bit word around one is read, and the entry put into the common form
        const uint32_t ix = lo32 & (dist ? (uint32_t)C::DST_N - 1u : (uint32_t)C::LIT_N - 1u);
        const uint32_t w = tab[dist ? (uint32_t)C::LIT_N + (ix >> 1) : ix];
        const uint32_t b = (w >> (8u * (ix & 1u))) & 255u;
        e = dist ? ((b & 7u) ? (b & 7u) | (1u << 4) | ((b >> 3) << 6) : E_NONE) : w;
    } else {
        e = tab[(dist ? (uint32_t)C::LIT_N : 0u) + (lo32 & (dist ? (uint32_t)C::DST_N - 1u : (uint32_t)C::LIT_N - 1u))];
    }
    if ((e & 0x3Fu) == E_NONE) {  // a code longer than the first level, or none: the next 15 bits MSB first against the lengths' limits
        const uint32_t x = brev32(lo32) >> 17;
        const uint32_t half = dist ? 16u : 0u;
        uint32_t at = 0xFFFFFFFFu;
#pragma unroll
        for (int j = C::NP - 1; j >= 0; --j) {  // (the limits rise with the length: the last one that holds is the shortest length)
            const uint32_t lim = (L.plim[j] >> half) & 0xFFFFu, bas = (L.pbas[j] >> half) & 0xFFFFu;
            if (x < lim) at = (bas + (x >> (14 - C::DB - j))) & 0xFFFFu;
        }
        e = at < (uint32_t)C::NLONG ? tab[C::LIT_N + C::DST_HW + at] : E_NONE;
    }
    const uint32_t nb = e & 15u, kind = (e >> 4) & 3u, val = e >> 6;
    const bool is_lit = kind == 0u, is_len = !dist && kind == 1u, is_eob = kind == 2u;  // (the distance table holds kind 1 only)
    // a length's or a distance's base and extra bits (RFC 1951's tables are regular: computed)
    const uint32_t eb_l = (val < 8u || val == 28u) ? 0u : ((val >> 2) - 1u) & 7u;  // (masked: a literal's value runs through here too)
    const uint32_t base_l = val < 8u ? 3u + val : (val == 28u ? 258u : 3u + ((4u + (val & 3u)) << eb_l));
    const uint32_t eb_d = val < 4u ? 0u : ((val >> 1) - 1u) & 15u;
    const uint32_t base_d = val < 4u ? 1u + val : 1u + ((2u + (val & 1u)) << eb_d);
    const uint32_t eb = dist ? eb_d : (is_len ? eb_l : 0u);
    const uint32_t value = (dist ? base_d : base_l) + ((lo32 >> nb) & ((1u << eb) - 1u));  // (code + extra bits <= 28: inside the low dword)
    drop(L, nb + eb);
    // the token: a literal is one slot; a match leaves when its distance is known, two slots
    const uint32_t ns = dist ? 2u : (is_lit ? 1u : 0u);
    const bool room = L.on + ns <= L.on_end;
    if (is_lit && room) tok[L.on] = (uint16_t)val;
    if (dist && room) {
        tok[L.on] = (uint16_t)(TOK_MATCH | (L.pend - 3u));
        tok[L.on + 1u] = (uint16_t)(value - 1u);
    }
    L.on += room ? ns : 0u;
    L.text_len += dist ? L.pend : (is_lit ? 1u : 0u);
    L.pend = is_len ? value : L.pend;
    uint32_t st = is_len ? (uint32_t)ST_DIST : (uint32_t)ST_LIT;
    st = is_eob ? (L.final_seen ? (uint32_t)ST_DONE : (uint32_t)ST_HEADER) : st;
    if (nb == 0u || !room) {
        if (!L.status) L.status = nb == 0u ? (uint32_t)QD_INFLATE_BAD_CODE : (uint32_t)QD_INFLATE_TOKEN_SPACE;
        st = ST_DONE;
    }
    L.state = st;
    const bool more = L.have <= 32;
    L.buf |= more ? (uint64_t)next_word << (L.have & 63u) : 0ull;
    L.have += more ? 32u : 0u;
    L.rd += more ? 1u : 0u;
}
// A turn of a lane inside a stored block: a byte of it leaves as a literal (rare in fastq: a loop of its own, so that the turns
// above do not carry it).
template <class C>
QD3_HD void turn_stored(Lane<C>& L, const uint32_t* ring, uint32_t lane, uint16_t* tok) {
    const uint32_t next_word = ring[ring_at<C>(L.rd, lane)];
    if (L.on < L.on_end) {
        tok[L.on++] = (uint16_t)((uint32_t)L.buf & 255u);
    } else {
        if (!L.status) L.status = QD_INFLATE_TOKEN_SPACE;
        L.state = ST_DONE;
    }
    drop(L, 8);
    ++L.text_len;
    if (--L.stored_left == 0 && L.state != ST_DONE) L.state = L.final_seen ? (uint32_t)ST_DONE : (uint32_t)ST_HEADER;
    if (L.have <= 32) {
        L.buf |= (uint64_t)next_word << L.have;
        L.have += 32;
        ++L.rd;
    }
}

template <class C>
QD3_HD void lane_init(Lane<C>& L, const Unit& u, uint32_t* lens) {
    L.comp = u.base;
    L.clast = u.wend >= 4 ? (u.wend - 4) >> 2 : 0;
    L.bit_stop = u.bit_stop;
    L.bit_end = u.bit_end;
    L.cands = u.cands;
    L.n_cands = u.n_cands;
    L.lens = lens;
    L.on0 = L.on = (uint32_t)u.tok_off;  // (slot indices are 32 bit: a launch's token buffer holds less than 2^32 slots)
    L.on_end = L.on0 + u.tok_cap;
    L.text_len = L.pend = L.stored_left = 0;
    L.state = ST_HEADER;
    L.status = L.final_seen = 0;
    L.blk_bit = u.bit_start;
    L.blk_slots = L.blk_text = 0;
#pragma unroll
    for (int j = 0; j < C::NP; ++j) L.plim[j] = L.pbas[j] = 0;
    rd_init(L, u.bit_start);
    if (u.bit_start >= u.bit_end) {  // nothing to decode
        L.state = ST_DONE;
        L.status = QD_INFLATE_TRUNCATED;
    }
}

// the lane is done: its last slots leave, its result is written
template <class C>
QD3_HD void lane_finish(Lane<C>& L, Result* r) {
    const uint32_t n = slots_made(L);
    if (!L.status && L.state == ST_DONE && bitpos(L) > L.bit_end) L.status = QD_INFLATE_TRUNCATED;
    r->status = L.status;
    r->final_seen = (L.final_seen && !L.status) ? 1u : 0u;
    r->n_slots = n;
    r->text_len = L.text_len;
    r->bit_next = bitpos(L);
    r->blk_bit = L.blk_bit;
    r->blk_slots = L.blk_slots;
    r->blk_text = L.blk_text;
}

}  // namespace qd3
