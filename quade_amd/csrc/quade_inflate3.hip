// Third inflater (r05): every DEFLATE symbol decoded ONCE, by one lane, 64 streams per wave -- then the tokens resolved by workgroups.
//
// What it replaces: the gunzip inside pyFastq.FastqReader that the reference's chunk loops draw their records from
// (src/Quade.py:203-206, 234-236) -- for BGZF blocks as quade_inflate.hip's forms do, and (the gzip kernels further down) for the
// reference's real input format: ordinary single-member .fastq.gz files (its test/dataset/*.fastq.gz).
//
//   inflate3_tokens<Cfg>      one LANE per unit (a BGZF block; a stretch of a gzip member from one deflate block to another): header
//                             parse, table build and the symbol decode are the lane's own serial work out of its slice of LDS
//                             (inflate3_lane.h); a wave advances 64 units in lock step.  Out: 16-bit token slots per unit.
//   inflate3_resolve_bgzf     one workgroup per BGZF block: token lengths -> positions (scan), literals into the block's text in LDS,
//                             matches as parents resolved by pointer jumping window by window (the second form's match stage, fed from
//                             the token list instead of a second decode), CRC-32 against the trailer, text out.
// Both are byte / integer work: no MFMA.  The token kernel is latency-bound by design (one dependent LDS look-up per code and lane);
// its throughput is the number of units in flight, which a launch of thousands of blocks supplies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "crc_lds.h"
#include "quade_inflate.h"
#include "inflate3_lane.h"
#include "quade_inflate3.h"
#include "quade_pool.h"

namespace {

// Two sizes of the lane decoder.  A wave's LDS -- 64 lanes' tables and its input ring -- decides how many waves a CU holds, and a CU
// has four SIMDs: a token launch is a latency-bound kernel whose throughput is the number of waves in flight.
//   CfgS  (what a launch runs)  7-bit literal/length table, 6-bit distance table of BYTE entries, 88 symbols with longer codes,
//         a ring of 8 chunks: 500 B + 128 B a lane, 39.25 KB a wave -> FOUR waves per CU, one per SIMD.  fastq's deflate blocks
//         have ~45 literal/length symbols behind 7 bits and <= 18 distance symbols behind 6 (tools/deflate_stats.py).
//   CfgA  (the redo)  8 / 7 bits, 112 long symbols -- what the fixed code has behind 8 bits -- and a ring of 16 chunks: 78 KB a wave,
//         two per CU.  A unit that CfgS gave up with "table space" (a fixed block; a block of byte soup) is decoded again by a
//         second launch of this size, in which every other lane retires at once.
using CfgS = qd3::Cfg<7, 6, 88, true, 8, 12>;
using CfgA = qd3::Cfg<8, 7, 112>;
static_assert((CfgS::LANE_DW * 64 + CfgS::RING_DW) * 4 * 4 <= 160 * 1024 - 2048, "four workgroups per CU");
static_assert((CfgA::LANE_DW * 64 + CfgA::RING_DW) * 4 * 2 <= 160 * 1024, "two workgroups per CU");

// units[] (stretches of a gzip member) or jobs[] (BGZF: unit i = block i, its slot region i * QD_INFLATE3_TOK_STRIDE)
// redo: only the units whose result says "table space" (the launch before this one, of a smaller configuration, gave them up) --
// by a small grid whose waves walk all groups of 64 units (a launch as wide as the first would spend a millisecond placing
// workgroups of 78 KB that find nothing to do).
// WPB waves per workgroup: independent of each other (nothing here is a workgroup barrier), each with its own slice of the
// workgroup's LDS -- a workgroup of four places four waves on ONE CU (one per SIMD) instead of wherever the dispatcher finds
// room, which leaves whole CUs to the kernels of the other streams (the coder's workgroups need ~90 KB: a CU that holds two
// token waves has no room for one).
template <class C, int WPB>
__global__ __launch_bounds__(64 * WPB) void inflate3_tokens(const qd3::Unit* units, const qd_inflate3_job* jobs, uint32_t n_units, uint16_t* tokens, uint32_t* lens_scratch,
                                                            qd3::Result* res, uint32_t wait_rounds, uint32_t redo) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: the ring's LDS-DMA base goes through M0)
    uint32_t* const lds3 = lds_all + wave * (uint32_t)(C::LANE_DW * 64 + C::RING_DW);
    uint32_t* const ring = lds3;  // the wave's input ring first (16-byte aligned slots), the lanes' tables behind it
    uint16_t* const tab = reinterpret_cast<uint16_t*>(lds3 + C::RING_DW + lane * (uint32_t)C::LANE_DW);
#pragma unroll 1
  for (uint32_t group = blockIdx.x * (uint32_t)WPB + wave; 64u * group < n_units; group += gridDim.x * (uint32_t)WPB) {
    const uint32_t u = 64u * group + lane;
    bool mine = u < n_units;
    if (redo) {
        mine = mine && res[u].status == (uint32_t)QD_INFLATE_TABLE_SPACE;
        if (!__ballot(mine)) continue;
    }
    qd3::Lane<C> L;
    {
        qd3::Unit un{nullptr, 0, ~0ull, 0, 0, 0, 0, nullptr, 0, 0};
        if (mine) {
            if (jobs) {
                const qd_inflate3_job j = jobs[u];
                const uintptr_t a = reinterpret_cast<uintptr_t>(j.payload);
                un.base = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)15);
                un.bit_start = 8ull * (a & 15u);
                un.bit_end = un.bit_start + 8ull * j.in_len;
                un.tok_off = (uint64_t)u * QD_INFLATE3_TOK_STRIDE;
                un.tok_cap = QD_INFLATE3_TOK_STRIDE;
                un.wend = (uint32_t)(((a & 15u) + j.in_len + 3u) / 4u) + 80u;
            } else {
                un = units[u];
            }
        }
        qd3::lane_init(L, un, lens_scratch + (size_t)(mine ? u : 0) * qd3::LENS_DW);
        if (!mine) {
            L.state = qd3::ST_DONE;
            L.status = 0;
        }
    }
    qd3::topup(L, ring, lane, mine);
    qd3::ring_wait();
    qd3::landed_all(L);
    qd3::prime(L, ring, lane);
    // Lanes at a block header wait until the wave does headers together (a header is ~300 turns' worth of one lane's serial work,
    // paid by the whole wave whoever takes part): at once when nobody decodes, else after wait_rounds more rounds of turns.
    uint32_t waited = 0;
#pragma unroll 1
    for (;;) {
        const bool hdr = L.state == qd3::ST_HEADER, dec = L.state <= qd3::ST_STORED;
        const uint64_t mh = __ballot(hdr), md = __ballot(dec);
        if (!mh && !md) break;
        if (mh && (!md || waited >= wait_rounds)) {
            qd3::topup(L, ring, lane, hdr);  // a header starts with its lane's ring full
            qd3::ring_wait();
            qd3::landed_all(L);
            if (hdr) qd3::header<C>(L, tab, ring, lane);
            qd3::topup(L, ring, lane, L.state <= qd3::ST_STORED);  // ... and the turns behind it too
            qd3::ring_wait();
            qd3::landed_all(L);
            waited = 0;
            continue;
        }
        // (a lane that has run past its input's end -- the last stretch of a gzip step, inside a block that needs bytes not there yet,
        //  or a damaged block -- would decode whatever lies behind until its slots are full: it stops here; what it says counts up
        //  to the last block boundary inside the input)
        if (dec && qd3::bitpos(L) > L.bit_end + 64u) qd3::fail(L, QD_INFLATE_TRUNCATED);
        qd3::ring_wait();  // what the last round requested is there ...
        qd3::landed_all(L);
        qd3::topup(L, ring, lane, dec);  // ... what it used up is requested again, and lands while this round runs
        if (__ballot(L.state == qd3::ST_STORED)) {  // (stored blocks: byte soup, not fastq)
#pragma unroll 1
            for (int t = 0; t < C::ROUND_TURNS; ++t) {
                if (L.state == qd3::ST_STORED) qd3::turn_stored<C>(L, ring, lane, tokens);
                else if (L.state <= qd3::ST_DIST) qd3::turn<C>(L, tab, ring, lane, tokens);
            }
        } else {
#pragma unroll 1
            for (int t = 0; t < C::ROUND_TURNS; ++t)
                if (L.state <= qd3::ST_DIST) qd3::turn<C>(L, tab, ring, lane, tokens);
        }
        if (mh) ++waited;
    }
    if (mine) qd3::lane_finish(L, res + u);
  }
}

// ---- BGZF: tokens -> text, one workgroup per block ------------------------------------------------------------------------------------
template <int NT, int Q>
struct R3Lds {
    uint32_t ow[16384 + 4];  // the block's text (<= 64 KiB)
    uint16_t par[Q];         // the window's parents (absolute positions; par[p - qb] == p: final) -- later the CRC stage's tables
    uint32_t wsum[NT / 64 + 1];
    uint32_t ctl[8];
};

template <int NT>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += y;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t s = wsum[w];
        if ((uint32_t)w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

template <int NT, int Q>
__global__ __launch_bounds__(NT) void inflate3_resolve_bgzf(const qd_inflate3_job* jobs, uint32_t n_blocks, const uint16_t* tokens, const qd3::Result* res,
                                                            int32_t* status) {
    static_assert(Q % NT == 0 && Q / NT <= 32 && 65536 % Q == 0, "a lane's share of a window");
    static_assert((size_t)Q * 2 >= 4096 + (NT / 64) * 4, "the CRC stage's tables lie where the parents were");
    constexpr int K = Q / NT;
    constexpr int CRC_SW = NT >= 1024 ? 17 : (NT >= 512 ? 33 : 65);
    static_assert((size_t)NT * CRC_SW * 4 >= 65536, "the CRC stage's slices");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw3[];
    R3Lds<NT, Q>& S = *reinterpret_cast<R3Lds<NT, Q>*>(lds_raw3);
    uint8_t* const ob = reinterpret_cast<uint8_t*>(S.ow);
    // (volatile: a lane reads a parent's state, then -- only if that is final -- its byte; LDS address space spelled out, or volatile
    //  accesses through generic pointers become flat_load / flat_store)
    typedef volatile __attribute__((address_space(3))) uint8_t lds_vu8;
    typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;
    lds_vu8* const vob = (lds_vu8*)ob;
    lds_vu16* const par = (lds_vu16*)S.par;
    const uint32_t i = blockIdx.x, tid = threadIdx.x;
    if (i >= n_blocks) return;
    const qd_inflate3_job blk = jobs[i];
    const qd3::Result r = res[i];
    const uint32_t olen = blk.out_len;
    int err = (int)r.status;
    if (!err && olen > 65536u) err = QD_INFLATE_OVERRUN;
    if (!err && !r.final_seen) err = QD_INFLATE_BAD_TYPE;
    if (!err && r.text_len != olen) err = r.text_len > olen ? QD_INFLATE_OVERRUN : QD_INFLATE_LENGTH;
    if (!err && r.bit_next > 8ull * ((reinterpret_cast<uintptr_t>(blk.payload) & 15u) + (uint64_t)blk.in_len)) err = QD_INFLATE_TRUNCATED;
    if (tid < 8) S.ctl[tid] = 0;
    __syncthreads();
    if (!err) {
        const uint16_t* tk = tokens + (size_t)i * QD_INFLATE3_TOK_STRIDE;
        const uint32_t n_slots = r.n_slots;
        // positions of a window: par[p - qb] = p (final) until a match says otherwise
        auto init_par = [&](uint32_t w) {
            const uint32_t qb = w * Q;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t p = qb + tid + (uint32_t)k * NT;
                par[p - qb] = (uint16_t)p;
            }
            __syncthreads();
        };
        // Every byte a match produces has a parent, the byte `distance` before it; chains are shortened by pointer jumping: a byte
        // whose parent is final takes its value and becomes final, another adopts its parent's parent (quade_inflate.hip's match
        // stage).  A lane keeps a mask of its unsettled positions and sweeps them in rising order.
        auto finalize = [&](uint32_t w) {
            const uint32_t qb = w * Q, hi = min(qb + (uint32_t)Q, olen);
            if (tid == 0) S.ctl[5] = S.ctl[6] = S.ctl[7] = 0;
            __syncthreads();
            uint32_t pend = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t p = qb + tid + (uint32_t)k * NT;
                if (p < hi && par[p - qb] != (uint16_t)p) pend |= 1u << k;
            }
            constexpr int MATCH_ROUNDS = 24;
#pragma unroll 1
            for (int round = 0; round < MATCH_ROUNDS; ++round) {
#pragma unroll 1
                for (uint32_t left = pend; left; left &= left - 1u) {
                    const uint32_t k = (uint32_t)__builtin_ctz(left), p = qb + tid + k * NT;
                    const uint32_t q = par[p - qb];
                    if (q < qb || par[q - qb] == q) {
                        vob[p] = vob[q];
                        par[p - qb] = (uint16_t)p;
                        pend &= ~(1u << k);
                    } else {
                        par[p - qb] = par[q - qb];
                    }
                }
                uint32_t* flag = &S.ctl[5];
                if (pend) flag[round % 3] = 1;
                __syncthreads();
                const uint32_t more = flag[round % 3];
                if (tid == 0) flag[(round + 2) % 3] = 0;
                if (!more) break;
                if (round == MATCH_ROUNDS - 1 && tid == 0) S.ctl[4] = 1;  // (chains halve every round: cannot happen; such a block is not shipped)
            }
            __syncthreads();
        };
        uint32_t base_pos = 0, cur_win = 0;
        init_par(0);
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < n_slots; c0 += NT * 4u) {
            const uint32_t s0 = c0 + 4u * tid;
            uint64_t four = 0;
            uint32_t prev = 0, next = 0;
            if (s0 < n_slots) {
                four = *reinterpret_cast<const uint64_t*>(tk + s0);
                if (s0) prev = tk[s0 - 1];
                if (s0 + 4u < n_slots) next = tk[s0 + 4u];
            }
            uint32_t sl[5], ln[4], at[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[j] = (uint32_t)(four >> (16 * j)) & 0xFFFFu;
            sl[4] = next;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool valid = s0 + (uint32_t)j < n_slots;
                const bool is_dist = ((j ? sl[j - 1] : prev) & qd3::TOK_MATCH) != 0;
                ln[j] = (!valid || is_dist) ? 0u : ((sl[j] & qd3::TOK_MATCH) ? (sl[j] & 0xFFu) + 3u : 1u);
                sum += ln[j];
            }
            uint32_t total;
            uint32_t pos = base_pos + block_scan_excl<NT>(sum, S.wsum, total);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                at[j] = pos;
                if (ln[j] == 1u && pos < olen) ob[pos] = (uint8_t)sl[j];  // (a match is at least 3 bytes: length 1 is a literal)
                pos += ln[j];
            }
            // the windows this chunk's text touches, in order: a window is closed (its chains resolved) when the text moves past it
            const uint32_t chunk_hi = min(base_pos + total, olen);
            const uint32_t w_first = cur_win, w_last = max(w_first, chunk_hi ? (chunk_hi - 1u) / (uint32_t)Q : 0u);
#pragma unroll 1
            for (uint32_t w = w_first; w <= w_last; ++w) {
                if (w != cur_win) {
                    finalize(cur_win);
                    cur_win = w;
                    init_par(w);
                }
                const uint32_t wlo = w * Q, whi = min(wlo + (uint32_t)Q, olen);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ln[j] < 3u) continue;
                    const uint32_t d0 = at[j], len = ln[j], dist = sl[j + 1] + 1u;
                    if (dist > d0 || d0 + len > olen) {  // a BGZF block never reaches behind its own start
                        S.ctl[4] = 1;
                        continue;
                    }
                    const uint32_t a = max(d0, wlo), z = min(d0 + len, whi);
                    for (uint32_t p = a; p < z; ++p) par[p - wlo] = (uint16_t)(p - dist);
                }
            }
            base_pos += total;
        }
        finalize(cur_win);
        if (S.ctl[4]) err = QD_INFLATE_BAD_DISTANCE;
    }
    __syncthreads();
    if (blk.check_crc) {  // the block's CRC-32 against its trailer while the text is still in LDS
        uint32_t* crc_t = reinterpret_cast<uint32_t*>(S.par);
        qdcrc::stage_tables<NT>(crc_t);
        __syncthreads();
        const uint32_t crc = qdcrc::crc32_lds<NT, CRC_SW>(S.ow, err ? 0u : olen, crc_t, crc_t + 1024);
        if (!err && crc != blk.expect_crc) err = QD_INFLATE_CRC;
    }
    if (!err) {  // the text leaves: bytes up to the first 4-byte boundary of the destination, dwords, the tail
        uint8_t* const o = blk.out;
        const uint32_t head = min(olen, (uint32_t)((4u - ((uintptr_t)o & 3u)) & 3u));
        if (tid < head) o[tid] = ob[tid];
        const uint32_t nd = (olen - head) >> 2;
        for (uint32_t k = tid; k < nd; k += NT) {
            const uint32_t a = head + 4 * k, wi = a >> 2;
            *reinterpret_cast<uint32_t*>(o + a) = __builtin_amdgcn_alignbyte(S.ow[wi + 1], S.ow[wi], a & 3u);
        }
        for (uint32_t k = head + 4 * nd + tid; k < olen; k += NT) o[k] = ob[k];
    }
    if (tid == 0) status[i] = err;
}

// ======================================================================================================================================
// Ordinary gzip members (the reference's input format: src/Quade.py:203-206 opens plain .fastq.gz; its test/dataset files are single
// members): one DEFLATE stream of any length.  The published two-pass scheme (pugz / rapidgzip; quade_pgz.cpp is the host's form of
// it) on the device:
//   gz_probe          the stream's bytes are cut into stretches; a wave per stretch tries every bit offset for a block header that holds
//                     (type 2, counts in range, the code-length code complete, both codes complete, an end-of-block code);
//   inflate3_tokens   a lane per stretch decodes from its block start to the next stretch's: tokens, as for BGZF blocks;
//   (host)            a stretch counts only if it began exactly where its predecessor stopped: every accepted boundary is proven by
//                     the chain from the stream's true start, never guessed; text offsets by prefix sums of the stretches' lengths;
//   gz_resolve        a workgroup per unit (~1 MB of text: a run of stretches) turns tokens into 16-bit symbols -- a byte, or a marker
//                     "byte i of the 32 KiB before this unit" -- through a ring of the last 32 KiB + one window in LDS;
//   gz_windows        one workgroup per stream walks its units in order: a unit's last 32 KiB resolved against its predecessor's (the
//                     only serial step: 64 KB read, 32 KB written per unit);
//   gz_fixup          symbols -> bytes with every unit's resolved window.
// The member's CRC-32 and ISIZE are checked from per-unit CRCs (qd_text_crc32 + combine).
struct GzStretch {
    const uint32_t* base;  // 16-byte aligned; bit positions count from here
    uint64_t bit_from;     // first position to try
    uint64_t bit_to;       // positions tried are < bit_to
    uint64_t bit_end;      // the input ends here
};

__device__ __forceinline__ uint64_t gz_bits(const uint32_t* base, uint64_t bit) {  // 57+ valid bits from `bit` on
    const uint64_t w = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint64_t lo = base[w], mid = base[w + 1], hi = base[w + 2];
    return sh ? ((lo | (mid << 32)) >> sh) | (hi << (64u - sh)) : (lo | (mid << 32));
}

// does a dynamic block's header hold at `bit`?  clt: 128 bytes of the lane's own LDS
// text_only: ... and no byte that text does not hold (anything but tab, newline, carriage return and 32 .. 126) has a code.  A header
// that "holds" at a position where no block starts -- one in ~3 of what the first form reported from fastq at 16 KiB stretches -- has
// code lengths drawn from noise, spread over all 256 literals; a real block of fastq has none for ~160 of them.  (pugz asks the
// same of the decoded text; here the lengths say it before anything is decoded.)  Streams that are not text: qd_gz::decode's
// second try, without this.
__device__ __forceinline__ bool gz_text_lengths_only(uint32_t a, uint32_t b) {  // no non-text byte among the literals [a, b)
    b = min(b, 256u);
    if (a >= b) return true;
    const uint64_t non_text[4] = {0x00000000FFFFD9FFull, 0x8000000000000000ull, ~0ull, ~0ull};
    bool hit = false;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t lo = max(a, 64u * w), hi = min(b, 64u * w + 64u);
        if (lo < hi) {
            const uint64_t bits = (hi - lo == 64u ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << (lo - 64u * w);
            hit = hit || (non_text[w] & bits) != 0;
        }
    }
    return !hit;
}
__device__ bool gz_header_holds(const uint32_t* base, uint64_t bit, uint64_t bit_end, uint8_t* clt, bool text_only) {
    if (bit + 17 + 12 > bit_end) return false;
    uint64_t x = gz_bits(base, bit);
    if ((x & 7u) != 4u) return false;  // not the last block, dynamic codes
    const uint32_t nlit = ((uint32_t)(x >> 3) & 31u) + 257u, ndist = ((uint32_t)(x >> 8) & 31u) + 1u, ncl = ((uint32_t)(x >> 13) & 15u) + 4u;
    if (nlit > 286u || ndist > 30u) return false;
    uint64_t pos = bit + 17;
    x = gz_bits(base, pos);
    uint64_t cl = 0, cc = 0;
    for (uint32_t k = 0; k < ncl; ++k) {
        const uint32_t sym = (uint32_t)((k < 12 ? qd3::CLORDER_LO >> (5u * k) : qd3::CLORDER_HI >> (5u * (k - 12u))) & 31u);
        const uint32_t l = (uint32_t)(x >> (3u * k)) & 7u;
        cl |= (uint64_t)l << (3u * sym);
        if (l) cc += 1ull << (8u * l);
    }
    pos += 3u * ncl;
    uint64_t nx = 0;
    {
        uint32_t code = 0, kraft = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7; ++l) {
            code = (code + (l > 1 ? (uint32_t)(cc >> (8u * (l - 1u))) & 255u : 0u)) << 1;
            nx |= (uint64_t)code << (8u * l);
            kraft += ((uint32_t)(cc >> (8u * l)) & 255u) << (7u - l);
        }
        if (kraft != 128u) return false;  // (zlib wants this code complete)
    }
    for (uint32_t s = 0; s < 19; ++s) {
        const uint32_t l = (uint32_t)(cl >> (3u * s)) & 7u;
        if (!l) continue;
        const uint32_t c = (uint32_t)(nx >> (8u * l)) & 255u;
        nx += 1ull << (8u * l);
        const uint32_t rev = __brev(c) >> (32u - l);
        for (uint32_t k = rev; k < 128u; k += 1u << l) clt[k] = (uint8_t)(s | (l << 5));
    }
    // the two codes' lengths: only their Kraft sums, the end-of-block code's length and the longest length are kept
    const uint32_t total = nlit + ndist;
    uint32_t idx = 0, prev = 0, sum_l = 0, sum_d = 0, max_l = 0, max_d = 0, n_l = 0, n_d = 0, eob = 0;
    while (idx < total) {
        if (pos + 14 > bit_end) return false;
        x = gz_bits(base, pos);
        const uint32_t e = clt[(uint32_t)x & 127u], nb = e >> 5, s = e & 31u;
        x >>= nb;
        pos += nb;
        uint32_t rep = 1, v = s;
        if (s == 16) {
            if (idx == 0) return false;
            v = prev;
            rep = 3u + ((uint32_t)x & 3u);
            pos += 2;
        } else if (s == 17) {
            v = 0;
            rep = 3u + ((uint32_t)x & 7u);
            pos += 3;
        } else if (s == 18) {
            v = 0;
            rep = 11u + ((uint32_t)x & 127u);
            pos += 7;
        }
        if (idx + rep > total) return false;
        if (v) {
            const uint32_t in_l = idx < nlit ? min(rep, nlit - idx) : 0u, in_d = rep - in_l;
            sum_l += in_l << (15u - v);
            sum_d += in_d << (15u - v);
            n_l += in_l;
            n_d += in_d;
            if (in_l) max_l = max(max_l, v);
            if (in_d) max_d = max(max_d, v);
            if (idx <= 256u && idx + rep > 256u) eob = v;
            if (sum_l > 32768u || sum_d > 32768u) return false;  // over-subscribed
            if (text_only && in_l && !gz_text_lengths_only(idx, idx + in_l)) return false;
        }
        idx += rep;
        prev = v;
    }
    if (!eob) return false;
    if (sum_l != 32768u && !(n_l == 1u && max_l == 1u)) return false;
    if (sum_d != 32768u && n_d != 0u && !(n_d == 1u && max_d == 1u)) return false;
    return true;
}

// The three tests of gz_header_holds as stages of their own, so that a wave runs the dear ones on full sets of survivors:
//   1. three header bits + the two counts in range: every offset (one in nine passes);
//   2. the code-length code complete (Kraft sum 128): the survivors, 64 at a time (one in ~200 passes);
//   3. the run-length coded lengths decode and both codes are complete: their survivors, 16 or more at a time.
// (With all three inside one function a wave paid the second on every iteration and the third on one in 25 -- some lane always
//  needs them: 18 ms per launch, as long as the token kernel.  profiles/r05_gz_first_form.txt)
__device__ __forceinline__ bool gz_stage2(const uint32_t* base, uint64_t bit) {
    const uint32_t ncl = ((uint32_t)(gz_bits(base, bit) >> 13) & 15u) + 4u;
    const uint64_t x = gz_bits(base, bit + 17);
    uint32_t kraft = 0;
#pragma unroll
    for (uint32_t k = 0; k < 19; ++k) {
        const uint32_t l = (uint32_t)(x >> (3u * k)) & 7u;
        kraft += (k < ncl && l) ? 128u >> l : 0u;
    }
    return kraft == 128u;
}

// found[i] = the first position in [bit_from, bit_to) of stretch i at which a dynamic block's header holds, or ~0.  One wave per stretch.
// Stage 1 takes 2 048 positions a step: a lane holds 64 bits of the stream and tests its 32 positions at once with shifts and ands of
// the whole word (bits b .. b+2 = 0 0 1; the upper four bits of both counts not all set: a count of 30 or 31) -- ~60 instructions for
// what the first form spent 32 iterations of ~40 on.  Its survivors (one position in nine) are written to a queue in ascending
// order and go through stage 2 sixty-four at a time; what survives that waits for stage 3 as before.
constexpr uint32_t GZ_Q2_RUN = 16;  // survivors of stage 2 that wait for a stage-3 pass at most (a true block start waits with them)
constexpr uint32_t GZ_Q1_CAP = 704;  // (bits 0 0 1 cannot start at two of three neighbouring positions: at most 683 of 2 048 pass)
constexpr uint32_t GZ_PROBE_LDS = 64 * 128 + (GZ_Q1_CAP + 128) * 4;
__global__ __launch_bounds__(64) void gz_probe(const GzStretch* stretches, uint32_t n, uint64_t* found, uint32_t text_only) {
    extern __shared__ __attribute__((aligned(16))) uint8_t probe_lds[];
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const GzStretch st = stretches[i];
    uint8_t* clt = probe_lds + 128u * lane;
    typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
    lds_vu32* q1 = (lds_vu32*)(probe_lds + 64 * 128);  // positions (bits from the step's first) that passed stage 1
    lds_vu32* q2 = q1 + GZ_Q1_CAP;                      // offsets from bit_from that passed stage 2: up to 16 + 64 wait
    uint32_t n2 = 0;                                    // (wave-uniform)
    uint64_t hit = ~0ull;
    auto stage3 = [&]() {  // every queued survivor of stage 2 (ascending): the first that holds, if any
        for (uint32_t b = 0; b < n2 && hit == ~0ull; b += 64) {
            const bool have = b + lane < n2;
            const uint64_t bit = st.bit_from + (have ? q2[b + lane] : 0u);
            const bool ok = have && gz_header_holds(st.base, bit, st.bit_end, clt, text_only != 0);
            const uint64_t m = __ballot(ok);
            if (m) hit = st.bit_from + q2[b + (uint32_t)__builtin_ctzll(m)];
        }
        n2 = 0;
    };
    // a position needs 29 bits of input behind it (the three header bits, the counts, four lengths of the code-length code), and lies in [bit_from, bit_to)
    const uint64_t p_end = min(st.bit_to, st.bit_end >= 29 ? st.bit_end - 28 : 0ull);
#pragma unroll 1
    for (uint64_t p0 = st.bit_from & ~31ull; p0 < p_end && hit == ~0ull; p0 += 2048) {
        const uint64_t mine = p0 + 32u * lane;  // this lane's positions: mine + [0, 32)
        uint32_t mask = 0;
        if (mine < p_end && mine + 32 > st.bit_from) {
            const uint64_t w = mine >> 5;
            const uint64_t two = (uint64_t)st.base[w] | ((uint64_t)st.base[w + 1] << 32);
            const uint64_t a = ~two & ~(two >> 1) & (two >> 2);                                  // not the last block, dynamic codes
            const uint64_t hl = (two >> 4) & (two >> 5) & (two >> 6) & (two >> 7);               // 30 or 31 literal/length codes + 257
            const uint64_t hd = (two >> 9) & (two >> 10) & (two >> 11) & (two >> 12);            // 30 or 31 distance codes + 1
            mask = (uint32_t)(a & ~hl & ~hd);
            if (mine < st.bit_from) mask &= ~0u << (uint32_t)(st.bit_from - mine);
            if (mine + 32 > p_end) mask &= ~0u >> (uint32_t)(mine + 32 - p_end);
        }
        // ascending into q1: this lane's survivors behind those of the lanes below it
        uint32_t cnt = (uint32_t)__popc(mask), incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += y;
        }
        const uint32_t n1 = __shfl(incl, 63, 64);
        {
            uint32_t at = incl - cnt;
            for (uint32_t m = mask; m; m &= m - 1u) q1[at++] = 32u * lane + (uint32_t)__builtin_ctz(m);
        }
        // stage 2, sixty-four at a time
        for (uint32_t b = 0; b < n1; b += 64) {
            const bool have = b + lane < n1;
            const uint64_t pos = p0 + (have ? q1[b + lane] : 0u);
            const bool ok = have && gz_stage2(st.base, pos);
            const uint64_t m = __ballot(ok);
            if (ok) q2[n2 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(pos - st.bit_from);
            n2 += (uint32_t)__popcll(m);
            if (n2 >= GZ_Q2_RUN) {
                stage3();
                if (hit != ~0ull) break;
            }
        }
    }
    if (hit == ~0ull) stage3();
    if (lane == 0) found[i] = hit;
}

// ---- tokens -> 16-bit symbols, a workgroup per unit -------------------------------------------------------------------------------------
struct GzUnit {              // a run of consecutive stretches of one stream: ~1 MB of text
    uint32_t first, n;       // its stretches: results / token regions [first, first + n)
    uint32_t stream, pad;
    uint64_t sym_off;        // where its symbols go in the launch's symbol buffer (= where its text goes, counted over the launch)
    uint64_t text_len;
};
constexpr uint32_t GZ_MARK = 0x8000u;  // symbol: a byte, or GZ_MARK + i = byte i of the 32 KiB in front of the unit (i = 32767: the last one)
constexpr uint32_t GZ_WIN = 32768u;

template <int NT, int Q>
struct GzLds {
    uint16_t ring[GZ_WIN + Q];  // the last 32 Ki symbols + the window being made; position p lives at (p + GZ_WIN) mod (GZ_WIN + Q)
    uint16_t par[Q];
    uint32_t wsum[NT / 64 + 1];
    uint32_t ctl[8];
};

template <int NT, int Q>
__global__ __launch_bounds__(NT) void gz_resolve(const GzUnit* units, uint32_t n_units, const qd3::Unit* stretches, const qd3::Result* res, const uint32_t* use_slots,
                                                 const uint16_t* tokens, uint16_t* sym, uint16_t* wout, int32_t* status) {
    static_assert(Q % NT == 0 && Q / NT <= 32 && GZ_WIN % Q == 0, "a lane's share of a window");
    constexpr int K = Q / NT;
    constexpr uint32_t RN = GZ_WIN + Q;
    extern __shared__ __attribute__((aligned(16))) uint8_t gz_lds_raw[];
    GzLds<NT, Q>& S = *reinterpret_cast<GzLds<NT, Q>*>(gz_lds_raw);
    typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;
    lds_vu16* const ring = (lds_vu16*)S.ring;
    lds_vu16* const par = (lds_vu16*)S.par;
    const uint32_t ui = blockIdx.x, tid = threadIdx.x;
    if (ui >= n_units) return;
    const GzUnit U = units[ui];
    for (uint32_t k = tid; k < GZ_WIN; k += NT) ring[k] = (uint16_t)(GZ_MARK | k);
    if (tid < 8) S.ctl[tid] = 0;
    __syncthreads();
    auto rix = [](uint32_t p) { return (p + GZ_WIN) % RN; };  // (p: position in the unit's text; p + GZ_WIN >= 0 for what a match may reach)
    auto init_par = [&](uint32_t w) {
        const uint32_t rb = rix(w * Q);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t i = tid + (uint32_t)k * NT;
            par[i] = (uint16_t)(rb + i);
        }
        __syncthreads();
    };
    uint16_t* const out = sym + U.sym_off;
    const uint32_t olen = (uint32_t)U.text_len;
    // chains resolved inside window w (quade_inflate.hip's match stage, on ring indices), then the window's symbols leave
    auto finalize = [&](uint32_t w) {
        const uint32_t qb = w * Q, hi = min(qb + (uint32_t)Q, olen), rb = rix(qb);
        if (tid == 0) S.ctl[5] = S.ctl[6] = S.ctl[7] = 0;
        __syncthreads();
        uint32_t pend = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t i = tid + (uint32_t)k * NT;
            if (qb + i < hi && par[i] != (uint16_t)(rb + i)) pend |= 1u << k;
        }
        constexpr int MATCH_ROUNDS = 24;
#pragma unroll 1
        for (int round = 0; round < MATCH_ROUNDS; ++round) {
#pragma unroll 1
            for (uint32_t left = pend; left; left &= left - 1u) {
                const uint32_t k = (uint32_t)__builtin_ctz(left), i = tid + k * NT;
                const uint32_t q = par[i];
                const bool outside = q < rb || q >= rb + (uint32_t)Q;  // (the ring's other pages: final)
                if (outside || par[q - rb] == q) {
                    ring[rb + i] = ring[q];
                    par[i] = (uint16_t)(rb + i);
                    pend &= ~(1u << k);
                } else {
                    par[i] = par[q - rb];
                }
            }
            uint32_t* flag = &S.ctl[5];
            if (pend) flag[round % 3] = 1;
            __syncthreads();
            const uint32_t more = flag[round % 3];
            if (tid == 0) flag[(round + 2) % 3] = 0;
            if (!more) break;
            if (round == MATCH_ROUNDS - 1 && tid == 0) S.ctl[4] = 1;
        }
        __syncthreads();
        for (uint32_t i = tid; qb + i < hi; i += NT) out[qb + i] = ring[rb + i];
        __syncthreads();
    };
    uint32_t base_pos = 0, cur_win = 0;
    init_par(0);
#pragma unroll 1
    for (uint32_t su = U.first; su < U.first + U.n; ++su) {
        const uint16_t* tk = tokens + stretches[su].tok_off;
        const uint32_t n_slots = use_slots[su];
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < n_slots; c0 += NT * 4u) {
            const uint32_t s0 = c0 + 4u * tid;
            uint64_t four = 0;
            uint32_t prev = 0, next = 0;
            if (s0 < n_slots) {
                four = *reinterpret_cast<const uint64_t*>(tk + s0);
                if (s0) prev = tk[s0 - 1];
                if (s0 + 4u < n_slots) next = tk[s0 + 4u];
            }
            uint32_t sl[5], ln[4], at[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[j] = (uint32_t)(four >> (16 * j)) & 0xFFFFu;
            sl[4] = next;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool valid = s0 + (uint32_t)j < n_slots;
                const bool is_dist = ((j ? sl[j - 1] : prev) & qd3::TOK_MATCH) != 0;
                ln[j] = (!valid || is_dist) ? 0u : ((sl[j] & qd3::TOK_MATCH) ? (sl[j] & 0xFFu) + 3u : 1u);
                sum += ln[j];
            }
            uint32_t total;
            uint32_t pos = base_pos + block_scan_excl<NT>(sum, S.wsum, total);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                at[j] = pos;
                pos += ln[j];
            }
            const uint32_t chunk_hi = min(base_pos + total, olen);
            const uint32_t w_first = cur_win, w_last = max(w_first, chunk_hi ? (chunk_hi - 1u) / (uint32_t)Q : 0u);
#pragma unroll 1
            for (uint32_t w = w_first; w <= w_last; ++w) {
                if (w != cur_win) {
                    finalize(cur_win);
                    cur_win = w;
                    init_par(w);
                }
                const uint32_t wlo = w * Q, whi = min(wlo + (uint32_t)Q, olen), rb = rix(wlo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t d0 = at[j], len = ln[j];
                    if (len == 1u) {  // a literal: into its window's page when that window is the current one
                        if (d0 >= wlo && d0 < whi) ring[rb + (d0 - wlo)] = (uint16_t)sl[j];
                        continue;
                    }
                    if (len < 3u) continue;
                    const uint32_t dist = sl[j + 1] + 1u;
                    if (dist > d0 + GZ_WIN || d0 + len > olen) {  // reaches further back than DEFLATE's window, or beyond the unit's text
                        S.ctl[4] = 1;
                        continue;
                    }
                    const uint32_t a = max(d0, wlo), z = min(d0 + len, whi);
                    // (the source's place in the ring, from the window's: p - dist lies between 32 Ki in front of the window and its
                    //  end -- one wrap either way instead of a division per byte)
                    int32_t src = (int32_t)rb + (int32_t)(a - wlo) - (int32_t)dist;
                    src += src < 0 ? (int32_t)RN : 0;
                    for (uint32_t p = a; p < z; ++p) {
                        par[p - wlo] = (uint16_t)src;
                        ++src;
                        src -= src >= (int32_t)RN ? (int32_t)RN : 0;
                    }
                }
            }
            base_pos += total;
        }
    }
    finalize(cur_win);
    // the unit's last 32 Ki symbols: what the next unit's markers refer to
    {
        uint16_t* wo = wout + (size_t)ui * GZ_WIN;
        for (uint32_t j = tid; j < GZ_WIN; j += NT) wo[j] = ring[rix(olen - GZ_WIN + j)];  // (wraps below zero into the markers of this unit's own window)
    }
    if (tid == 0) status[ui] = (S.ctl[4] || base_pos != olen) ? QD_INFLATE_BAD_DISTANCE : 0;
}

// one workgroup per stream: its units in order; win_in[u] = the 32 KiB of text in front of unit u (bytes), carried[stream] = those behind
// the stream's last unit (for the next launch).  valid[stream]: how many bytes of the incoming window exist (a marker further back: error)
struct GzChain {
    uint32_t first_unit, n_units;  // the stream's units of this launch
    uint32_t valid;                // bytes of text in front of the first unit that exist (<= 32 Ki)
    uint32_t pad;
    uint8_t* carried;              // 32 KiB: in: the text in front of the first unit (its last `valid` bytes count); out: in front of the next launch
};
// Thread t owns symbols [32 t, 32 t + 32) of every unit's window: the next unit's 64 bytes of symbols are on their way (four 16-byte
// loads) while this unit's are mapped, so a step costs LDS work and one barrier, not a trip to memory per symbol (the first form
// loaded its symbols one by one inside the step: 16 us a unit, 6.8 ms for the longest stream's 420 units).
__global__ __launch_bounds__(1024) void gz_windows(const GzChain* chains, const GzUnit* units, const uint16_t* wout, uint8_t* win_in, int32_t* chain_status) {
    __shared__ __attribute__((aligned(16))) uint8_t W[2][GZ_WIN];
    __shared__ uint32_t bad;
    const GzChain C = chains[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (uint32_t j = tid; j < GZ_WIN / 4; j += 1024) reinterpret_cast<uint32_t*>(W[0])[j] = reinterpret_cast<const uint32_t*>(C.carried)[j];
    __syncthreads();
    uint32_t cur = 0;
    uint64_t have = C.valid;  // text that exists in front of the current unit (saturates at 32 Ki)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 nx[4];
    const uint32_t u_end = C.first_unit + C.n_units;
    auto fetch = [&](uint32_t u) {
        const u32x4* src = reinterpret_cast<const u32x4*>(wout + (size_t)u * GZ_WIN + 32u * tid);
#pragma unroll
        for (int k = 0; k < 4; ++k) nx[k] = src[k];
    };
    if (C.n_units) fetch(C.first_unit);
    for (uint32_t u = C.first_unit; u < u_end; ++u) {
        u32x4 sy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sy[k] = nx[k];
        if (u + 1 < u_end) fetch(u + 1);
        const uint64_t utext = units[u].text_len;
        {   // the window in front of this unit: what its markers (and gz_fixup's) refer to
            const u32x4* w = reinterpret_cast<const u32x4*>(W[cur] + 32u * tid);
            u32x4* wi = reinterpret_cast<u32x4*>(win_in + (size_t)u * GZ_WIN + 32u * tid);
            wi[0] = w[0];
            wi[1] = w[1];
        }
        const uint32_t floor_idx = have >= GZ_WIN ? 0u : GZ_WIN - (uint32_t)have;  // markers below it point in front of the stream's start
        const uint32_t own_from = GZ_WIN - (uint32_t)min((uint64_t)GZ_WIN, utext + have);
        uint32_t outw[8];
        bool any_bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t two = sy[k][q];
                uint32_t packed = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t s = (two >> (16 * h)) & 0xFFFFu, j = 32u * tid + 8u * (uint32_t)k + 2u * (uint32_t)q + (uint32_t)h;
                    uint32_t b = s;
                    if (s & GZ_MARK) {
                        const uint32_t i = s & 0x7FFFu;
                        if (i < floor_idx) {
                            // (a marker in front of the stream's start can only be one of this unit's own untouched window slots when the
                            //  unit's text is shorter than 32 Ki: those bytes do not exist and nobody may refer to them)
                            b = 0;
                            any_bad = any_bad || j >= own_from;
                        } else {
                            b = W[cur][i];
                        }
                    }
                    packed |= (b & 0xFFu) << (8 * h);
                }
                const int at = 8 * k + 2 * q;  // byte index within the thread's 32
                if ((at & 3) == 0) outw[at >> 2] = packed;
                else outw[at >> 2] |= packed << 16;
            }
        }
        if (any_bad) bad = 1;
        {
            u32x4* o = reinterpret_cast<u32x4*>(W[cur ^ 1] + 32u * tid);
            u32x4 a0, a1;
#pragma unroll
            for (int k = 0; k < 4; ++k) a0[k] = outw[k], a1[k] = outw[4 + k];
            o[0] = a0;
            o[1] = a1;
        }
        __syncthreads();
        cur ^= 1;
        have = min<uint64_t>(GZ_WIN, have + utext);
    }
    for (uint32_t j = tid; j < GZ_WIN / 4; j += 1024) reinterpret_cast<uint32_t*>(C.carried)[j] = reinterpret_cast<const uint32_t*>(W[cur])[j];
    __syncthreads();
    if (tid == 0) chain_status[blockIdx.x] = bad ? QD_INFLATE_BAD_DISTANCE : 0;
}

// symbols -> bytes: out[stream text] = a byte, or the unit's window byte a marker names.  One workgroup per GZ_FIX_TILE symbols of a unit.
constexpr uint32_t GZ_FIX_TILE = 16384;
struct GzFixTile {
    uint32_t unit, off;  // symbols [off, off + GZ_FIX_TILE) of the unit
};
__global__ __launch_bounds__(256) void gz_fixup(const GzFixTile* tiles, uint32_t n_tiles, const GzUnit* units, uint8_t* const* stream_out, const uint64_t* unit_out_off,
                                                const uint16_t* sym, const uint8_t* win_in, const uint32_t* unit_floor, int32_t* unit_status) {
    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const GzFixTile T = tiles[t];
    const GzUnit U = units[T.unit];
    const uint16_t* s = sym + U.sym_off + T.off;
    uint8_t* o = stream_out[U.stream] + unit_out_off[T.unit] + T.off;
    const uint8_t* W = win_in + (size_t)T.unit * GZ_WIN;
    const uint32_t n = (uint32_t)min<uint64_t>(GZ_FIX_TILE, U.text_len - T.off), floor_idx = unit_floor[T.unit];
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t v = s[i];
        uint32_t b = v;
        if (v & GZ_MARK) {
            const uint32_t k = v & 0x7FFFu;
            bad = bad || k < floor_idx;
            b = W[k];
        }
        o[i] = (uint8_t)b;
    }
    if (bad) unit_status[T.unit] = QD_INFLATE_BAD_DISTANCE;
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

__global__ void inflate3_jobs_from_blocks(const uint8_t* comp, uint8_t* out, const qd_inflate_block* blocks, const uint32_t* expect_crc, uint32_t n, qd_inflate3_job* jobs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const qd_inflate_block b = blocks[i];
    qd_inflate3_job j;
    j.payload = comp + b.in_off;
    j.out = out + b.out_off;
    j.in_len = b.in_len;
    j.out_len = b.out_len;
    j.expect_crc = expect_crc ? expect_crc[i] : 0u;
    j.check_crc = expect_crc ? 1u : 0u;
    jobs[i] = j;
}

template <class C, int WPB>
hipError_t launch_tokens_of(const qd3::Unit* units, const qd_inflate3_job* jobs, uint32_t n_units, uint16_t* tokens, uint32_t* lens, qd3::Result* res, uint32_t redo, hipStream_t st) {
    static const int wait_turns = env_int("QUADE_INFLATE3_WAIT", 256);
    static_assert((C::LANE_DW * 64 + C::RING_DW) % 4 == 0, "every wave's slice of the workgroup's LDS starts 16-byte aligned (the ring's LDS-DMA slots)");
    const size_t lds = ((size_t)C::LANE_DW * 64 + C::RING_DW) * 4 * WPB;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(inflate3_tokens<C, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint32_t groups = (n_units + 63) / 64, blocks = (groups + WPB - 1) / WPB;
    hipLaunchKernelGGL((inflate3_tokens<C, WPB>), dim3(redo ? std::min<uint32_t>(blocks, 128u) : blocks), dim3(64 * WPB), lds, st, units, jobs, n_units, tokens, lens, res,
                       (uint32_t)((wait_turns + C::ROUND_TURNS - 1) / C::ROUND_TURNS), redo);
    return hipGetLastError();
}
// every unit's tokens: the small configuration (four waves per CU), then the large one for the units it had no table space for
// (QUADE_INFLATE3_CFG=a: the large one alone, as the round's first form ran -- for A/B runs)
hipError_t launch_tokens(const qd3::Unit* units, const qd_inflate3_job* jobs, uint32_t n_units, uint16_t* tokens, uint32_t* lens, qd3::Result* res, hipStream_t st) {
    static const bool large_only = [] {
        const char* v = getenv("QUADE_INFLATE3_CFG");
        return v && (*v == 'a' || *v == 'A');
    }();
    if (large_only) return launch_tokens_of<CfgA, 1>(units, jobs, n_units, tokens, lens, res, 0, st);
    static const int wpb = env_int("QUADE_INFLATE3_WPB", 4);  // (A/B: 1 = a workgroup per wave, as the first form)
    const hipError_t e = wpb == 1 ? launch_tokens_of<CfgS, 1>(units, jobs, n_units, tokens, lens, res, 0, st)
                                  : launch_tokens_of<CfgS, 4>(units, jobs, n_units, tokens, lens, res, 0, st);
    return e != hipSuccess ? e : launch_tokens_of<CfgA, 1>(units, jobs, n_units, tokens, lens, res, 1, st);
}

template <int NT, int Q>
hipError_t launch_resolve(const qd_inflate3_job* jobs, uint32_t n_blocks, const uint16_t* tokens, const qd3::Result* res, int32_t* status, hipStream_t st) {
    const size_t lds = sizeof(R3Lds<NT, Q>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(inflate3_resolve_bgzf<NT, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((inflate3_resolve_bgzf<NT, Q>), dim3(n_blocks), dim3(NT), lds, st, jobs, n_blocks, tokens, res, status);
    return hipGetLastError();
}

struct Scratch3 {
    uint16_t* tokens;
    qd3::Result* res;
    uint32_t* lens;
    qd_inflate3_job* jobs;
};
Scratch3 carve(void* scratch, uint32_t n_blocks) {
    uint8_t* s = static_cast<uint8_t*>(scratch);
    Scratch3 c;
    c.tokens = reinterpret_cast<uint16_t*>(s);
    s += (((size_t)n_blocks * QD_INFLATE3_TOK_STRIDE * 2) + 255) & ~(size_t)255;
    c.res = reinterpret_cast<qd3::Result*>(s);
    s += (((size_t)n_blocks * sizeof(qd3::Result)) + 255) & ~(size_t)255;
    c.lens = reinterpret_cast<uint32_t*>(s);
    s += (((size_t)n_blocks * qd3::LENS_DW * 4) + 255) & ~(size_t)255;
    c.jobs = reinterpret_cast<qd_inflate3_job*>(s);
    return c;
}

}  // namespace

size_t qd_inflate3_scratch_bytes(uint32_t n_blocks) {
    const size_t n = n_blocks;
    return n * QD_INFLATE3_TOK_STRIDE * 2 + n * sizeof(qd3::Result) + n * qd3::LENS_DW * 4 + n * sizeof(qd_inflate3_job) + 4 * 256;
}

hipError_t qd_launch_inflate3_jobs(const qd_inflate3_job* d_jobs, uint32_t n_blocks, int32_t* status, void* scratch, hipStream_t st) {
    if (n_blocks == 0) return hipSuccess;
    if ((uintptr_t)scratch & 255u) return hipErrorInvalidValue;
    const Scratch3 c = carve(scratch, n_blocks);
    static const int shape = env_int("QUADE_INFLATE3_RESOLVE", 0);  // 0: 512 lanes, windows of 4 Ki (two workgroups per CU); 1: 1 024 lanes, 16 Ki; 2: 1 024 lanes, 4 Ki
    hipError_t e = launch_tokens(nullptr, d_jobs, n_blocks, c.tokens, c.lens, c.res, st);
    if (e != hipSuccess) return e;
    if (shape == 1) return launch_resolve<1024, 16384>(d_jobs, n_blocks, c.tokens, c.res, status, st);
    if (shape == 2) return launch_resolve<1024, 4096>(d_jobs, n_blocks, c.tokens, c.res, status, st);
    return launch_resolve<512, 4096>(d_jobs, n_blocks, c.tokens, c.res, status, st);
}

hipError_t qd_launch_inflate3(const uint8_t* comp, size_t comp_bytes, const qd_inflate_block* blocks, uint32_t n_blocks, uint8_t* out, int32_t* status, void* scratch,
                              hipStream_t st, const uint32_t* expect_crc) {
    (void)comp_bytes;
    if (n_blocks == 0) return hipSuccess;
    if ((uintptr_t)scratch & 255u) return hipErrorInvalidValue;
    const Scratch3 c = carve(scratch, n_blocks);
    hipLaunchKernelGGL(inflate3_jobs_from_blocks, dim3((n_blocks + 255) / 256), dim3(256), 0, st, comp, out, blocks, expect_crc, n_blocks, c.jobs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return qd_launch_inflate3_jobs(c.jobs, n_blocks, status, scratch, st);
}

// ======================================================================================================================================
// qd_gz: the host side of the gzip kernels above
#include <algorithm>
#include <vector>

#include "quade_text.h"

namespace {
struct GBuf {  // grow-only device allocation
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) qd_pool_put(p, cap);  // (drains the device first)
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 65536;
        return qd_pool_get(want, (void**)&p, &cap);  // (quade_pool.h: from what an earlier pipeline of this process released)
    }
    ~GBuf() {
        if (p) qd_pool_put(p, cap);
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};
struct HBuf {  // grow-only page-locked host allocation
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 4096;
        const hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    ~HBuf() {
        if (p) (void)hipHostFree(p);
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};
#define GZCHK(call)                     \
    do {                                \
        const hipError_t e_ = (call);   \
        if (e_ != hipSuccess) return e_; \
    } while (0)
}  // namespace

class qd_gz_impl {
  public:
    // per step, between decode() and resolve()
    struct Acc {
        std::vector<uint32_t> stretch;  // accepted stretches (indices into units_), in order
        std::vector<uint32_t> use_slots, text;
    };
    std::vector<qd3::Unit> units_;
    std::vector<Acc> acc_;
    GBuf d_stretch, d_found, d_units, d_res, d_tokens, d_lens, d_use, d_gzunits, d_chains, d_tiles, d_sym, d_wout, d_winin, d_ustat, d_cstat, d_outptr, d_uoff, d_floor,
        d_ranges, d_crc;
    HBuf h_found, h_res, h_back, h_up;
    std::vector<GzUnit> gzunits_;
    std::vector<uint32_t> unit_step_;  // R-unit -> step
    std::vector<qd_crc_range> ranges_;
    size_t n_runits_ = 0, n_chains_ = 0;
    qd_gz_stats st_{};
};

qd_gz::qd_gz() : p_(new qd_gz_impl()) {
    const int st = env_int("QUADE_GZ_STRETCH_KB", 0), ut = env_int("QUADE_GZ_UNIT_KB", 0);  // (measurement knobs)
    if (st > 0) stretch_bytes = (uint64_t)st << 10;
    if (ut > 0) unit_text = (uint64_t)ut << 10, unit_text_given = true;
}
qd_gz::~qd_gz() { delete p_; }
qd_gz_stats qd_gz::stats() const { return p_->st_; }

hipError_t qd_gz::reserve(uint64_t comp_bytes, uint64_t text_bytes) {
    qd_gz_impl& G = *p_;
    const uint64_t stretch = std::max<uint64_t>(stretch_bytes, 256), n_st = comp_bytes / stretch + 64, n_units = text_bytes / std::min<uint64_t>(std::max<uint64_t>(unit_text, 1024), 256u << 10) + n_st / 8 + 64;
    GZCHK(G.d_tokens.need((size_t)std::min<uint64_t>(comp_bytes * 8 + n_st * 8192, 0xFFFF0000ull * 2)));  // (four slots a byte: a second try at eight grows it)
    GZCHK(G.d_sym.need((size_t)text_bytes * 2 + 4096));
    GZCHK(G.d_stretch.need((size_t)n_st * sizeof(GzStretch)));
    GZCHK(G.d_found.need((size_t)n_st * 8));
    GZCHK(G.d_units.need((size_t)n_st * sizeof(qd3::Unit)));
    GZCHK(G.d_res.need((size_t)n_st * sizeof(qd3::Result)));
    GZCHK(G.d_lens.need((size_t)n_st * qd3::LENS_DW * 4));
    GZCHK(G.d_use.need((size_t)n_st * 4));
    GZCHK(G.d_wout.need((size_t)n_units * GZ_WIN * 2));
    GZCHK(G.d_winin.need((size_t)n_units * GZ_WIN));
    GZCHK(G.d_tiles.need((size_t)(text_bytes / GZ_FIX_TILE + n_units + 64) * sizeof(GzFixTile)));
    GZCHK(G.d_ranges.need((size_t)(text_bytes / 65536 + n_units + 64) * sizeof(qd_crc_range) + (n_units + 1) * 4 + 512));
    GZCHK(G.d_crc.need((size_t)(text_bytes / 65536 + 2 * n_units + 64) * 4));
    GZCHK(G.h_found.need((size_t)n_st * sizeof(GzStretch)));
    GZCHK(G.h_res.need((size_t)n_st * sizeof(qd3::Unit)));
    GZCHK(G.h_up.need((size_t)n_st * (sizeof(qd3::Unit) + 4) + (size_t)n_units * 64 + (size_t)(text_bytes / GZ_FIX_TILE + text_bytes / 65536) * 16 + 65536));
    return hipSuccess;
}

hipError_t qd_gz::decode(qd_gz_step* steps, int n, hipStream_t st, int slots_per_byte, bool text_filter) {
    qd_gz_impl& G = *p_;
    const uint64_t STRETCH = std::max<uint64_t>(stretch_bytes, 256);
    G.acc_.assign((size_t)n, qd_gz_impl::Acc());
    // 1. stretches: a known start per stream, a search range for every further one
    std::vector<GzStretch> probe;
    std::vector<std::pair<int, uint32_t>> probe_of;  // (step, stretch number)
    std::vector<uint32_t> n_st((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        qd_gz_step& s = steps[i];
        s.text_len = 0;
        s.bit_next = s.bit_start;
        s.member_end = 0;
        s.failed = 0;
        s.starved = 0;
        s.crc32 = 0;
        if (((uintptr_t)s.comp & 15u) || s.bit_start >= 8 * s.comp_bytes) {
            if ((uintptr_t)s.comp & 15u) s.failed = QD_INFLATE_TRUNCATED;
            continue;
        }
        const uint64_t byte0 = s.bit_start >> 3;
        const uint64_t k_max = (s.comp_bytes - byte0 + STRETCH - 1) / STRETCH;
        n_st[(size_t)i] = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(k_max, 1u << 20));
        for (uint32_t k = 1; k < n_st[(size_t)i]; ++k) {
            GzStretch q;
            q.base = reinterpret_cast<const uint32_t*>(s.comp);
            q.bit_from = 8 * (byte0 + k * STRETCH);
            q.bit_to = std::min<uint64_t>(8 * (byte0 + (k + 1) * STRETCH), 8 * s.comp_bytes);
            q.bit_end = 8 * s.comp_bytes;
            probe.push_back(q);
            probe_of.push_back({i, k});
        }
    }
    std::vector<uint64_t> found(probe.size(), ~0ull);
    if (!probe.empty()) {
        GZCHK(G.d_stretch.need(probe.size() * sizeof(GzStretch)));
        GZCHK(G.d_found.need(probe.size() * 8));
        GZCHK(G.h_found.need(probe.size() * std::max(sizeof(GzStretch), (size_t)8)));
        memcpy(G.h_found.p, probe.data(), probe.size() * sizeof(GzStretch));
        GZCHK(hipMemcpyAsync(G.d_stretch.p, G.h_found.p, probe.size() * sizeof(GzStretch), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gz_probe, dim3((uint32_t)probe.size()), dim3(64), GZ_PROBE_LDS, st, G.d_stretch.as<GzStretch>(), (uint32_t)probe.size(), G.d_found.as<uint64_t>(),
                           text_filter ? 1u : 0u);
        GZCHK(hipGetLastError());
        GZCHK(hipStreamSynchronize(st));  // (the table above has been read: the staging buffer takes the answer)
        GZCHK(hipMemcpyAsync(G.h_found.p, G.d_found.p, probe.size() * 8, hipMemcpyDeviceToHost, st));
        GZCHK(hipStreamSynchronize(st));
        memcpy(found.data(), G.h_found.p, probe.size() * 8);
        G.st_.stretches += (int64_t)probe.size();
    }
    // 2. units: from every start that was found to the next one
    struct Span {
        int step;
        uint64_t start, stop;
    };
    std::vector<std::vector<uint64_t>> starts((size_t)n);
    for (int i = 0; i < n; ++i)
        if (n_st[(size_t)i]) starts[(size_t)i].push_back(steps[i].bit_start);
    for (size_t q = 0; q < probe.size(); ++q)
        if (found[q] != ~0ull) starts[(size_t)probe_of[q].first].push_back(found[q]);
    // the streams' candidates on the device: a unit that runs past a false one stops at the next that is a block boundary
    std::vector<uint64_t> cand_all;
    std::vector<size_t> cand_at((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        std::vector<uint64_t>& v = starts[(size_t)i];
        std::sort(v.begin(), v.end());
        cand_at[(size_t)i] = cand_all.size();
        cand_all.insert(cand_all.end(), v.begin(), v.end());
    }
    cand_at[(size_t)n] = cand_all.size();
    GZCHK(G.d_found.need(std::max<size_t>(cand_all.size(), probe.size()) * 8 + 64));
    GZCHK(G.h_found.need(std::max<size_t>(cand_all.size() * 8, probe.size() * std::max(sizeof(GzStretch), (size_t)8)) + 64));
    if (!cand_all.empty()) {
        memcpy(G.h_found.p, cand_all.data(), cand_all.size() * 8);
        GZCHK(hipMemcpyAsync(G.d_found.p, G.h_found.p, cand_all.size() * 8, hipMemcpyHostToDevice, st));
    }
    auto make_unit = [&](int i, uint64_t start, uint64_t stop, uint64_t tok_off, uint32_t cap) {
        const qd_gz_step& s = steps[i];
        qd3::Unit u;
        u.base = reinterpret_cast<const uint32_t*>(s.comp);
        u.bit_start = start;
        u.bit_stop = stop;
        u.bit_end = 8 * s.comp_bytes;
        u.tok_off = tok_off;
        u.tok_cap = cap;
        u.wend = (uint32_t)std::min<uint64_t>((s.comp_bytes + 3) / 4 + 80, 0xFFFFFFF0u);
        u.cands = G.d_found.as<uint64_t>() + cand_at[(size_t)i];
        u.n_cands = (uint32_t)(cand_at[(size_t)i + 1] - cand_at[(size_t)i]);
        u.pad = 0;
        return u;
    };
    // Four slots per byte of the stretch: fastq makes at most two, so a unit may run on through a neighbour whose start was a false
    // candidate.  (A code has at least one bit and a token at most two slots: eight always hold -- the second try for text that
    // compresses beyond that, one symbol repeated under a Huffman-only coder.  A launch's slots are indexed with 32 bits: at four a
    // byte it takes a gigabyte of compressed bytes.)
    const uint64_t per_byte = slots_per_byte >= 8 ? 8 : 4;
    auto cap_for = [&](const qd_gz_step& s, uint64_t start, uint64_t stop) {
        const uint64_t bytes = ((stop == ~0ull ? 8 * s.comp_bytes : stop) - start + 7) / 8;
        return (uint32_t)std::min<uint64_t>((per_byte * bytes + 4096 + 3) & ~(uint64_t)3, 1u << 29);
    };
    G.units_.clear();
    std::vector<std::pair<int, uint32_t>> unit_of;  // unit -> (step, index among the step's starts)
    uint64_t tok_total = 0;
    for (int i = 0; i < n; ++i) {
        std::vector<uint64_t>& v = starts[(size_t)i];
        for (size_t k = 0; k < v.size(); ++k) {
            const uint64_t stop = k + 1 < v.size() ? v[k + 1] : ~0ull;
            const uint32_t cap = cap_for(steps[i], v[k], stop);
            if (tok_total + cap >= 0xFFFF0000ull) {  // a launch's slots are indexed with 32 bits: what does not fit waits for the next step
                v.resize(k);
                steps[i].starved = 1;
                break;
            }
            G.units_.push_back(make_unit(i, v[k], stop, tok_total, cap));
            unit_of.push_back({i, (uint32_t)k});
            tok_total += cap;
        }
        // (the last unit of a shortened list runs to the end of the input like any last unit: it stops for good at a block boundary)
        if (!v.empty() && !G.units_.empty() && unit_of.back().first == i) G.units_.back().bit_stop = ~0ull;
    }
    if (G.units_.empty()) return hipSuccess;
    const size_t nu0 = G.units_.size();
    // 3. tokens: one launch
    std::vector<qd3::Result> res(nu0);
    std::vector<uint8_t> dropped(nu0, 0);  // units whose start turned out to be no block boundary (their predecessor ran through them)
    {
        GZCHK(G.d_units.need(nu0 * sizeof(qd3::Unit)));
        GZCHK(G.d_res.need(nu0 * sizeof(qd3::Result)));
        GZCHK(G.d_lens.need(nu0 * qd3::LENS_DW * 4));
        GZCHK(G.d_tokens.need((size_t)tok_total * 2 + 64));
        GZCHK(G.h_res.need(nu0 * std::max(sizeof(qd3::Unit), sizeof(qd3::Result))));
        memcpy(G.h_res.p, G.units_.data(), nu0 * sizeof(qd3::Unit));
        GZCHK(hipMemcpyAsync(G.d_units.p, G.h_res.p, nu0 * sizeof(qd3::Unit), hipMemcpyHostToDevice, st));
        GZCHK(launch_tokens(G.d_units.as<qd3::Unit>(), nullptr, (uint32_t)nu0, G.d_tokens.as<uint16_t>(), G.d_lens.as<uint32_t>(), G.d_res.as<qd3::Result>(), st));
        GZCHK(hipStreamSynchronize(st));
        GZCHK(hipMemcpyAsync(G.h_res.p, G.d_res.p, nu0 * sizeof(qd3::Result), hipMemcpyDeviceToHost, st));
        GZCHK(hipStreamSynchronize(st));
        memcpy(res.data(), G.h_res.p, nu0 * sizeof(qd3::Result));
    }
    G.st_.units += (int64_t)nu0;
    static const bool debug = getenv("QUADE_GZ_DEBUG") != nullptr;
    if (debug) {
        size_t n_found = 0;
        for (uint64_t f : found) n_found += f != ~0ull;
        fprintf(stderr, "[qd_gz] decode: %d steps, %zu stretches probed, %zu block starts found, %zu units\n", n, probe.size(), n_found, nu0);
        uint64_t sl_sum = 0, sl_max = 0, bits_sum = 0, bits_max = 0;  // (a launch lasts as long as its longest lane)
        for (size_t q = 0; q < nu0; ++q) {
            const uint64_t bits = res[q].bit_next > G.units_[q].bit_start ? res[q].bit_next - G.units_[q].bit_start : 0;
            sl_sum += res[q].n_slots, sl_max = std::max<uint64_t>(sl_max, res[q].n_slots);
            bits_sum += bits, bits_max = std::max(bits_max, bits);
        }
        if (nu0) fprintf(stderr, "[qd_gz]   per unit: slots mean %llu max %llu, compressed bytes mean %llu max %llu\n", (unsigned long long)(sl_sum / nu0), (unsigned long long)sl_max,
                         (unsigned long long)(bits_sum / nu0 / 8), (unsigned long long)(bits_max / 8));
        for (size_t q = 0; q < nu0; ++q)
            if (res[q].status || dropped[q] || q < 3 || q + 2 >= nu0)
                fprintf(stderr, "[qd_gz]   unit %zu step %d%s: bits [%llu, stop %llu) end %llu -> status %u final %u slots %u text %u next %llu blk %llu/%u/%u\n", q, unit_of[q].first,
                        dropped[q] ? " (dropped)" : "", (unsigned long long)G.units_[q].bit_start, (unsigned long long)G.units_[q].bit_stop, (unsigned long long)G.units_[q].bit_end,
                        res[q].status, res[q].final_seen, res[q].n_slots, res[q].text_len, (unsigned long long)res[q].bit_next, (unsigned long long)res[q].blk_bit, res[q].blk_slots,
                        res[q].blk_text);
    }
    // 4. the chain from every stream's true start: a unit counts when it began exactly where its predecessor stopped
    for (int i = 0; i < n; ++i) {
        qd_gz_step& s = steps[i];
        if (s.failed || !n_st[(size_t)i]) continue;
        qd_gz_impl::Acc& A = G.acc_[(size_t)i];
        uint64_t at = s.bit_start;
        bool stop_here = false;
        for (size_t q = 0; q < nu0 && !stop_here; ++q) {
            if (unit_of[q].first != i) continue;
            const qd3::Unit& u = G.units_[q];
            const qd3::Result& r = res[q];
            if (u.bit_start < at) {  // its predecessor ran through this start: no block began there
                dropped[q] = 1;
                ++G.st_.chain_retries;
                continue;
            }
            if (u.bit_start != at) {  // (cannot happen: a unit stops on one of the stream's candidates, and all of them are units)
                s.failed = QD_INFLATE_CHAIN;
                break;
            }
            const bool last = u.bit_stop == ~0ull;
            if (r.status == 0 && (r.final_seen || (!last && r.bit_next >= u.bit_stop && r.bit_next <= u.bit_end))) {
                A.stretch.push_back((uint32_t)q);
                A.use_slots.push_back(r.n_slots);
                A.text.push_back(r.text_len);
                s.text_len += r.text_len;
                at = r.bit_next;
                if (r.final_seen) {
                    s.member_end = 1;
                    stop_here = true;
                }
                continue;
            }
            // it ran out of input (or of room) inside a block -- the step's last unit, or one that ran through every later candidate:
            // up to the last block boundary it passed
            const bool ran_out = r.status == QD_INFLATE_TRUNCATED || r.status == QD_INFLATE_TOKEN_SPACE || last;
            if (ran_out && r.blk_bit > u.bit_start) {
                A.stretch.push_back((uint32_t)q);
                A.use_slots.push_back(r.blk_slots);
                A.text.push_back(r.blk_text);
                s.text_len += r.blk_text;
                at = r.blk_bit;
                ++G.st_.partial_last;
            } else if (at == s.bit_start && (s.at_end || !ran_out || r.status == QD_INFLATE_TABLE_SPACE || r.status == QD_INFLATE_TOKEN_SPACE)) {
                s.failed = (int32_t)(r.status ? r.status : QD_INFLATE_TRUNCATED);  // nothing decoded, and more input will not help
            }
            // (a unit from a proven start that does not decode: what was proven so far counts; the next step starts at the damage, makes
            //  no progress there and gives the stream up)
            stop_here = true;
        }
        s.bit_next = at;
        if (debug)
            fprintf(stderr, "[qd_gz]   step %d: start %llu -> next %llu, text %llu, member_end %d, failed %d, %zu stretches accepted\n", i, (unsigned long long)s.bit_start,
                    (unsigned long long)at, (unsigned long long)s.text_len, s.member_end, s.failed, A.stretch.size());
        if (s.failed) {
            s.text_len = 0;
            s.bit_next = s.bit_start;
            s.member_end = 0;
            A = qd_gz_impl::Acc();
        }
    }
    if (text_filter) {
        // A stream with many stretches of which the probe found (next to) no block start: not text -- its real block headers were
        // refused with the false ones, and one lane would have to decode it all.  Everything once more with the plain probe.
        bool again = false;
        for (int i = 0; i < n; ++i) {
            if (n_st[(size_t)i] < 8) continue;
            const size_t cands = cand_at[(size_t)i + 1] - cand_at[(size_t)i];
            again = again || 16 * cands < n_st[(size_t)i];
        }
        if (again) {
            ++G.st_.plain_probes;
            return decode(steps, n, st, slots_per_byte, false);
        }
    }
    if (per_byte < 8) {
        bool again = false;
        for (int i = 0; i < n; ++i) again = again || steps[i].failed == QD_INFLATE_TOKEN_SPACE;
        if (again) return decode(steps, n, st, 8, text_filter);  // (everything once more: rare -- text that compresses beyond four tokens a byte)
    }
    return hipSuccess;
}

// QUADE_GZ_RESOLVE=1024: gz_resolve<1024, 4096> (one workgroup per CU, windows of 4 096 symbols) instead of <512, 2048> (two per CU) -- A/B
static bool resolve_wide() {
    static const bool wide = env_int("QUADE_GZ_RESOLVE", 512) == 1024;
    return wide;
}

hipError_t qd_gz::resolve(qd_gz_step* steps, int n, uint8_t* const* out, hipStream_t st) {
    qd_gz_impl& G = *p_;
    // Text per unit (one resolving workgroup, two of them per CU: 512 in flight).  Given (a test, a measurement): as given.  Else the
    // launch's text in whole rounds of 508 units of at most unit_text each -- 750 units of 2 MB took two rounds of which the second was
    // half empty (profiles/r05_e2e_gz_timeline_before.txt: 16.5 ms) -- and at least 256 KB: every unit costs a window of its own.
    uint64_t UNIT_TEXT = std::max<uint64_t>(unit_text, 1024);
    if (!unit_text_given) {
        uint64_t total = 0;
        for (int i = 0; i < n; ++i)
            if (!steps[i].failed) total += steps[i].text_len;
        const uint64_t per_round = resolve_wide() ? 254 : 508;  // (workgroups in flight: one of 1 024 lanes per CU, or two of 512)
        const uint64_t rounds = std::max<uint64_t>(1, (total + UNIT_TEXT * per_round - 1) / (UNIT_TEXT * per_round));
        UNIT_TEXT = std::max<uint64_t>(256u << 10, total / (rounds * per_round) + 1);
    }
    // units of ~1 MB of text: runs of accepted stretches; chains: a stream's units in order
    G.gzunits_.clear();
    G.unit_step_.clear();
    G.ranges_.clear();
    std::vector<GzChain> chains;
    std::vector<int> chain_step;
    std::vector<uint32_t> use((size_t)G.units_.size(), 0);
    std::vector<uint64_t> unit_out_off;
    std::vector<uint32_t> unit_floor;
    std::vector<GzFixTile> tiles;
    std::vector<uint32_t> first_range;  // unit -> its first CRC range (one more entry behind the last unit)
    uint64_t sym_total = 0;
    for (int i = 0; i < n; ++i) {
        const qd_gz_impl::Acc& A = G.acc_[(size_t)i];
        if (steps[i].failed || A.stretch.empty() || steps[i].text_len == 0) continue;
        GzChain c;
        c.first_unit = (uint32_t)G.gzunits_.size();
        c.valid = steps[i].carried_valid;
        c.pad = 0;
        c.carried = steps[i].carried;
        uint64_t text_before = 0;
        size_t k = 0;
        while (k < A.stretch.size()) {
            GzUnit u;
            u.first = A.stretch[k];
            u.n = 0;
            u.stream = (uint32_t)chains.size();
            u.pad = 0;
            u.sym_off = sym_total;
            u.text_len = 0;
            // (stretches of a unit must be neighbours in the unit table: a dropped candidate between two of them ends the unit)
            while (k < A.stretch.size() && A.stretch[k] == u.first + u.n && (u.n == 0 || u.text_len < UNIT_TEXT) && u.text_len + A.text[k] < 0xF0000000ull) {
                use[A.stretch[k]] = A.use_slots[k];
                u.text_len += A.text[k];
                ++u.n;
                ++k;
            }
            if (u.text_len == 0 && k < A.stretch.size()) continue;  // (stretches without text: nothing to resolve)
            const uint64_t have = std::min<uint64_t>(GZ_WIN, (uint64_t)steps[i].carried_valid + text_before);
            unit_floor.push_back((uint32_t)(GZ_WIN - have));
            unit_out_off.push_back(text_before);
            for (uint64_t o = 0; o < u.text_len; o += GZ_FIX_TILE) tiles.push_back(GzFixTile{(uint32_t)G.gzunits_.size(), (uint32_t)o});
            first_range.push_back((uint32_t)G.ranges_.size());
            for (uint64_t o = 0; o < u.text_len; o += 65536) G.ranges_.push_back(qd_crc_range{text_before + o, (uint32_t)std::min<uint64_t>(65536, u.text_len - o), 0});  // (the CRC kernel's ranges: <= 64 KiB)
            G.gzunits_.push_back(u);
            G.unit_step_.push_back((uint32_t)i);
            sym_total += u.text_len;
            text_before += u.text_len;
        }
        c.n_units = (uint32_t)G.gzunits_.size() - c.first_unit;
        chains.push_back(c);
        chain_step.push_back(i);
        steps[i].carried_valid = (uint32_t)std::min<uint64_t>(GZ_WIN, (uint64_t)steps[i].carried_valid + steps[i].text_len);
    }
    const size_t nr = G.gzunits_.size();
    G.n_runits_ = nr;
    G.n_chains_ = chains.size();
    if (!nr) return hipSuccess;
    // device tables
    const size_t nu = G.units_.size();
    GZCHK(G.d_units.need(nu * sizeof(qd3::Unit)));
    GZCHK(G.d_use.need(nu * 4));
    GZCHK(G.d_gzunits.need(nr * sizeof(GzUnit)));
    GZCHK(G.d_chains.need(chains.size() * sizeof(GzChain)));
    GZCHK(G.d_tiles.need(tiles.size() * sizeof(GzFixTile)));
    GZCHK(G.d_sym.need(sym_total * 2 + 64));
    GZCHK(G.d_wout.need(nr * (size_t)GZ_WIN * 2));
    GZCHK(G.d_winin.need(nr * (size_t)GZ_WIN));
    GZCHK(G.d_ustat.need(nr * 4 * 2));
    GZCHK(G.d_cstat.need(chains.size() * 4));
    GZCHK(G.d_outptr.need(chains.size() * sizeof(uint8_t*)));
    GZCHK(G.d_uoff.need(nr * 8));
    GZCHK(G.d_floor.need(nr * 4));
    first_range.push_back((uint32_t)G.ranges_.size());
    const size_t n_ranges = G.ranges_.size();
    GZCHK(G.d_ranges.need(n_ranges * sizeof(qd_crc_range) + (nr + 1) * 4 + 256));
    GZCHK(G.d_crc.need((nr + n_ranges) * 4));
    std::vector<uint8_t*> outp(chains.size());
    for (size_t c = 0; c < chains.size(); ++c) outp[c] = out[chain_step[c]];
    size_t up = 0;
    auto stage = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        memcpy(G.h_up.p + up, src, bytes);
        const hipError_t e = hipMemcpyAsync(dst, G.h_up.p + up, bytes, hipMemcpyHostToDevice, st);
        up += (bytes + 255) & ~(size_t)255;
        return e;
    };
    GZCHK(hipStreamSynchronize(st));  // (the staging buffer below may still be read by the previous step's copies)
    GZCHK(G.h_up.need(nu * (sizeof(qd3::Unit) + 4) + nr * (sizeof(GzUnit) + 8 + 4 + 4) + n_ranges * sizeof(qd_crc_range) + chains.size() * (sizeof(GzChain) + 8) +
                      tiles.size() * sizeof(GzFixTile) + 20 * 256));
    GZCHK(stage(G.d_units.p, G.units_.data(), nu * sizeof(qd3::Unit)));
    GZCHK(stage(G.d_use.p, use.data(), nu * 4));
    GZCHK(stage(G.d_gzunits.p, G.gzunits_.data(), nr * sizeof(GzUnit)));
    GZCHK(stage(G.d_chains.p, chains.data(), chains.size() * sizeof(GzChain)));
    GZCHK(stage(G.d_tiles.p, tiles.data(), tiles.size() * sizeof(GzFixTile)));
    GZCHK(stage(G.d_outptr.p, outp.data(), chains.size() * sizeof(uint8_t*)));
    GZCHK(stage(G.d_uoff.p, unit_out_off.data(), nr * 8));
    GZCHK(stage(G.d_floor.p, unit_floor.data(), nr * 4));
    GZCHK(hipMemsetAsync(G.d_ustat.p, 0, nr * 4 * 2, st));
    if (resolve_wide()) {
        constexpr int RNT = 1024, RQ = 4096;
        const size_t lds = sizeof(GzLds<RNT, RQ>);
        GZCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gz_resolve<RNT, RQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gz_resolve<RNT, RQ>), dim3((uint32_t)nr), dim3(RNT), lds, st, G.d_gzunits.as<GzUnit>(), (uint32_t)nr, G.d_units.as<qd3::Unit>(), G.d_res.as<qd3::Result>(),
                           G.d_use.as<uint32_t>(), G.d_tokens.as<uint16_t>(), G.d_sym.as<uint16_t>(), G.d_wout.as<uint16_t>(), G.d_ustat.as<int32_t>());
    } else {
        constexpr int RNT = 512, RQ = 2048;
        const size_t lds = sizeof(GzLds<RNT, RQ>);
        GZCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(gz_resolve<RNT, RQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gz_resolve<RNT, RQ>), dim3((uint32_t)nr), dim3(RNT), lds, st, G.d_gzunits.as<GzUnit>(), (uint32_t)nr, G.d_units.as<qd3::Unit>(), G.d_res.as<qd3::Result>(),
                           G.d_use.as<uint32_t>(), G.d_tokens.as<uint16_t>(), G.d_sym.as<uint16_t>(), G.d_wout.as<uint16_t>(), G.d_ustat.as<int32_t>());
    }
    GZCHK(hipGetLastError());
    hipLaunchKernelGGL(gz_windows, dim3((uint32_t)chains.size()), dim3(1024), 0, st, G.d_chains.as<GzChain>(), G.d_gzunits.as<GzUnit>(), G.d_wout.as<uint16_t>(), G.d_winin.p,
                       G.d_cstat.as<int32_t>());
    GZCHK(hipGetLastError());
    hipLaunchKernelGGL(gz_fixup, dim3((uint32_t)tiles.size()), dim3(256), 0, st, G.d_tiles.as<GzFixTile>(), (uint32_t)tiles.size(), G.d_gzunits.as<GzUnit>(),
                       G.d_outptr.as<uint8_t*>(), G.d_uoff.as<uint64_t>(), G.d_sym.as<uint16_t>(), G.d_winin.p, G.d_floor.as<uint32_t>(), G.d_ustat.as<int32_t>() + nr);
    GZCHK(hipGetLastError());
    // CRC-32 of every unit's text: ranges of 64 KiB (a stream's units lie in its own text buffer: one launch per stream), combined per unit
    {
        qd_crc_range* d_rg = G.d_ranges.as<qd_crc_range>();
        uint32_t* d_first = reinterpret_cast<uint32_t*>(G.d_ranges.p + ((n_ranges * sizeof(qd_crc_range) + 255) & ~(size_t)255));
        uint32_t* d_sub = G.d_crc.as<uint32_t>() + nr;
        GZCHK(stage(d_rg, G.ranges_.data(), n_ranges * sizeof(qd_crc_range)));
        GZCHK(stage(d_first, first_range.data(), (nr + 1) * 4));
        for (size_t c = 0; c < chains.size(); ++c) {
            const uint32_t r0 = first_range[chains[c].first_unit], r1 = first_range[chains[c].first_unit + chains[c].n_units];
            GZCHK(qd_text_crc32(outp[c], d_rg + r0, r1 - r0, d_sub + r0, st));
        }
        GZCHK(qd_text_crc32_combine(d_rg, d_sub, d_first, (uint32_t)nr, G.d_crc.as<uint32_t>(), 1, st));
    }
    GZCHK(G.h_back.need(nr * 12 + chains.size() * 4 + 64));
    GZCHK(hipMemcpyAsync(G.h_back.p, G.d_ustat.p, nr * 8, hipMemcpyDeviceToHost, st));
    GZCHK(hipMemcpyAsync(G.h_back.p + nr * 8, G.d_crc.p, nr * 4, hipMemcpyDeviceToHost, st));
    GZCHK(hipMemcpyAsync(G.h_back.p + nr * 12, G.d_cstat.p, chains.size() * 4, hipMemcpyDeviceToHost, st));
    // (chain c belongs to step chain_step[c]: finish() needs it)
    G.unit_step_.push_back(0xFFFFFFFFu);
    for (size_t c = 0; c < chains.size(); ++c) G.unit_step_.push_back((uint32_t)chain_step[c]);
    return hipSuccess;
}

uint32_t qd_crc32_combine_host(uint32_t crc1, uint32_t crc2, uint64_t len2);  // quade_io.cpp (zlib's)

hipError_t qd_gz::finish(qd_gz_step* steps, int n) {
    qd_gz_impl& G = *p_;
    const size_t nr = G.n_runits_;
    if (!nr) return hipSuccess;
    const int32_t* ustat = G.h_back.as<int32_t>();
    const uint32_t* crc = reinterpret_cast<const uint32_t*>(G.h_back.p + nr * 8);
    const int32_t* cstat = reinterpret_cast<const int32_t*>(G.h_back.p + nr * 12);
    std::vector<uint8_t> first((size_t)n, 1);
    for (size_t u = 0; u < nr; ++u) {
        const uint32_t i = G.unit_step_[u];
        if (i >= (uint32_t)n) continue;
        if (ustat[u] || ustat[nr + u]) steps[i].failed = ustat[u] ? ustat[u] : ustat[nr + u];
        steps[i].crc32 = first[i] ? crc[u] : qd_crc32_combine_host(steps[i].crc32, crc[u], G.gzunits_[u].text_len);
        first[i] = 0;
    }
    for (size_t c = 0; c < G.n_chains_; ++c) {
        const uint32_t i = G.unit_step_[nr + 1 + c];
        if (i < (uint32_t)n && cstat[c]) steps[i].failed = cstat[c];
    }
    if (getenv("QUADE_GZ_DEBUG")) {
        for (size_t u = 0; u < nr; ++u)
            if (ustat[u] || ustat[nr + u] || u < 2) fprintf(stderr, "[qd_gz] finish: unit %zu (step %u): resolve status %d, fixup status %d, crc %08x, text %llu\n", u, G.unit_step_[u], ustat[u],
                                                            ustat[nr + u], crc[u], (unsigned long long)G.gzunits_[u].text_len);
        for (size_t c = 0; c < G.n_chains_; ++c) fprintf(stderr, "[qd_gz] finish: chain %zu status %d\n", c, cstat[c]);
    }
    return hipSuccess;
}
