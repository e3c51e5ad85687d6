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

namespace {

using CfgA = qd3::Cfg<8, 7, 112>;   // 996 B of tables per lane (112 long symbols: what the fixed code has behind 8 bits) + the wave's 16 KB input ring = 78 KB: two waves per CU
static_assert((CfgA::LANE_DW * 64 + qd3::RING_DW) * 4 * 2 <= 160 * 1024, "two workgroups per CU");

// units[] (stretches of a gzip member) or jobs[] (BGZF: unit i = block i, its slot region i * QD_INFLATE3_TOK_STRIDE)
template <class C>
__global__ __launch_bounds__(64) void inflate3_tokens(const qd3::Unit* units, const qd_inflate3_job* jobs, uint32_t n_units, uint16_t* tokens, uint32_t* lens_scratch,
                                                      qd3::Result* res, uint32_t wait_rounds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds3[];
    const uint32_t lane = threadIdx.x, u = blockIdx.x * 64u + lane;
    uint32_t* const ring = lds3;  // the wave's input ring first (16-byte aligned slots), the lanes' tables behind it
    uint16_t* const tab = reinterpret_cast<uint16_t*>(lds3 + qd3::RING_DW + lane * (uint32_t)C::LANE_DW);
    qd3::Lane<C> L;
    {
        qd3::Unit un{nullptr, 0, ~0ull, 0, 0, 0, 0};
        if (u < n_units) {
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
        qd3::lane_init(L, un, tokens, lens_scratch + (size_t)(u < n_units ? u : 0) * qd3::LENS_DW);
        if (u >= n_units) {
            L.state = qd3::ST_DONE;
            L.status = 0;
        }
    }
    qd3::topup(L, ring, lane, u < n_units);
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
        qd3::ring_wait();  // what the last round requested is there ...
        qd3::landed_all(L);
        qd3::topup(L, ring, lane, dec);  // ... what it used up is requested again, and lands while this round runs
#pragma unroll 1
        for (int t = 0; t < qd3::ROUND_TURNS; ++t)
            if (L.state <= qd3::ST_STORED) qd3::turn<C>(L, tab, ring, lane);
        if (mh) ++waited;
    }
    if (u < n_units) qd3::lane_finish(L, res + u);
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

template <class C>
hipError_t launch_tokens(const qd3::Unit* units, const qd_inflate3_job* jobs, uint32_t n_units, uint16_t* tokens, uint32_t* lens, qd3::Result* res, hipStream_t st) {
    static const int wait_turns = env_int("QUADE_INFLATE3_WAIT", 256);
    const size_t lds = ((size_t)C::LANE_DW * 64 + qd3::RING_DW) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(inflate3_tokens<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(inflate3_tokens<C>, dim3((n_units + 63) / 64), dim3(64), lds, st, units, jobs, n_units, tokens, lens, res,
                       (uint32_t)((wait_turns + qd3::ROUND_TURNS - 1) / qd3::ROUND_TURNS));
    return hipGetLastError();
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
    hipError_t e = launch_tokens<CfgA>(nullptr, d_jobs, n_blocks, c.tokens, c.lens, c.res, st);
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
