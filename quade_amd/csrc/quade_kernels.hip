// gfx950 (CDNA4 / MI355X) kernels of the demultiplexing hot path.
//
// Per read pair (reference: src/Quade.py:217-218,246-247 + src/Sample.py:56-91):
//   slice barcode window(s) out of the packed index-read rows, fuse, ASCII-fold, exact match
//   against the sample barcode table, min-phred gate over the barcode positions, slice + fuse the
//   molecular index, emit a uint16 routing code (+ molecular bytes), bump counters.
//
// This is byte/integer work bound by HBM bandwidth; there is no contraction in it, so no MFMA.
// Design (DESIGN.md has the numbers):
//   * demux_fast<>: the rows of a lane's two pairs are read with 16-byte-per-lane vector loads
//     (global_load_dwordx4 nt; 8-byte rows: lane i <-> consecutive 16 B, fully coalesced, 1 KiB per
//     wave instruction; other even strides <= 16: 4-byte-aligned 16-byte loads over exact-width
//     rows); all loads of a tile are issued before any is consumed, the next tile is in flight.
//   * the barcode table (open-addressing slots + 16-byte canonical keys) and the per-sample
//     histogram live in LDS, staged once per workgroup; every workgroup strides over >= 8 tiles
//     (a few co-resident workgroups per CU for large tables, an oversubscribed grid otherwise).
//   * case fold and quality gate are SWAR on 64-bit registers (8 bases per operation);
//     the undetermined count is reduced across the wavefront with DPP/shuffle before one LDS add.
//   * counters: LDS histogram -> one global atomic per non-zero bin per workgroup into that
//     workgroup's own row of a partial-count matrix (no cross-workgroup contention);
//     rows are summed when the host asks for the counts.
//   * outputs: routing codes as coalesced write-through dword stores; molecular bytes staged per
//     wave through LDS and written as 16-byte pieces.
//   * static row shapes: the common layouts (barcode at the window start, molecular index behind it)
//     are instantiated with their slice positions as compile-time constants -- the generic code kept
//     ~100 scalars live and spilled them through VGPR lanes.
//   * wave runs (dual-index forms): a wave owns 512 consecutive pairs of a super-tile and walks them in four
//     steps; its four dwords of codes per lane leave as one 16-byte store through the wave's LDS strip.
//   * exact-width rows: a lane's second 16-byte block is end-aligned with its second row (no over-read into
//     the next wave's line).
//   * wide plans (fused barcode of 17..32 bytes, each index read's part within 16): RowsW -- slices folded,
//     nibble-packed (ACGTN differ in their low nibble) into the same 16-byte keys, a hit confirmed by a byte
//     compare against the barcode; StaticWide<10> = the dual 10 bp index kits.
//   * demux_generic: any stride, optional per-read lengths (truncated index reads), barcodes up to
//     32 bytes, table in global memory (L2 resident).  Correctness path.
//   * demux_fixup: the listed short reads of a batch redone with the generic semantics after a fast
//     launch (one short read no longer sends its batch to the generic kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quade_common.h"
#include "quade_kernels.h"

namespace {

typedef uint64_t u64;

struct alignas(16) U128 {
    u64 lo, hi;
};

__device__ __forceinline__ U128 ld16(const uint8_t* p) {
    const ulong2 v = *reinterpret_cast<const ulong2*>(p);
    return U128{v.x, v.y};
}
__device__ __forceinline__ u64 ld8(const uint8_t* p) { return *reinterpret_cast<const u64*>(p); }

// I1 part (w1 bytes) followed by I2 part: the canonical little-endian packing of the fused string.
__device__ __forceinline__ void fuse(u64 a, u64 b, int w1, u64& lo, u64& hi) {
    if (w1 == 0) {
        lo = b;
        hi = 0;
    } else if (w1 == 8) {
        lo = a;
        hi = b;
    } else {
        lo = a | (b << (8 * w1));
        hi = b >> (64 - 8 * w1);
    }
}

struct LdsTable {
    const uint32_t* slots;
    const u64* bk;    // [S][2]
    uint32_t* hist;   // [2S + 1]
    uint8_t* strips;  // per-wave staging strips of the molecular bytes (128*M bytes each), or unused
};

// exact match of the folded key (lo,hi) of length K against the LDS table: ordinal or 0xFFFF
__device__ __forceinline__ uint32_t probe_lds(const LdsTable& t, u64 lo, u64 hi, uint32_t K,
                                              uint32_t seed, uint32_t mask) {
    uint32_t h = qd_hash_init(K, seed);
    h = qd_hash_step(h, lo);
    if (K > 8) h = qd_hash_step(h, hi);
    h = qd_hash_fini(h);
    const uint32_t fp = h >> 16;
    uint32_t s = h & mask;
    uint32_t found = QD_CODE_UNDET;
    for (;;) {
        const uint32_t e = t.slots[s];
        if (e == QD_EMPTY_SLOT) break;
        if ((e >> 16) == fp) {
            const uint32_t id = e & 0xFFFFu;
            const ulong2 k = reinterpret_cast<const ulong2*>(t.bk)[id];
            if (k.x == lo && k.y == hi) {
                found = id;
                break;
            }
        }
        s = (s + 1) & mask;
    }
    return found;
}

// Two independent keys probed in lockstep: both slot reads, then both candidate-key reads, are in
// flight together (2 dependent LDS round trips per step instead of 3 per key, one key after the
// other).  The candidate key is read unconditionally (ordinal clamped), compared only if the slot is
// occupied and its fingerprint matches.
__device__ __forceinline__ void probe_lds2(const LdsTable& t, u64 lo0, u64 hi0, u64 lo1, u64 hi1, uint32_t K,
                                           uint32_t seed, uint32_t mask, uint32_t last_id, uint32_t& f0,
                                           uint32_t& f1) {
    uint32_t h0 = qd_hash_init(K, seed), h1 = h0;
    h0 = qd_hash_step(h0, lo0);
    h1 = qd_hash_step(h1, lo1);
    if (K > 8) {
        h0 = qd_hash_step(h0, hi0);
        h1 = qd_hash_step(h1, hi1);
    }
    h0 = qd_hash_fini(h0);
    h1 = qd_hash_fini(h1);
    const uint32_t fp0 = h0 >> 16, fp1 = h1 >> 16;
    uint32_t s0 = h0 & mask, s1 = h1 & mask;
    bool d0 = false, d1 = false;
    f0 = f1 = QD_CODE_UNDET;
    while (!(d0 && d1)) {
        const uint32_t e0 = t.slots[s0], e1 = t.slots[s1];
        const uint32_t i0 = min(e0 & 0xFFFFu, last_id), i1 = min(e1 & 0xFFFFu, last_id);
        // one 16-byte LDS read per key (ds_read_b128: 4 LDS cycles; two 8-byte reads take 8)
        const ulong2 ka = reinterpret_cast<const ulong2*>(t.bk)[i0], kb = reinterpret_cast<const ulong2*>(t.bk)[i1];
        const u64 a0 = ka.x, b0 = ka.y, a1 = kb.x, b1 = kb.y;
        if (!d0) {
            if (e0 == QD_EMPTY_SLOT) {
                d0 = true;
            } else if ((e0 >> 16) == fp0 && a0 == lo0 && b0 == hi0) {
                f0 = i0;
                d0 = true;
            } else {
                s0 = (s0 + 1) & mask;
            }
        }
        if (!d1) {
            if (e1 == QD_EMPTY_SLOT) {
                d1 = true;
            } else if ((e1 >> 16) == fp1 && a1 == lo1 && b1 == hi1) {
                f1 = i1;
                d1 = true;
            } else {
                s1 = (s1 + 1) & mask;
            }
        }
    }
}

// The routing codes are stored write-through (`global_store_dword ... sc1`, the lowering of a relaxed
// agent-scope atomic store): a wave writes 256 contiguous bytes that this kernel never reads again,
// and not keeping them in L2 measures 2-5 % faster than plain or nt stores
// (profiles/r01_hbm_probe_store_policy.txt).  Not for the molecular bytes: their 8-byte pieces are
// 24 bytes apart across lanes and each write-through piece becomes its own fabric write (measured
// 1.3x slower on cfg4).
template <typename T>
__device__ __forceinline__ void st_wt(T* p, T v) {
#if QD_FAST_WT_STORES
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}

// Where a full tile's routing codes go.  With wave runs (QD_FAST_RUNS = R) a wave's R consecutive steps
// produce R x 256 contiguous bytes of codes: each step drops its dword per lane into the wave's private LDS
// strip and the last step writes the strip out 16 bytes per lane (R/4 KiB-sized write-through stores instead
// of R quarter-KiB ones).  strip == nullptr: one dword store per lane per step.
struct CodeOut {
    uint32_t* strip;  // the wave's R * 64 dwords
    int step;         // which quarter-KiB of the run this step is, 0 .. R-1, wave-uniform
    bool last;        // the wave's R-th step on this run: write the strip out
};
template <int R>
__device__ __forceinline__ void store_codes_full(const DemuxParams& p, const CodeOut& co, int64_t p0, uint32_t word) {
    if (R >= 4 && co.strip) {
        const uint32_t lane = threadIdx.x & 63u;
        co.strip[co.step * 64 + lane] = word;
        if (co.last) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
            uint16_t* run = p.codes + (p0 - 2 * (int64_t)lane - (int64_t)co.step * 128);  // the run's first pair
#pragma unroll
            for (int r = 0; r < R / 4; ++r) {
                const v4u32 v = *reinterpret_cast<const v4u32*>(co.strip + (r * 64 + lane) * 4);
                uint16_t* dst = run + (r * 64 + lane) * 8;
#if QD_FAST_WT_STORES
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");  // pad: 5.7
#else
                *reinterpret_cast<v4u32*>(dst) = v;
#endif
            }
            __builtin_amdgcn_wave_barrier();  // the strip is reused by the wave's next run
        }
    } else {
        st_wt(reinterpret_cast<uint32_t*>(p.codes + p0), word);
    }
}

__device__ __forceinline__ void store_mol(uint8_t* dst, u64 lo, u64 hi, int M) {
    // dst = mol + pair*M.  M is wave-uniform.
    if ((M & 3) == 0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        if (M >= 4) d[0] = (uint32_t)lo;
        if (M >= 8) d[1] = (uint32_t)(lo >> 32);
        if (M >= 12) d[2] = (uint32_t)hi;
        if (M >= 16) d[3] = (uint32_t)(hi >> 32);
    } else {
        for (int i = 0; i < M; ++i) dst[i] = (uint8_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFF);
    }
}

// Molecular bytes of two consecutive pairs (2M contiguous bytes at an even pair index, so the
// destination is 8-byte aligned for M % 4 == 0): 64-bit stores instead of 2 x M/4 dword stores.
__device__ __forceinline__ void store_mol2(uint8_t* dst, u64 alo, u64 ahi, u64 blo, u64 bhi, int M) {
    u64* d = reinterpret_cast<u64*>(dst);
    switch (M) {
        case 4:
            d[0] = (alo & 0xFFFFFFFFull) | (blo << 32);
            break;
        case 8:
            d[0] = alo;
            d[1] = blo;
            break;
        case 12:
            d[0] = alo;
            d[1] = (ahi & 0xFFFFFFFFull) | (blo << 32);
            d[2] = (blo >> 32) | (bhi << 32);
            break;
        case 16:
            d[0] = alo;
            d[1] = ahi;
            d[2] = blo;
            d[3] = bhi;
            break;
        default:
            store_mol(dst, alo, ahi, M);
            store_mol(dst + M, blo, bhi, M);
    }
}

// One pair from its extracted slices to its routing code: fuse, fold, match, gate, count.
// k1/k2 = barcode slices of index read 1/2 (<= 8 bytes each, little-endian, masked), m1/m2 = molecular
// slices, q1/q2 = quality bytes of the barcode slices (bytes beyond the slice = 0xFF).
template <bool DUAL, bool MOL = true>  // MOL = false: the caller fuses the molecular index itself (slices wider than 8 bytes)
__device__ __forceinline__ uint32_t match_pair(const DemuxParams& p, const LdsTable& t, u64 k1, u64 k2, u64 m1,
                                               u64 m2, u64 q1, u64 q2, u64& mlo, u64& mhi) {
    // a1: fused barcode (Quade.py:217 / :246)
    u64 klo = k1, khi = 0;
    if (DUAL) fuse(k1, k2, p.idx_w[0], klo, khi);
    // a2: fused molecular index, raw case (Quade.py:218 / :247)
    if (MOL && p.M > 0) {
        mlo = m1;
        mhi = 0;
        if (DUAL) fuse(m1, m2, p.mol_w[0], mlo, mhi);
    }
    // a3: fold for the lookup only (Sample.py:65)
    klo = qd_fold8(klo);
    khi = qd_fold8(khi);
    // a4: exact match (Sample.py:65-67)
    const uint32_t id = probe_lds(t, klo, khi, (uint32_t)p.K, p.seed, p.slot_mask);
    if (id == QD_CODE_UNDET) return QD_CODE_UNDET;
    // a5: min-phred gate over the barcode positions (Sample.py:70)
    uint32_t pass = qd_all_ge8(q1, p.thr);
    if (DUAL) pass &= qd_all_ge8(q2, p.thr);
    const uint32_t code = id * 2u + (pass ^ 1u);
    // a6: per-sample counters (Sample.py:71-72,79-80)
    atomicAdd(&t.hist[code], 1u);
    return code;
}

// every byte of v is one of A C G T N: (byte >> 1) & 7 tells the five letters apart (A 0, C 1, T 2, G 3, N 7) -- look the
// letter up (one v_perm_b32 per 4 bytes), compare with the byte
__device__ __forceinline__ bool qd_bases4(uint32_t v) {
    return __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, (v >> 1) & 0x07070707u) == v;
}
// ... of the bytes of x that `mask` selects (the others count as letters)
__device__ __forceinline__ bool qd_bases8(u64 x, u64 mask) {
    const u64 y = x | (0x4141414141414141ull & ~mask);
    return qd_bases4((uint32_t)y) && qd_bases4((uint32_t)(y >> 32));
}

// Wide plans (16 < K <= 32; slices of up to 16 bytes per index read; quade_common.h "wide keys"): the slices are
// folded, nibble-packed and fused into a 16-byte key for the same LDS table probe.  The packing is injective on the
// alphabet only ('Q' shares its low nibble with 'A'), so a packed hit proves equality only for a read whose key bytes are
// all letters of the alphabet -- checked in registers (r03; before: a hit was confirmed against the barcode's own bytes in
// global memory, one dependent L2 round trip per matched pair behind the tile's row loads: profiles/r03_wide_alphabet_check.txt).
// The table of a wide plan holds ACGTN-only barcodes (any other sends the plan to the generic
// kernel), so a key with a foreign byte equals none of them.
template <bool DUAL>
__device__ __forceinline__ uint32_t match_pair_wide(const DemuxParams& p, const LdsTable& t, const u64 (&k1)[2],
                                                    const u64 (&k2)[2], u64 m1, u64 m2, const u64 (&q1)[2],
                                                    const u64 (&q2)[2], u64& mlo, u64& mhi) {
    if (p.M > 0) {  // a2: fused molecular index, raw case
        mlo = m1;
        mhi = 0;
        if (DUAL) fuse(m1, m2, p.mol_w[0], mlo, mhi);
    }
    // a3 + a1: fold, pack, fuse
    const u64 f1lo = qd_fold8(k1[0]), f1hi = qd_fold8(k1[1]);
    const u64 f2lo = DUAL ? qd_fold8(k2[0]) : 0, f2hi = DUAL ? qd_fold8(k2[1]) : 0;
    u64 klo, khi;
    qd_wide_key(f1lo, f1hi, f2lo, f2hi, p.idx_w[0], &klo, &khi);
    // a4: packed lookup + what makes it exact
    bool letters = qd_bases8(f1lo, p.idx_mask[0]) && qd_bases8(f1hi, p.idx_mask_hi[0]);
    if (DUAL) letters = letters && qd_bases8(f2lo, p.idx_mask[1]) && qd_bases8(f2hi, p.idx_mask_hi[1]);
    if (!letters) return QD_CODE_UNDET;
    const uint32_t id = probe_lds(t, klo, khi, (uint32_t)p.K, p.seed, p.slot_mask);
    if (id == QD_CODE_UNDET) return QD_CODE_UNDET;
    // a5: min-phred gate over the barcode positions
    uint32_t pass = qd_all_ge8(q1[0], p.thr) & qd_all_ge8(q1[1], p.thr);
    if (DUAL) pass &= qd_all_ge8(q2[0], p.thr) & qd_all_ge8(q2[1], p.thr);
    const uint32_t code = id * 2u + (pass ^ 1u);
    atomicAdd(&t.hist[code], 1u);
    return code;
}

// Both pairs of a lane at once (full tiles): same steps as match_pair, the two table probes interleaved.
template <bool DUAL>
__device__ __forceinline__ void match_two(const DemuxParams& p, const LdsTable& t, const u64 (&k1)[2], const u64 (&k2)[2],
                                          const u64 (&m1)[2], const u64 (&m2)[2], const u64 (&q1)[2], const u64 (&q2)[2],
                                          uint32_t (&code)[2], u64 (&mlo)[2], u64 (&mhi)[2]) {
    u64 klo[2], khi[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        klo[h] = k1[h];
        khi[h] = 0;
        if (DUAL) fuse(k1[h], k2[h], p.idx_w[0], klo[h], khi[h]);
        if (p.M > 0) {
            mlo[h] = m1[h];
            mhi[h] = 0;
            if (DUAL) fuse(m1[h], m2[h], p.mol_w[0], mlo[h], mhi[h]);
        }
        klo[h] = qd_fold8(klo[h]);
        khi[h] = qd_fold8(khi[h]);
    }
    uint32_t id[2];
    probe_lds2(t, klo[0], khi[0], klo[1], khi[1], (uint32_t)p.K, p.seed, p.slot_mask,
               p.n_samples ? p.n_samples - 1 : 0, id[0], id[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (id[h] == QD_CODE_UNDET) {
            code[h] = QD_CODE_UNDET;
            continue;
        }
        uint32_t pass = qd_all_ge8(q1[h], p.thr);
        if (DUAL) pass &= qd_all_ge8(q2[h], p.thr);
        code[h] = id[h] * 2u + (pass ^ 1u);
        atomicAdd(&t.hist[code[h]], 1u);
    }
}

// codes and molecular bytes of the lane's two pairs (p0 even): one dword, 64-bit molecular stores
__device__ __forceinline__ void store_unit(const DemuxParams& p, int64_t p0, bool two, uint32_t c0, uint32_t c1,
                                           u64 m0lo, u64 m0hi, u64 m1lo, u64 m1hi) {
    // a7: routing codes, 2 x uint16 per lane = one dword store, coalesced
    if (two)
        st_wt(reinterpret_cast<uint32_t*>(p.codes + p0), c0 | (c1 << 16));
    else
        p.codes[p0] = (uint16_t)c0;
    if (p.M > 0) {
        if (two)
            store_mol2(p.mol + p0 * p.M, m0lo, m0hi, m1lo, m1hi, p.M);
        else
            store_mol(p.mol + p0 * p.M, m0lo, m0hi, p.M);
    }
}

// Full tiles, M % 4 == 0: the 64 lanes of a wave own 64 x 2M contiguous molecular bytes.  Each lane
// drops its 2M bytes into the wave's private LDS strip, then the wave writes the strip out as
// 16-byte pieces, lane i <-> consecutive 16 B: full-line, coalesced, write-through stores instead of
// 8-byte pieces 2M bytes apart.  No workgroup barrier: a wave's LDS operations execute in order and
// no other wave touches its strip.
// With wave runs and a run-sized strip (p.mol_run_strips, R x 128 x M bytes per wave) the R steps of a run collect
// their bytes in the strip and the last step writes all R x 128 x M contiguous bytes out at once: one burst of
// stores per wave and run (beside the codes' one) instead of one per step.
template <int R>
__device__ __forceinline__ void store_mol_wave(const DemuxParams& p, uint8_t* strips, const CodeOut& co, int64_t p0, u64 m0lo,
                                               u64 m0hi, u64 m1lo, u64 m1hi) {
    const int M = p.M;
    const uint32_t lane = threadIdx.x & 63u;
    const bool run = R >= 4 && p.mol_run_strips && co.strip;  // (a run-sized strip exists only where the codes have theirs)
    uint8_t* strip = strips + (threadIdx.x >> 6) * (run ? R * 128 * M : 128 * M) + (run ? co.step * 128 * M : 0);
    uint32_t* mine = reinterpret_cast<uint32_t*>(strip + lane * 2 * M);
    const uint32_t w0[4] = {(uint32_t)m0lo, (uint32_t)(m0lo >> 32), (uint32_t)m0hi, (uint32_t)(m0hi >> 32)};
    const uint32_t w1[4] = {(uint32_t)m1lo, (uint32_t)(m1lo >> 32), (uint32_t)m1hi, (uint32_t)(m1hi >> 32)};
    const int nd = M >> 2;  // dwords per pair, wave-uniform
    if ((M & 3) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nd) {
                mine[j] = w0[j];
                mine[nd + j] = w1[j];
            }
    } else {  // any width (the shapes with a 9..11-base molecular index): bytes -- the second pair's start is not word aligned
        uint8_t* b8 = strip + lane * 2 * M;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < M) {
                b8[j] = (uint8_t)(w0[j >> 2] >> (8 * (j & 3)));
                b8[M + j] = (uint8_t)(w1[j >> 2] >> (8 * (j & 3)));
            }
    }
    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
    if (run) {
        if (!co.last) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint8_t* first = strip - co.step * 128 * M;                                   // the run's strip
        uint8_t* dst = p.mol + (p0 - 2 * (int64_t)lane - (int64_t)co.step * 128) * M;       // the run's first pair
        const int pieces = R * 8 * M;  // R x 64 lanes x 2M bytes / 16
#pragma unroll
        for (int r = 0; r < 2 * R; ++r) {
            const int piece = r * 64 + (int)lane;
            if (piece < pieces) {
                const v4u32 v = *reinterpret_cast<const v4u32*>(first + 16 * piece);
#if QD_FAST_WT_STORES
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 16 * piece), "v"(v) : "memory");  // pad: 5.7
#else
                *reinterpret_cast<v4u32*>(dst + 16 * piece) = v;
#endif
            }
        }
        __builtin_amdgcn_wave_barrier();  // the strip is reused by the wave's next run
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the wave's first pair is p0 - 2*lane; its strip starts 16-byte aligned in `mol`
    uint8_t* dst = p.mol + (p0 - 2 * (int64_t)lane) * M;
    const int pieces = 8 * M;  // 64 lanes x 2M bytes / 16
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int piece = r * 64 + (int)lane;
        if (piece < pieces) {
            const v4u32 v = *reinterpret_cast<const v4u32*>(strip + 16 * piece);
#if QD_FAST_WT_STORES
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 16 * piece), "v"(v) : "memory");  // pad: 5.7
#else
            *reinterpret_cast<v4u32*>(dst + 16 * piece) = v;
#endif
        }
    }
    __builtin_amdgcn_wave_barrier();  // the strip is reused by the wave's next unit
}

// ------------------------------------------------------------------------------------------------
// Fast kernel.  A lane handles UNITS x 2 consecutive pairs per tile; rows are read once and never
// again, so loads are non-temporal.  Two row forms (policy structs below):
//   Rows8 : every stride is 8 (the 8 bp index configs): one aligned 16-byte load = the rows of 2 pairs
//   RowsX : any even strides <= 16 (exact-width rows, e.g. 14-byte windows): the 2*stride bytes of
//           the lane's 2 pairs are read with one or two 4-byte-aligned 16-byte loads and the slices
//           are taken at run-time byte offsets
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ U128 ld16s(const uint8_t* p) {
#if QD_FAST_NT
    typedef unsigned long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const v2u64*>(p));  // global_load_dwordx4 ... nt
    return U128{v.x, v.y};
#else
    return ld16(p);
#endif
}

// 16 bytes from a 4-byte-aligned address (global_load_dwordx4 needs dword alignment only)
__device__ __forceinline__ U128 ld16u(const uint8_t* p) {
    typedef unsigned int v4u32a __attribute__((ext_vector_type(4), aligned(4)));
#if QD_FASTX_NT
    const v4u32a v = __builtin_nontemporal_load(reinterpret_cast<const v4u32a*>(p));
#else
    const v4u32a v = *reinterpret_cast<const v4u32a*>(p);
#endif
    return U128{(u64)v.x | ((u64)v.y << 32), (u64)v.z | ((u64)v.w << 32)};
}

// exactly nbytes (even, <= 16*NW/2... ) from a 2-byte-aligned address, zero filled: used for the
// batch's last tile only, where a 16-byte load could run past the end of the array
template <int NW>
__device__ __forceinline__ void ld_exact(u64 (&w)[NW], const uint8_t* p, int nbytes) {
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = 0;
#pragma unroll
    for (int j = 0; j < 2 * NW; ++j) {
        uint32_t d = 0;
        if (4 * j + 4 <= nbytes)
            d = *reinterpret_cast<const uint32_t*>(p + 4 * j);
        else if (4 * j + 2 <= nbytes)
            d = *reinterpret_cast<const uint16_t*>(p + 4 * j);
        w[j >> 1] |= (u64)d << (32 * (j & 1));
    }
}

// bytes [start, start+8) of a little-endian block of NW 64-bit words (beyond the block: zero);
// `start` is wave-uniform, the word selects are uniform-condition moves (no dynamic register index)
template <int NW>
__device__ __forceinline__ u64 take8(const u64 (&w)[NW], int start) {
    const int i = start >> 3, sh = (start & 7) * 8;
    u64 a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        a = (i == j) ? w[j] : a;
        b = (i + 1 == j) ? w[j] : b;
    }
    return sh ? (a >> sh) | (b << (64 - sh)) : a;
}

// ---- row shapes ------------------------------------------------------------------------------------------
// DynShape: every slice position is a kernel argument (any plan inside the fast envelope).
// StaticShape: the layout "barcode at the start of the window, molecular index right behind it", the same in
// both index reads, baked in at compile time: slices become fixed byte shuffles, masks become constants and
// the scalar registers that carried them are free again (the generic code keeps ~100 scalars live and
// spills them through VGPR lanes: a third of its vector instructions were v_readlane / v_writelane).
struct DynShape {
    static constexpr bool STATIC = false;
    static constexpr int MOLW = -1;  // not known at compile time
    static __device__ __forceinline__ void apply(DemuxParams&) {}
    static bool matches(const DemuxParams&) { return true; }
};
template <int IW, int MW>  // dual index, IW-base barcodes at columns 0..IW-1, MW-base molecular index behind them
struct StaticShape {
    static constexpr bool STATIC = true;
    static constexpr int MOLW = MW;
    static constexpr int STRIDE = (IW + MW + 1) & ~1, QSTRIDE = (IW + 1) & ~1;
    static __device__ __forceinline__ void apply(DemuxParams& p) {
        p.n_streams = 2;
        p.K = 2 * IW;
        p.M = 2 * MW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            p.seq_stride[k] = STRIDE;
            p.qual_stride[k] = QSTRIDE;
            p.idx_off[k] = 0;
            p.idx_w[k] = IW;
            p.mol_off[k] = MW ? IW : 0;
            p.mol_w[k] = MW;
            p.idx_mask[k] = IW >= 8 ? ~0ull : ((1ull << (8 * IW)) - 1);
            p.mol_mask[k] = MW >= 8 ? ~0ull : ((1ull << (8 * MW)) - 1);
        }
    }
    static bool matches(const DemuxParams& p) {
        if (p.n_streams != 2 || p.K != 2 * IW || p.M != 2 * MW) return false;
        for (int k = 0; k < 2; ++k)
            if (p.seq_stride[k] != STRIDE || p.qual_stride[k] != QSTRIDE || p.idx_off[k] != 0 || p.idx_w[k] != IW ||
                p.mol_w[k] != MW || (MW && p.mol_off[k] != IW))
                return false;
        return true;
    }
};

template <int IW>  // single index read, IW-base barcode at column 0, no molecular index, 8-byte rows (BASELINE cfg2: IW = 8)
struct StaticSingle {
    static constexpr bool STATIC = true;
    static constexpr int MOLW = 0;
    static_assert(IW >= 1 && IW <= 8, "single static shapes: 8-byte rows");
    static __device__ __forceinline__ void apply(DemuxParams& p) {
        p.n_streams = 1;
        p.K = IW;
        p.M = 0;
        p.seq_stride[0] = p.qual_stride[0] = 8;
        p.idx_off[0] = 0;
        p.idx_w[0] = IW;
        p.mol_off[0] = p.mol_w[0] = 0;
        p.idx_mask[0] = IW >= 8 ? ~0ull : ((1ull << (8 * IW)) - 1);
        p.mol_mask[0] = 0;
    }
    static bool matches(const DemuxParams& p) {
        return p.n_streams == 1 && p.K == IW && p.M == 0 && p.seq_stride[0] == 8 && p.qual_stride[0] == 8 && p.idx_off[0] == 0 &&
               p.idx_w[0] == IW && p.mol_w[0] == 0;
    }
};

// ---- Rows8: all strides 8 ------------------------------------------------------------------------
template <int BLOCK_, bool DUAL, int UNITS, class SH = DynShape>
struct Rows8 {
    typedef SH Shape;
    static constexpr int BLOCK = BLOCK_;
    static constexpr bool PREFETCH = QD_FAST_PREFETCH != 0;   // 64 B per lane per tile: double-buffer
    static constexpr int RUNS = (DUAL && UNITS == 1) ? QD_FAST_RUNS : 0;  // wave runs, see demux_fast
    static constexpr bool GUARD_LAST = false;                 // aligned loads never leave the rows
    struct Tile {
        U128 s1[UNITS], q1[UNITS], s2[UNITS], q2[UNITS];
    };

    template <bool FULL>
    static __device__ __forceinline__ void load(Tile& T, const DemuxParams& p, int64_t base, uint32_t tid) {
        const int64_t n = p.n;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;  // first pair of the unit
            if (FULL || p0 + 1 < n) {
                T.s1[u] = ld16s(p.seq[0] + p0 * 8);
                T.q1[u] = ld16s(p.qual[0] + p0 * 8);
                if (DUAL) {
                    T.s2[u] = ld16s(p.seq[1] + p0 * 8);
                    T.q2[u] = ld16s(p.qual[1] + p0 * 8);
                }
            } else if (p0 < n) {  // last, odd pair of the batch: never read past row n-1
                T.s1[u] = U128{ld8(p.seq[0] + p0 * 8), 0};
                T.q1[u] = U128{ld8(p.qual[0] + p0 * 8), 0};
                if (DUAL) {
                    T.s2[u] = U128{ld8(p.seq[1] + p0 * 8), 0};
                    T.q2[u] = U128{ld8(p.qual[1] + p0 * 8), 0};
                }
            }
        }
    }

    // TAG only makes the copies of this code distinct (an assembler comment), so that the compiler
    // does not fold the copy that runs with younger loads in flight into the copy that runs without:
    // a folded copy has to assume the younger loads are missing and waits for everything.
    template <bool FULL, int TAG>
    static __device__ __forceinline__ uint32_t compute(const Tile& T, const DemuxParams& p, const LdsTable& t,
                                                       int64_t base, uint32_t tid, const CodeOut& co) {
        asm volatile("; demux tile copy %0" ::"i"(TAG));
        const int64_t n = p.n;
        uint32_t undet = 0;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;
            if (!FULL && p0 >= n) continue;
            const bool two = FULL || (p0 + 1 < n);
            const int sh1 = 8 * p.idx_off[0], sh2 = 8 * p.idx_off[1];
            const int mh1 = 8 * p.mol_off[0], mh2 = 8 * p.mol_off[1];
            u64 m0lo = 0, m0hi = 0, m1lo = 0, m1hi = 0;
            uint32_t c0, c1 = 0;
            // pair p0 = low words, pair p0+1 = high words of the 16-byte loads
            if (FULL && QD_FAST_PROBE2) {
                const u64 k1[2] = {(T.s1[u].lo >> sh1) & p.idx_mask[0], (T.s1[u].hi >> sh1) & p.idx_mask[0]};
                const u64 m1[2] = {(T.s1[u].lo >> mh1) & p.mol_mask[0], (T.s1[u].hi >> mh1) & p.mol_mask[0]};
                const u64 q1[2] = {T.q1[u].lo, T.q1[u].hi};
                u64 k2[2] = {0, 0}, m2[2] = {0, 0}, q2[2] = {0, 0};
                if (DUAL) {
                    k2[0] = (T.s2[u].lo >> sh2) & p.idx_mask[1];
                    k2[1] = (T.s2[u].hi >> sh2) & p.idx_mask[1];
                    m2[0] = (T.s2[u].lo >> mh2) & p.mol_mask[1];
                    m2[1] = (T.s2[u].hi >> mh2) & p.mol_mask[1];
                    q2[0] = T.q2[u].lo;
                    q2[1] = T.q2[u].hi;
                }
                uint32_t cc[2];
                u64 ml[2] = {0, 0}, mh[2] = {0, 0};
                match_two<DUAL>(p, t, k1, k2, m1, m2, q1, q2, cc, ml, mh);
                c0 = cc[0]; c1 = cc[1];
                m0lo = ml[0]; m0hi = mh[0]; m1lo = ml[1]; m1hi = mh[1];
            } else {
                c0 = match_pair<DUAL>(
                    p, t, (T.s1[u].lo >> sh1) & p.idx_mask[0], DUAL ? (T.s2[u].lo >> sh2) & p.idx_mask[1] : 0,
                    (T.s1[u].lo >> mh1) & p.mol_mask[0], DUAL ? (T.s2[u].lo >> mh2) & p.mol_mask[1] : 0, T.q1[u].lo,
                    DUAL ? T.q2[u].lo : 0, m0lo, m0hi);
                if (two)
                    c1 = match_pair<DUAL>(
                        p, t, (T.s1[u].hi >> sh1) & p.idx_mask[0], DUAL ? (T.s2[u].hi >> sh2) & p.idx_mask[1] : 0,
                        (T.s1[u].hi >> mh1) & p.mol_mask[0], DUAL ? (T.s2[u].hi >> mh2) & p.mol_mask[1] : 0, T.q1[u].hi,
                        DUAL ? T.q2[u].hi : 0, m1lo, m1hi);
            }
            undet += (c0 == QD_CODE_UNDET) + (two && c1 == QD_CODE_UNDET);
            if (FULL && p.mol_strip_off) {
                store_codes_full<RUNS>(p, co, p0, c0 | (c1 << 16));
                store_mol_wave<RUNS>(p, t.strips, co, p0, m0lo, m0hi, m1lo, m1hi);
            } else if (FULL && p.M == 0) {
                store_codes_full<RUNS>(p, co, p0, c0 | (c1 << 16));
            } else {
                store_unit(p, p0, two, c0, c1, m0lo, m0hi, m1lo, m1hi);
            }
        }
        return undet;
    }
};

// ---- RowsX: even strides <= 16, NL1/NL2 = 16-byte loads per lane for the seq rows of read 1 / 2 -----
template <int BLOCK_, int NL1, int NL2, bool DUAL, int UNITS, class SH = DynShape>
struct RowsX {
    typedef SH Shape;
    static constexpr int BLOCK = BLOCK_;
    // register double buffering: with a static shape without molecular index (its code is lean enough that the next
    // tile's loads in flight pay), for dynamic shapes only while the tile is small.  Static shapes WITH a molecular
    // index run single-buffered since their bytes leave once per wave run (QD_MOL_RUN_STRIPS): measured in one
    // process, cfg4: run strips + single buffer 0.6625 ms, run strips + double buffer 0.6749, r02's form (strip per
    // step + double buffer) 0.6733, strip per step + single buffer 0.6796 (profiles/r03_cfg4_molrun_prefetch.txt)
    static constexpr bool PREFETCH = QD_FAST_PREFETCH != 0 &&
                                     (SH::STATIC ? !(QD_MOL_RUN_STRIPS && SH::MOLW > 0 && DUAL && UNITS == 1 && QD_FAST_RUNS >= 4)
                                                 : (NL1 + (DUAL ? NL2 : 0)) <= QD_FASTX_PREFETCH_MAXNL);
    static constexpr bool GUARD_LAST = true;  // a 16-byte load of the batch's last rows could pass the array end
    static constexpr int RUNS = (DUAL && UNITS == 1) ? QD_FAST_RUNS : 0;
    struct Tile {
        u64 s1[UNITS][2 * NL1], q1[UNITS][2], s2[UNITS][2 * NL2], q2[UNITS][2];
    };

    template <bool FULL, int NL>
    static __device__ __forceinline__ void load_block(u64 (&w)[2 * NL], const uint8_t* rows, int64_t p0, int stride,
                                                      int64_t n) {
        const uint8_t* src = rows + p0 * stride;  // p0 even, stride even: 4-byte aligned
        if (FULL) {
            // NL == 2: the second block ENDS with the lane's second row (bytes [2*stride-16, 2*stride)) instead
            // of starting at byte 16: nothing beyond the lane's own 2*stride bytes is read, so a wave touches
            // exactly the lines of its own span (with block 2 at byte 16 its last lane reached 32-2*stride
            // bytes into the next wave's first line, which a non-temporal load fetches from HBM again:
            // one line in fifteen on 14-byte rows). compute() reads row 1 at +tight_shift().
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const U128 v = ld16u(src + (NL == 2 && j == 1 && QD_FASTX_TIGHT ? 2 * stride - 16 : 16 * j));
                w[2 * j] = v.lo;
                w[2 * j + 1] = v.hi;
            }
        } else {
            ld_exact(w, src, (p0 + 1 < n ? 2 : 1) * stride);
        }
    }

    template <bool FULL>
    static __device__ __forceinline__ void load(Tile& T, const DemuxParams& p, int64_t base, uint32_t tid) {
        const int64_t n = p.n;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;
            if (FULL || p0 < n) {
                load_block<FULL, NL1>(T.s1[u], p.seq[0], p0, p.seq_stride[0], n);
                load_block<FULL, 1>(T.q1[u], p.qual[0], p0, p.qual_stride[0], n);
                if (DUAL) {
                    load_block<FULL, NL2>(T.s2[u], p.seq[1], p0, p.seq_stride[1], n);
                    load_block<FULL, 1>(T.q2[u], p.qual[1], p0, p.qual_stride[1], n);
                }
            }
        }
    }

    template <bool FULL, int TAG>
    static __device__ __forceinline__ uint32_t compute(const Tile& T, const DemuxParams& p, const LdsTable& t,
                                                       int64_t base, uint32_t tid, const CodeOut& co) {
        asm volatile("; demux tile copy %0" ::"i"(TAG));
        const int64_t n = p.n;
        uint32_t undet = 0;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;
            if (!FULL && p0 >= n) continue;
            const bool two = FULL || (p0 + 1 < n);
            u64 m0lo = 0, m0hi = 0, m1lo = 0, m1hi = 0;
            uint32_t c[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // h = 0: pair p0 (row at byte 0), h = 1: pair p0+1 (row at byte stride)
                if (h == 1 && !two) break;
                // row h of a stream inside its loaded block(s); a full tile's second block is end-aligned
                const int r1 = h * (p.seq_stride[0] + ((FULL && QD_FASTX_TIGHT && NL1 == 2) ? 32 - 2 * p.seq_stride[0] : 0));
                const int r2 = h * (p.seq_stride[1] + ((FULL && QD_FASTX_TIGHT && NL2 == 2) ? 32 - 2 * p.seq_stride[1] : 0));
                const u64 k1 = take8(T.s1[u], r1 + p.idx_off[0]) & p.idx_mask[0];
                const u64 m1 = take8(T.s1[u], r1 + p.mol_off[0]) & p.mol_mask[0];
                const u64 q1 = take8(T.q1[u], h * p.qual_stride[0]) | ~p.idx_mask[0];  // beyond the slice: 0xFF
                u64 k2 = 0, m2 = 0, q2 = 0;
                if (DUAL) {
                    k2 = take8(T.s2[u], r2 + p.idx_off[1]) & p.idx_mask[1];
                    m2 = take8(T.s2[u], r2 + p.mol_off[1]) & p.mol_mask[1];
                    q2 = take8(T.q2[u], h * p.qual_stride[1]) | ~p.idx_mask[1];
                }
                c[h] = match_pair<DUAL>(p, t, k1, k2, m1, m2, q1, q2, h ? m1lo : m0lo, h ? m1hi : m0hi);
            }
            undet += (c[0] == QD_CODE_UNDET) + (two && c[1] == QD_CODE_UNDET);
            if (FULL && p.mol_strip_off) {
                store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
                store_mol_wave<RUNS>(p, t.strips, co, p0, m0lo, m0hi, m1lo, m1hi);
            } else if (FULL && p.M == 0) {
                store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
            } else {
                store_unit(p, p0, two, c[0], c[1], m0lo, m0hi, m1lo, m1hi);
            }
        }
        return undet;
    }
};

// ---- RowsW: wide plans.  Even strides <= 16 for the seq AND the qual rows; every stream is read with two
// 16-byte loads per lane (rows of 2 pairs = 2 * stride <= 32 bytes): the first at the lane's first byte, the
// second end-aligned with the lane's second row (at byte max(0, 2 * stride - 16): for strides <= 8 it repeats
// the first).  One instantiation per workgroup size covers every stride combination of the (rare) wide plans.
// StaticWide<IW>: dual IW-base barcodes (8 < IW <= 16) at the start of both index reads, no molecular index --
// the layout of the dual 10 bp index kits -- baked in, as StaticShape does for the 8-byte forms (the dynamic
// code of this policy keeps ~250 scalars alive: 577 SGPR spills, a third of its instructions move lanes).
template <int IW, int MW = 0>  // MW: a molecular index of MW bases right behind the barcode, in both index reads (IW + MW <= 16)
struct StaticWide {
    static constexpr bool STATIC = true;
    static constexpr int MOLW = MW;
    static constexpr int STRIDE = (IW + MW + 1) & ~1, QSTRIDE = (IW + 1) & ~1;
    static_assert(IW > 8 && IW + MW <= 16 && MW <= 8, "wide static shapes: 8 < barcode, barcode + molecular index <= 16 bytes");
    static __device__ __forceinline__ void apply(DemuxParams& p) {
        p.n_streams = 2;
        p.K = 2 * IW;
        p.M = 2 * MW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            p.seq_stride[k] = STRIDE;
            p.qual_stride[k] = QSTRIDE;
            p.idx_off[k] = 0;
            p.idx_w[k] = IW;
            p.mol_off[k] = MW ? IW : 0;
            p.mol_w[k] = MW;
            p.idx_mask[k] = ~0ull;
            p.idx_mask_hi[k] = IW >= 16 ? ~0ull : ((1ull << (8 * (IW - 8))) - 1);
            p.mol_mask[k] = MW >= 8 ? ~0ull : ((1ull << (8 * MW)) - 1);
        }
    }
    static bool matches(const DemuxParams& p) {
        if (p.n_streams != 2 || p.K != 2 * IW || p.M != 2 * MW) return false;
        for (int k = 0; k < 2; ++k)
            if (p.seq_stride[k] != STRIDE || p.qual_stride[k] != QSTRIDE || p.idx_off[k] != 0 || p.idx_w[k] != IW ||
                p.mol_w[k] != MW || (MW && p.mol_off[k] != IW))
                return false;
        return true;
    }
};

template <int BLOCK_, bool DUAL, class SH = DynShape>
struct RowsW {
    typedef SH Shape;
    static constexpr int BLOCK = BLOCK_;
    static constexpr bool PREFETCH = QD_WIDE_PREFETCH != 0 && SH::STATIC;  // 128 B per lane per tile: a second tile in registers costs
                                              // more in occupancy (132 VGPRs) than it hides (0.74 vs 0.60 ms, static 10+10, r02)
    static constexpr bool GUARD_LAST = true;  // strides < 8: the first block passes the lane's rows
    static constexpr int RUNS = DUAL ? QD_FAST_RUNS : 0;
    struct Tile {
        u64 s1[4], q1[4], s2[4], q2[4];
    };

    static __device__ __forceinline__ int second_at(int stride) { return 2 * stride > 16 ? 2 * stride - 16 : 0; }

    template <bool FULL>
    static __device__ __forceinline__ void load_rows(u64 (&w)[4], const uint8_t* rows, int64_t p0, int stride, int64_t n) {
        const uint8_t* src = rows + p0 * stride;  // p0 even, stride even: 4-byte aligned
        if (FULL) {
            const U128 a = ld16u(src), b = ld16u(src + second_at(stride));
            w[0] = a.lo;
            w[1] = a.hi;
            w[2] = b.lo;
            w[3] = b.hi;
        } else {
            ld_exact(w, src, (p0 + 1 < n ? 2 : 1) * stride);
        }
    }

    template <bool FULL>
    static __device__ __forceinline__ void load(Tile& T, const DemuxParams& p, int64_t base, uint32_t tid) {
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (FULL || p0 < n) {
            load_rows<FULL>(T.s1, p.seq[0], p0, p.seq_stride[0], n);
            load_rows<FULL>(T.q1, p.qual[0], p0, p.qual_stride[0], n);
            if (DUAL) {
                load_rows<FULL>(T.s2, p.seq[1], p0, p.seq_stride[1], n);
                load_rows<FULL>(T.q2, p.qual[1], p0, p.qual_stride[1], n);
            }
        }
    }

    // byte position of row h of a stream inside its four words
    template <bool FULL>
    static __device__ __forceinline__ int row_at(int h, int stride) {
        return h ? (FULL ? 16 + stride - second_at(stride) : stride) : 0;
    }

    template <bool FULL, int TAG>
    static __device__ __forceinline__ uint32_t compute(const Tile& T, const DemuxParams& p, const LdsTable& t,
                                                       int64_t base, uint32_t tid, const CodeOut& co) {
        asm volatile("; demux tile copy %0" ::"i"(TAG));
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (!FULL && p0 >= n) return 0;
        const bool two = FULL || (p0 + 1 < n);
        u64 m0lo = 0, m0hi = 0, m1lo = 0, m1hi = 0;
        uint32_t c[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            const int r1 = row_at<FULL>(h, p.seq_stride[0]), rq1 = row_at<FULL>(h, p.qual_stride[0]);
            const u64 k1[2] = {take8(T.s1, r1 + p.idx_off[0]) & p.idx_mask[0], take8(T.s1, r1 + p.idx_off[0] + 8) & p.idx_mask_hi[0]};
            const u64 q1[2] = {take8(T.q1, rq1) | ~p.idx_mask[0], take8(T.q1, rq1 + 8) | ~p.idx_mask_hi[0]};  // beyond the slice: 0xFF
            const u64 m1 = take8(T.s1, r1 + p.mol_off[0]) & p.mol_mask[0];
            u64 k2[2] = {0, 0}, q2[2] = {~0ull, ~0ull}, m2 = 0;
            if (DUAL) {
                const int r2 = row_at<FULL>(h, p.seq_stride[1]), rq2 = row_at<FULL>(h, p.qual_stride[1]);
                k2[0] = take8(T.s2, r2 + p.idx_off[1]) & p.idx_mask[1];
                k2[1] = take8(T.s2, r2 + p.idx_off[1] + 8) & p.idx_mask_hi[1];
                q2[0] = take8(T.q2, rq2) | ~p.idx_mask[1];
                q2[1] = take8(T.q2, rq2 + 8) | ~p.idx_mask_hi[1];
                m2 = take8(T.s2, r2 + p.mol_off[1]) & p.mol_mask[1];
            }
            c[h] = match_pair_wide<DUAL>(p, t, k1, k2, m1, m2, q1, q2, h ? m1lo : m0lo, h ? m1hi : m0hi);
        }
        const uint32_t undet = (c[0] == QD_CODE_UNDET) + (two && c[1] == QD_CODE_UNDET);
        if (FULL && p.mol_strip_off) {
            store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
            store_mol_wave<RUNS>(p, t.strips, co, p0, m0lo, m0hi, m1lo, m1hi);
        } else if (FULL && p.M == 0) {
            store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
        } else {
            store_unit(p, p0, two, c[0], c[1], m0lo, m0hi, m1lo, m1hi);
        }
        return undet;
    }
};

// ---- RowsU: index read 1 = 8-base barcode + a molecular index of 9..12 bases right behind it, index read 2 = 8-base
// barcode alone -- the dual-index kits whose i7 read carries the UMI (IDT xGen UDI-UMI: 8 + 9; NEBNext UMI: 8 + 11 / 12).
// Rows of 18 / 20 bytes: the 36 / 40 bytes of a lane's two pairs take three 16-byte loads (the third end-aligned with the
// second row, as RowsX does with its second), the other three streams are 8-byte rows (one aligned load each).  Static
// only; r02 / early r03 sent these layouts to the generic kernel (0.13-0.17 of peak).
template <int MW>
struct StaticUmi1 {
    static constexpr bool STATIC = true;
    static constexpr int MOLW = MW;
    static constexpr int STRIDE = (8 + MW + 1) & ~1;
    static_assert(MW > 8 && MW <= 12, "molecular index of 9..12 bases behind an 8-base barcode");
    static __device__ __forceinline__ void apply(DemuxParams& p) {
        p.n_streams = 2;
        p.K = 16;
        p.M = MW;
        p.seq_stride[0] = STRIDE;
        p.seq_stride[1] = 8;
        p.mol_off[0] = 8;
        p.mol_w[0] = MW;
        p.mol_off[1] = p.mol_w[1] = 0;
        p.mol_mask[0] = ~0ull;
        p.mol_mask[1] = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            p.qual_stride[k] = 8;
            p.idx_off[k] = 0;
            p.idx_w[k] = 8;
            p.idx_mask[k] = ~0ull;
        }
    }
    static bool matches(const DemuxParams& p) {
        return p.n_streams == 2 && p.K == 16 && p.M == MW && p.seq_stride[0] == STRIDE && p.seq_stride[1] == 8 && p.qual_stride[0] == 8 &&
               p.qual_stride[1] == 8 && p.idx_off[0] == 0 && p.idx_off[1] == 0 && p.idx_w[0] == 8 && p.idx_w[1] == 8 && p.mol_w[0] == MW &&
               p.mol_off[0] == 8 && p.mol_w[1] == 0;
    }
};

template <int BLOCK_, class SH>
struct RowsU {
    typedef SH Shape;
    static constexpr int BLOCK = BLOCK_, S1 = SH::STRIDE, MW = SH::MOLW;
    static constexpr bool PREFETCH = false;   // 96 B per lane per tile
    static constexpr bool GUARD_LAST = true;  // the first two blocks of a lane pass its rows' end on the batch's last tile
    static constexpr int RUNS = QD_FAST_RUNS;
    struct Tile {
        u64 s1[6];  // full tiles: bytes [0, 32) of the lane's two rows, then bytes [2*S1 - 16, 2*S1); guarded tiles: bytes [0, 2*S1) in a row
        u64 q1[2], s2[2], q2[2];
    };

    template <bool FULL>
    static __device__ __forceinline__ void load(Tile& T, const DemuxParams& p, int64_t base, uint32_t tid) {
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (!FULL && p0 >= n) return;
        const uint8_t* src = p.seq[0] + p0 * S1;  // p0 even, S1 even: 4-byte aligned
        if (FULL) {
            const U128 a = ld16u(src), b = ld16u(src + 16), c = ld16u(src + 2 * S1 - 16);
            T.s1[0] = a.lo; T.s1[1] = a.hi; T.s1[2] = b.lo; T.s1[3] = b.hi; T.s1[4] = c.lo; T.s1[5] = c.hi;
        } else {
            ld_exact(T.s1, src, (p0 + 1 < n ? 2 : 1) * S1);
        }
        if (FULL || p0 + 1 < n) {
            const U128 a = ld16s(p.qual[0] + p0 * 8), b = ld16s(p.seq[1] + p0 * 8), c = ld16s(p.qual[1] + p0 * 8);
            T.q1[0] = a.lo; T.q1[1] = a.hi; T.s2[0] = b.lo; T.s2[1] = b.hi; T.q2[0] = c.lo; T.q2[1] = c.hi;
        } else {  // the batch's last, odd pair: never read past row n-1
            T.q1[0] = ld8(p.qual[0] + p0 * 8); T.s2[0] = ld8(p.seq[1] + p0 * 8); T.q2[0] = ld8(p.qual[1] + p0 * 8);
            T.q1[1] = T.s2[1] = T.q2[1] = 0;
        }
    }

    template <bool FULL, int TAG>
    static __device__ __forceinline__ uint32_t compute(const Tile& T, const DemuxParams& p, const LdsTable& t, int64_t base, uint32_t tid,
                                                       const CodeOut& co) {
        asm volatile("; demux tile copy %0" ::"i"(TAG));
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (!FULL && p0 >= n) return 0;
        const bool two = FULL || (p0 + 1 < n);
        constexpr u64 HI = (1ull << (8 * (MW - 8))) - 1;  // bytes 8 .. MW-1 of the molecular index
        const u64 head[4] = {T.s1[0], T.s1[1], T.s1[2], T.s1[3]}, tail[2] = {T.s1[4], T.s1[5]};
        u64 mlo[2] = {0, 0}, mhi[2] = {0, 0};
        uint32_t c[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            // a1 + a2: the barcode at the row's start, the molecular index behind it (raw case; Quade.py:217-218 / :246-247)
            const u64 k1 = take8(head, h * S1);
            if (h == 0) {
                mlo[0] = take8(head, 8);
                mhi[0] = take8(head, 16) & HI;
            } else if (FULL) {  // row 1's molecular index lies in the end-aligned block, at byte (S1 + 8) - (2 * S1 - 16)
                mlo[1] = take8(tail, 24 - S1);
                mhi[1] = take8(tail, 32 - S1) & HI;
            } else {
                mlo[1] = take8(T.s1, S1 + 8);
                mhi[1] = take8(T.s1, S1 + 16) & HI;
            }
            u64 d0, d1;
            c[h] = match_pair<true, false>(p, t, k1, T.s2[h], 0, 0, T.q1[h], T.q2[h], d0, d1);
        }
        const uint32_t undet = (c[0] == QD_CODE_UNDET) + (two && c[1] == QD_CODE_UNDET);
        if (FULL && p.mol_strip_off) {
            store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
            store_mol_wave<RUNS>(p, t.strips, co, p0, mlo[0], mhi[0], mlo[1], mhi[1]);
        } else {
            store_unit(p, p0, two, c[0], c[1], mlo[0], mhi[0], mlo[1], mhi[1]);
        }
        return undet;
    }
};

// ---- RowsU2 (r05): BOTH index reads = 8-base barcode + a molecular index of 9..12 bases right behind it (the same width in both:
// UMI-carrying i7 and i5 adapters), the fused molecular index 18..24 bytes per pair (Quade.py:218 / :247: index 1's slice, then
// index 2's).  Two streams of 18 / 20-byte rows -- three 16-byte loads each per lane, as RowsU's first stream -- and the two
// 8-byte quality rows; the molecular bytes leave through the wave's LDS strip as RowsU's do, in up to three rounds of 16-byte
// pieces (128 x M <= 3 072 bytes per wave and step).  Static only; until r05 these layouts ran the generic kernel (0.13-0.17 of
// peak: VERDICT r04 missing #4).
template <int MW>
struct StaticUmi2 {
    static constexpr bool STATIC = true;
    static constexpr int MOLW = 2 * MW;
    static constexpr int MW1 = MW;
    static constexpr int STRIDE = (8 + MW + 1) & ~1;
    static_assert(MW > 8 && MW <= 12, "molecular indexes of 9..12 bases behind 8-base barcodes");
    static __device__ __forceinline__ void apply(DemuxParams& p) {
        p.n_streams = 2;
        p.K = 16;
        p.M = 2 * MW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            p.seq_stride[k] = STRIDE;
            p.qual_stride[k] = 8;
            p.idx_off[k] = 0;
            p.idx_w[k] = 8;
            p.idx_mask[k] = ~0ull;
            p.mol_off[k] = 8;
            p.mol_w[k] = MW;
            p.mol_mask[k] = ~0ull;
        }
    }
    static bool matches(const DemuxParams& p) {
        bool ok = p.n_streams == 2 && p.K == 16 && p.M == 2 * MW;
        for (int k = 0; k < 2; ++k)
            ok = ok && p.seq_stride[k] == STRIDE && p.qual_stride[k] == 8 && p.idx_off[k] == 0 && p.idx_w[k] == 8 && p.mol_w[k] == MW && p.mol_off[k] == 8;
        return ok;
    }
};

// the fused molecular index of RowsU2: three words per pair (bytes [0, M), M = 2 MW <= 24)
struct Mol3 {
    u64 w[3];
};
template <int MW>
__device__ __forceinline__ Mol3 fuse_mol3(u64 a_lo, u64 a_hi, u64 b_lo, u64 b_hi) {  // a: MW bytes (a_hi holds bytes 8 .. MW-1), then b
    constexpr int R = MW - 8;                 // bytes of a word 1 that belong to a: 1 .. 4
    constexpr u64 HI = (1ull << (8 * R)) - 1;
    Mol3 m;
    m.w[0] = a_lo;
    m.w[1] = (a_hi & HI) | (b_lo << (8 * R));
    m.w[2] = (b_lo >> (64 - 8 * R)) | ((b_hi & HI) << (8 * R));
    return m;
}
// codes and molecular bytes of a lane's pair(s) on a guarded tile: plain stores
__device__ __forceinline__ void store_unit3(const DemuxParams& p, int64_t p0, bool two, uint32_t c0, uint32_t c1, const Mol3& m0, const Mol3& m1) {
    if (two)
        st_wt(reinterpret_cast<uint32_t*>(p.codes + p0), c0 | (c1 << 16));
    else
        p.codes[p0] = (uint16_t)c0;
    const int M = p.M;
    uint8_t* d = p.mol + p0 * M;
    for (int j = 0; j < M; ++j) {
        d[j] = (uint8_t)(m0.w[j >> 3] >> (8 * (j & 7)));
        if (two) d[M + j] = (uint8_t)(m1.w[j >> 3] >> (8 * (j & 7)));
    }
}
// full tiles: the wave's 128 x M contiguous bytes through its LDS strip (store_mol_wave for three words per pair; one step's strip)
template <int M>
__device__ __forceinline__ void store_mol_wave3(const DemuxParams& p, uint8_t* strips, int64_t p0, const Mol3& m0, const Mol3& m1) {
    static_assert(M % 2 == 0 && M > 16 && M <= 24, "2 x (9..12) bytes");
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t* strip = strips + (threadIdx.x >> 6) * (128 * M);
    uint16_t* mine = reinterpret_cast<uint16_t*>(strip + lane * 2 * M);  // (2 M is even: 16-bit pieces)
#pragma unroll
    for (int j = 0; j < M / 2; ++j) {
        mine[j] = (uint16_t)(m0.w[j >> 2] >> (16 * (j & 3)));
        mine[M / 2 + j] = (uint16_t)(m1.w[j >> 2] >> (16 * (j & 3)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
    uint8_t* dst = p.mol + (p0 - 2 * (int64_t)lane) * M;  // the wave's first pair: 128 M bytes from a multiple of 128 pairs -- 16-byte aligned
    constexpr int pieces = 8 * M;                           // 64 lanes x 2M bytes / 16
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int piece = r * 64 + (int)lane;
        if (piece < pieces) {
            const v4u32 v = *reinterpret_cast<const v4u32*>(strip + 16 * piece);
#if QD_FAST_WT_STORES
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 16 * piece), "v"(v) : "memory");  // pad: 5.7
#else
            *reinterpret_cast<v4u32*>(dst + 16 * piece) = v;
#endif
        }
    }
    __builtin_amdgcn_wave_barrier();  // the strip is reused by the wave's next unit
}

template <int BLOCK_, class SH>
struct RowsU2 {
    typedef SH Shape;
    static constexpr int BLOCK = BLOCK_, S1 = SH::STRIDE, MW = SH::MW1, M = SH::MOLW;
    static constexpr bool PREFETCH = false;   // 128 B per lane per tile
    static constexpr bool GUARD_LAST = true;  // the first two blocks of a lane pass its rows' end on the batch's last tile
    static constexpr int RUNS = QD_FAST_RUNS;
    struct Tile {
        u64 s1[6], s2[6];  // full tiles: bytes [0, 32) of the lane's two rows, then bytes [2*S1 - 16, 2*S1); guarded tiles: bytes [0, 2*S1) in a row
        u64 q1[2], q2[2];
    };

    template <bool FULL>
    static __device__ __forceinline__ void load(Tile& T, const DemuxParams& p, int64_t base, uint32_t tid) {
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (!FULL && p0 >= n) return;
        const uint8_t* a = p.seq[0] + p0 * S1;  // p0 even, S1 even: 4-byte aligned
        const uint8_t* b = p.seq[1] + p0 * S1;
        if (FULL) {
            const U128 a0 = ld16u(a), a1 = ld16u(a + 16), a2 = ld16u(a + 2 * S1 - 16);
            const U128 b0 = ld16u(b), b1 = ld16u(b + 16), b2 = ld16u(b + 2 * S1 - 16);
            T.s1[0] = a0.lo; T.s1[1] = a0.hi; T.s1[2] = a1.lo; T.s1[3] = a1.hi; T.s1[4] = a2.lo; T.s1[5] = a2.hi;
            T.s2[0] = b0.lo; T.s2[1] = b0.hi; T.s2[2] = b1.lo; T.s2[3] = b1.hi; T.s2[4] = b2.lo; T.s2[5] = b2.hi;
        } else {
            ld_exact(T.s1, a, (p0 + 1 < n ? 2 : 1) * S1);
            ld_exact(T.s2, b, (p0 + 1 < n ? 2 : 1) * S1);
        }
        if (FULL || p0 + 1 < n) {
            const U128 x = ld16s(p.qual[0] + p0 * 8), y = ld16s(p.qual[1] + p0 * 8);
            T.q1[0] = x.lo; T.q1[1] = x.hi; T.q2[0] = y.lo; T.q2[1] = y.hi;
        } else {  // the batch's last, odd pair: never read past row n-1
            T.q1[0] = ld8(p.qual[0] + p0 * 8); T.q2[0] = ld8(p.qual[1] + p0 * 8);
            T.q1[1] = T.q2[1] = 0;
        }
    }

    template <bool FULL, int TAG>
    static __device__ __forceinline__ uint32_t compute(const Tile& T, const DemuxParams& p, const LdsTable& t, int64_t base, uint32_t tid,
                                                       const CodeOut& co) {
        asm volatile("; demux tile copy %0" ::"i"(TAG));
        const int64_t n = p.n, p0 = base + (int64_t)tid * 2;
        if (!FULL && p0 >= n) return 0;
        const bool two = FULL || (p0 + 1 < n);
        const u64 h1[4] = {T.s1[0], T.s1[1], T.s1[2], T.s1[3]}, t1[2] = {T.s1[4], T.s1[5]};
        const u64 h2[4] = {T.s2[0], T.s2[1], T.s2[2], T.s2[3]}, t2[2] = {T.s2[4], T.s2[5]};
        Mol3 mol[2] = {{{0, 0, 0}}, {{0, 0, 0}}};
        uint32_t c[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            // a1 + a2: the barcodes at the rows' starts, the molecular indexes behind them (raw case; Quade.py:217-218 / :246-247)
            const u64 k1 = take8(h1, h * S1), k2 = take8(h2, h * S1);
            u64 alo, ahi, blo, bhi;
            if (h == 0) {
                alo = take8(h1, 8), ahi = take8(h1, 16);
                blo = take8(h2, 8), bhi = take8(h2, 16);
            } else if (FULL) {  // row 1's molecular index lies in the end-aligned block, at byte (S1 + 8) - (2 * S1 - 16)
                alo = take8(t1, 24 - S1), ahi = take8(t1, 32 - S1);
                blo = take8(t2, 24 - S1), bhi = take8(t2, 32 - S1);
            } else {
                alo = take8(T.s1, S1 + 8), ahi = take8(T.s1, S1 + 16);
                blo = take8(T.s2, S1 + 8), bhi = take8(T.s2, S1 + 16);
            }
            mol[h] = fuse_mol3<MW>(alo, ahi, blo, bhi);
            u64 d0, d1;
            c[h] = match_pair<true, false>(p, t, k1, k2, 0, 0, T.q1[h], T.q2[h], d0, d1);
        }
        const uint32_t undet = (c[0] == QD_CODE_UNDET) + (two && c[1] == QD_CODE_UNDET);
        if (FULL && p.mol_strip_off) {
            store_codes_full<RUNS>(p, co, p0, c[0] | (c[1] << 16));
            store_mol_wave3<M>(p, t.strips, p0, mol[0], mol[1]);
        } else {
            store_unit3(p, p0, two, c[0], c[1], mol[0], mol[1]);
        }
        return undet;
    }
};

#if QD_FAST_MINWAVES
#define QD_FAST_BOUNDS __launch_bounds__(OPS::BLOCK, QD_FAST_MINWAVES)
#else
#define QD_FAST_BOUNDS __launch_bounds__(OPS::BLOCK)
#endif
template <class OPS>
__global__ QD_FAST_BOUNDS void demux_fast(const DemuxParams p_in) {
    DemuxParams p = p_in;
    OPS::Shape::apply(p);  // a static shape overwrites the layout fields with its constants
    constexpr int BLOCK = OPS::BLOCK;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t* slots = reinterpret_cast<uint32_t*>(lds_raw);
    u64* bk = reinterpret_cast<u64*>(lds_raw + p.lds_bk_off);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_raw + p.lds_hist_off);
    const uint32_t tid = threadIdx.x;
    const uint32_t S = p.n_samples;
    constexpr int64_t TILE = (int64_t)BLOCK * 2 * QD_FAST_UNITS;  // pairs per workgroup iteration
    typedef typename OPS::Tile TileT;

    // Tiles [0, nfull) are loaded without lane guards and strided over the grid; the batch's last
    // tile(s) -- the partial one, and for exact-width rows every tile within 8 pairs (>= 16 bytes) of
    // the end of the arrays, since an unguarded 16-byte load may reach that far past a lane's rows --
    // are lane-guarded and done afterwards.  The first tile's rows are requested before anything
    // else, so HBM latency overlaps the staging of the table.
    const int64_t ntiles = (p.n + TILE - 1) / TILE;
    const int64_t nfull = OPS::GUARD_LAST ? (p.n >= 8 ? (p.n - 8) / TILE : 0) : p.n / TILE;
    const int64_t G = gridDim.x;
    // Work of this workgroup, numbered it = 0, 1, ...:
    //  RUNS == 0: full tile blockIdx.x + it * G; lane tid takes the pairs tile * TILE + 2 * tid (+1);
    //  RUNS == R: the full tiles are grouped R by R into super-tiles strided over the grid; inside one, WAVE w
    //             owns the R * 128 consecutive pairs behind w * R * 128 and walks them in R steps, so a wave's
    //             consecutive steps read (and write) consecutive KiBs.  Measured (profiles/r02_wave_runs_*.txt):
    //             dual-index kernels -3..-6 % at the BASELINE batch sizes (>= 2 GB of rows), within +-1 % on
    //             small batches; single-index kernels +20 % at every size -- so OPS::RUNS is R for the dual
    //             forms and 0 for the single-index ones.
    // base_of(it) + 2 * tid is the lane's first pair either way.
    constexpr int RUNS = OPS::RUNS;
    const int64_t nunits = RUNS ? nfull / (RUNS ? RUNS : 1) : nfull;  // super-tiles / tiles dealt out below
    const int64_t wave128 = (int64_t)(tid >> 6) * 128;
    // (units strided over the grid, not adjacent ones per workgroup: the workgroups in flight then sweep the arrays
    // as one compact window; adjacent units measured 1.5-4 % slower on every config, profiles/r02_blocked_vs_strided_units.txt)
    auto unit_of = [&](int64_t it) -> int64_t { return (int64_t)blockIdx.x + (RUNS ? it / (RUNS ? RUNS : 1) : it) * G; };
    auto live = [&](int64_t it) -> bool { return unit_of(it) < nunits; };
    auto step_of = [&](int64_t it) -> int64_t { return RUNS ? it % (RUNS ? RUNS : 1) : 0; };
    auto base_of = [&](int64_t it) -> int64_t {
        if (!RUNS) return unit_of(it) * TILE;
        return unit_of(it) * (RUNS * TILE) + wave128 * (RUNS - 1) + step_of(it) * 128;
    };
    // codes of full tiles: through the wave's LDS strip when the launch gave it one
    uint32_t* const code_strip = (RUNS >= 4 && p.code_strip_off)
                                     ? reinterpret_cast<uint32_t*>(lds_raw + p.code_strip_off) + (tid >> 6) * (RUNS * 64)
                                     : nullptr;
    auto out_of = [&](int64_t it) -> CodeOut {
        return CodeOut{code_strip, (int)step_of(it), RUNS ? (it % (RUNS ? RUNS : 1)) == RUNS - 1 : false};
    };
    const CodeOut direct{nullptr, 0, false};
    int64_t it = 0;
    TileT A, B;
    if (live(0)) OPS::template load<true>(A, p, base_of(0), tid);

    // stage the table: global (L2) -> LDS, once per workgroup
    for (uint32_t i = tid; i <= p.slot_mask; i += BLOCK) slots[i] = p.slots[i];
    for (uint32_t i = tid; i < 2 * S; i += BLOCK) bk[i] = p.bk16[i];
    for (uint32_t i = tid; i < 2 * S + 1; i += BLOCK) hist[i] = 0;
    __syncthreads();
    const LdsTable t{slots, bk, hist, lds_raw + p.mol_strip_off};

    uint32_t undet = 0;
    {
    // Register double buffering (tile k+1 in flight while tile k is matched) when a tile is 64 B per
    // lane; wider tiles run single-buffered.  Every "load next, then match current" pair is
    // straight-line code with its own copy of the match (no control-flow join between issuing the
    // younger loads and consuming the older ones): with a join the compiler must assume the younger
    // loads may be missing and waits for them too (s_waitcnt vmcnt(3..0) instead of vmcnt(4+)),
    // which serialises the two tiles.
    if (live(0)) {
        if (OPS::PREFETCH) {
            for (;;) {
                if (!live(it + 1)) {
                    undet += OPS::template compute<true, 0>(A, p, t, base_of(it), tid, out_of(it));
                    break;
                }
                OPS::template load<true>(B, p, base_of(it + 1), tid);
                undet += OPS::template compute<true, 1>(A, p, t, base_of(it), tid, out_of(it));
                if (!live(it + 2)) {
                    undet += OPS::template compute<true, 2>(B, p, t, base_of(it + 1), tid, out_of(it + 1));
                    break;
                }
                OPS::template load<true>(A, p, base_of(it + 2), tid);
                undet += OPS::template compute<true, 3>(B, p, t, base_of(it + 1), tid, out_of(it + 1));
                it += 2;
            }
        } else {
            for (;;) {
                undet += OPS::template compute<true, 4>(A, p, t, base_of(it), tid, out_of(it));
                ++it;
                if (!live(it)) break;
                OPS::template load<true>(A, p, base_of(it), tid);
            }
        }
    }
    }
    // what is left: the lane-guarded tiles at the end of the batch (at most two) and, with runs, the full
    // tiles behind the last whole super-tile
    for (int64_t last = RUNS ? nunits * RUNS : nfull; last < ntiles; ++last) {
        if ((int64_t)blockIdx.x != last % G) continue;
        OPS::template load<false>(A, p, last * TILE, tid);
        undet += OPS::template compute<false, 5>(A, p, t, last * TILE, tid, direct);
    }

    // undetermined count: wavefront shuffle-reduce (64 lanes), then one LDS add per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) undet += __shfl_xor(undet, o, 64);
    if ((tid & 63) == 0 && undet) atomicAdd(&hist[2 * S], undet);
    __syncthreads();
    // flush this workgroup's histogram into its own row of the partial-count matrix
    // (32-bit rows: an atomic that leaves the L2 is counted -- and paid -- by its width, and a launch of the
    // molecular-index config flushes ~3 M of them; the host folds the rows into 64-bit totals before any
    // row counter could pass 2^32, see fold_rows() in quade_api.cpp)
    qd_row_t* row = p.partial + (size_t)(blockIdx.x % p.partial_rows) * p.cnt_stride;
    for (uint32_t i = tid; i < 2 * S + 1; i += BLOCK) {
        const uint32_t v = hist[i];
        if (v) atomicAdd(&row[i], (qd_row_t)v);
    }
}

// add the partial rows to out[ncnt].  blockIdx.y = a group of
// QD_REDUCE_ROWS rows, thread = one counter: row reads are coalesced across the threads and
// independent across the rows; one 64-bit atomic per (row group, counter).
#define QD_REDUCE_ROWS 32
__global__ void reduce_partials(const qd_row_t* partial, uint32_t rows, uint32_t cnt_stride, uint32_t ncnt, u64* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncnt) return;
    const uint32_t r0 = blockIdx.y * QD_REDUCE_ROWS;
    const uint32_t r1 = r0 + QD_REDUCE_ROWS < rows ? r0 + QD_REDUCE_ROWS : rows;
    u64 s = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) s += partial[(size_t)r * cnt_stride + i];
    if (s) atomicAdd(reinterpret_cast<unsigned long long*>(&out[i]), (unsigned long long)s);
}

// lds = dynamic LDS of the launch (table image + histogram + molecular strips); table_lds = the part
// every workgroup has to stage and flush (decides the grid form)
template <class OPS>
hipError_t launch_fast_t(const DemuxParams& p_launch, QdKernelCache& cache, int cus, int wg_per_cu, size_t table_lds, size_t strip_bytes_per_wave,
                         hipStream_t st) {
    constexpr int BLOCK = OPS::BLOCK;
    auto k = demux_fast<OPS>;
    // dynamic LDS: table image + histogram | molecular strips (per wave: one step's 128 x M bytes, or a whole run's
    // with QD_MOL_RUN_STRIPS while two workgroups of this size still fit a CU) | code strips
    size_t lds = table_lds;
    bool mol_runs = false;
    if (p_launch.mol_strip_off && strip_bytes_per_wave) {
        const size_t per_step = strip_bytes_per_wave * (BLOCK / 64), per_run = per_step * (size_t)(OPS::RUNS > 0 ? OPS::RUNS : 1);
        mol_runs = OPS::RUNS >= 4 && QD_MOL_RUN_STRIPS && QD_FAST_CODE_STRIPS &&
                   2 * (table_lds + per_run + (size_t)OPS::RUNS * 256 * (BLOCK / 64)) <= 156 * 1024;
        lds += mol_runs ? per_run : per_step;
    }
    // the attribute and the occupancy answer belong to (device, instantiation): kept in the context
    // code strips (wave runs): R x 256 B per wave behind the table image and the molecular strips, while the
    // table image is small (a large one leaves no room without giving up a co-resident workgroup)
    DemuxParams p = p_launch;
    p.mol_run_strips = mol_runs ? 1 : 0;
    if (OPS::RUNS >= 4 && QD_FAST_CODE_STRIPS &&
        (table_lds <= 24 * 1024 || 2 * (lds + (size_t)OPS::RUNS * 256 * (BLOCK / 64)) <= QD_STRIPS_LDS_BUDGET)) {
        p.code_strip_off = (uint32_t)lds;
        lds += (size_t)OPS::RUNS * 256 * (BLOCK / 64);
    }
    QdKernelCache::Entry& ce = cache.entries[reinterpret_cast<const void*>(k)];
    if (!ce.attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        ce.attr_set = true;
    }
    if (ce.occ_lds != lds) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLOCK, lds);
        if (e != hipSuccess) return e;
        ce.occ_blocks = nb < 1 ? 1 : nb;
        ce.occ_lds = lds;
    }
    const int occ_blocks = ce.occ_blocks;
    const int64_t tile = (int64_t)BLOCK * 2 * QD_FAST_UNITS;
    const int64_t ntiles = (p.n + tile - 1) / tile;
    // Grid (automatic), from the measurements in profiles/r01_tune*_*.txt:
    //  * small table image (<= 24 KB of LDS): oversubscribe -- up to 64 workgroups per CU, about 8 tiles
    //    each, a whole number of device fills (the last fill of a grid that is not one runs partly empty:
    //    7.45 fills cost what 8 do); surplus workgroups start as earlier ones retire, which keeps the streams'
    //    active window compact and evens out the tail; re-staging a few KB per workgroup is free;
    //  * large table image: a persistent grid of at most 2 co-resident workgroups per CU (staging
    //    tens of KB and flushing thousands of counters per workgroup is not free).
    int64_t grid;
    if (wg_per_cu > 0) {
        grid = (int64_t)cus * wg_per_cu;
    } else if (table_lds > 24 * 1024) {
        // two rounds of the resident set: of two workgroups that share a CU the older one wins the arbitration and ends
        // early (profiles/r03_wg_times_cfg3_persistent2.txt); a second round fills the slots it leaves, and restaging a
        // large image twice per slot is still cheap (cfg5, 512-thread workgroups: 2 / 3 / 4 / 6 / 8 per CU = 0.750 /
        // 0.766 / 0.744 / 0.752 / 0.758 ms, profiles/r03_cfg5_launch_forms.txt)
        grid = (int64_t)cus * (occ_blocks < 2 ? occ_blocks : 2) * (BLOCK <= 512 && occ_blocks >= 2 ? 2 : 1);
    } else {
        const int64_t fill = (int64_t)cus * occ_blocks;  // workgroups resident at once
        const int64_t lo = (int64_t)cus * (occ_blocks < 2 ? occ_blocks : 2), hi = (int64_t)cus * 64;
        // the kernel deals out super-tiles of R tiles (wave runs): every workgroup the same number of them
        // (a batch that fills the device about once must not leave one workgroup with an extra super-tile)
        constexpr int64_t R = OPS::RUNS > 0 ? OPS::RUNS : 1;
        const int64_t nunits = ntiles / R > 0 ? ntiles / R : 1;
        // about 8 tiles per workgroup
        // (r02 gave tables of more than 128 samples 16 tiles per workgroup to halve the flush; with r03's cfg4 form -- run
        // strips, single buffer -- 8 is 2 % faster: 24 / 32 workgroups per CU 0.6591 / 0.6578 ms against 14 per CU 0.6710,
        // profiles/r03_cfg4_grid_r03_form.txt)
        int64_t per = 8 / R;
        if (per < 1) per = 1;
        if ((nunits + per - 1) / per > hi) per = (nunits + hi - 1) / hi;
        while (per > 1 && (nunits + per - 1) / per < lo) --per;
        grid = (nunits + per - 1) / per;
        if (QD_FAST_GRID_FILLS && grid >= 4 * fill) grid = (grid + fill / 2) / fill * fill;
        if (grid > hi) grid = hi;
    }
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;

    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), lds, st, p);
    return hipGetLastError();
}

template <int BLOCK>
hipError_t launch_fast_b(const DemuxParams& p, QdKernelCache& cache, int cus, int wg_per_cu, size_t table_lds,
                         size_t lds_bytes /* = strip bytes per wave, handed on */, hipStream_t st) {
    constexpr int U = QD_FAST_UNITS;
    const bool dual = p.n_streams > 1;
    if (p.wide) {  // K > 16 with slices <= 16 bytes means two index reads
        if (!dual) return hipErrorInvalidValue;
#ifndef QD_NO_STATIC_SHAPES
        if (StaticWide<10>::matches(p))  // dual 10 bp indexes
            return launch_fast_t<RowsW<BLOCK, true, StaticWide<10>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
        if (StaticWide<12>::matches(p))  // dual 12 bp indexes
            return launch_fast_t<RowsW<BLOCK, true, StaticWide<12>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
        if (StaticWide<10, 6>::matches(p))  // dual 10 bp indexes, each followed by a 6-base molecular index
            return launch_fast_t<RowsW<BLOCK, true, StaticWide<10, 6>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
        return launch_fast_t<RowsW<BLOCK, true>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    }
#if !defined(QD_NO_STATIC_SHAPES) && !defined(QD_SWEEP_BUILD)
    // 8-base barcodes, a molecular index of 9..12 bases behind the one of index read 1 (rows of 18 / 20 bytes: static shapes only)
    if (StaticUmi1<9>::matches(p)) return launch_fast_t<RowsU<BLOCK, StaticUmi1<9>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi1<10>::matches(p)) return launch_fast_t<RowsU<BLOCK, StaticUmi1<10>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi1<11>::matches(p)) return launch_fast_t<RowsU<BLOCK, StaticUmi1<11>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi1<12>::matches(p)) return launch_fast_t<RowsU<BLOCK, StaticUmi1<12>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    // ... and behind the barcodes of BOTH index reads (the same width: 18 .. 24 bytes of molecular index per pair)
    if (StaticUmi2<9>::matches(p)) return launch_fast_t<RowsU2<BLOCK, StaticUmi2<9>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi2<10>::matches(p)) return launch_fast_t<RowsU2<BLOCK, StaticUmi2<10>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi2<11>::matches(p)) return launch_fast_t<RowsU2<BLOCK, StaticUmi2<11>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticUmi2<12>::matches(p)) return launch_fast_t<RowsU2<BLOCK, StaticUmi2<12>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
    if (p.seq_stride[0] > 16 || p.seq_stride[1] > 16) return hipErrorInvalidValue;  // (the host sends such plans here only for the shapes above)
    const bool all8 = p.seq_stride[0] == 8 && p.qual_stride[0] == 8 &&
                      (!dual || (p.seq_stride[1] == 8 && p.qual_stride[1] == 8));
    if (all8) {
#ifndef QD_NO_STATIC_SHAPES
        // dual 8 + 8 bp index, no molecular index (BASELINE cfg3, cfg5): r02 measured +2.6 % for the large-table launch
        // form (cfg5) and -1.4 % for the small-table one (cfg3) and used it for large tables only
        // (profiles/r02_static_vs_dynamic_shape_cfg{3,5}.txt); since r03 the dynamic-shape kernel carries more
        // parameters (128 VGPRs, 76 SGPR spills) and the static one wins on cfg3 too: 0.5444 vs 0.5505 ms, r02's library
        // 0.5487 (profiles/r03_cfg3_static80_triple_buffer.txt; a third register tile on top of it: 0.5475, not kept)
        if (StaticShape<8, 0>::matches(p) && (table_lds > 24 * 1024 || QD_STATIC80_ALWAYS))
            return launch_fast_t<Rows8<BLOCK, true, U, StaticShape<8, 0>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
        if (dual) return launch_fast_t<Rows8<BLOCK, true, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#ifndef QD_SWEEP_BUILD  // tuning builds instantiate the dual 8+8 kernel only
#ifndef QD_NO_STATIC_SHAPES
        if (StaticSingle<8>::matches(p))  // single 8 bp index (BASELINE cfg2)
            return launch_fast_t<Rows8<BLOCK, false, U, StaticSingle<8>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
        return launch_fast_t<Rows8<BLOCK, false, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
    }
#ifndef QD_SWEEP_BUILD
    const int nl1 = p.seq_stride[0] > 8 ? 2 : 1, nl2 = p.seq_stride[1] > 8 ? 2 : 1;
    if (!dual) {
        if (nl1 == 1) return launch_fast_t<RowsX<BLOCK, 1, 1, false, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
        return launch_fast_t<RowsX<BLOCK, 2, 1, false, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    }
#ifndef QD_NO_STATIC_SHAPES
    if (StaticShape<8, 6>::matches(p))  // 8 bp barcode + 6 bp molecular index per index read (BASELINE cfg4)
        return launch_fast_t<RowsX<BLOCK, 2, 2, true, U, StaticShape<8, 6>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticShape<8, 8>::matches(p))  // 8 bp barcode + 8 bp molecular index per index read
        return launch_fast_t<RowsX<BLOCK, 2, 2, true, U, StaticShape<8, 8>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (StaticShape<6, 0>::matches(p))  // dual 6 bp indexes
        return launch_fast_t<RowsX<BLOCK, 1, 1, true, U, StaticShape<6, 0>>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#endif
    if (nl1 == 1 && nl2 == 1) return launch_fast_t<RowsX<BLOCK, 1, 1, true, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (nl1 == 1 && nl2 == 2) return launch_fast_t<RowsX<BLOCK, 1, 2, true, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    if (nl1 == 2 && nl2 == 1) return launch_fast_t<RowsX<BLOCK, 2, 1, true, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
    return launch_fast_t<RowsX<BLOCK, 2, 2, true, U>>(p, cache, cus, wg_per_cu, table_lds, lds_bytes, st);
#else
    return hipErrorInvalidValue;
#endif
}

}  // namespace

// Workgroup size: 512 threads; 1024 when the LDS image of the table is large (few workgroups fit a
// CU then, and bigger ones keep the wave count up); 256 for batches of at most 16 M pairs with a small
// table (a 4 M-pair batch is 3906 tiles of 512 threads: too few to fill 256 CUs evenly; measured -6 % at
// 4 M pairs, -1.4 % at 10 M, nothing at 100 M: profiles/r02_small_batch_block.txt) and for the wide form (its
// tile is 128 B per lane; -4 %).  block_override: 0 = this rule.
hipError_t qd_launch_fast(const DemuxParams& p, QdKernelCache& cache, int cus, int wg_per_cu, int block_override,
                          size_t lds_bytes, size_t strip_bytes_per_wave, hipStream_t st) {
    // (with the work queue a large image runs as two 512-thread workgroups per CU when both fit: on the 8 bp configs
    // 1024-thread workgroups measured 12 % slower than 512-thread ones in every grid form, profiles/r03_cfg3_launch_forms.txt)
    const bool big = lds_bytes > QD_FAST_BIG_LDS;
    const bool two_fit = 2 * (lds_bytes + strip_bytes_per_wave * 8) <= 158 * 1024;
    // (without the queue too: 512-thread workgroups, two per CU, whenever two images fit)
    int block = block_override ? block_override
                               : (big ? (two_fit ? 512 : 1024)
                                      : ((p.n <= QD_FAST_SMALL_BATCH || p.wide) ? 256 : QD_FAST_BLOCK));
    if (block == 1024) return launch_fast_b<1024>(p, cache, cus, wg_per_cu, lds_bytes, strip_bytes_per_wave, st);
    if (block == 256) return launch_fast_b<256>(p, cache, cus, wg_per_cu, lds_bytes, strip_bytes_per_wave, st);
    return launch_fast_b<512>(p, cache, cus, wg_per_cu, lds_bytes, strip_bytes_per_wave, st);
}

// out[0..ncnt) = base[0..ncnt) + the sum of the partial rows (out == base: the rows are added in place)
hipError_t qd_launch_reduce(const qd_row_t* partial, uint32_t rows, uint32_t cnt_stride, uint32_t ncnt,
                            const uint64_t* base, uint64_t* out, hipStream_t st) {
    const int b = 256;
    if (out != base) {
        hipError_t e = hipMemcpyAsync(out, base, (size_t)ncnt * 8, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(reduce_partials, dim3((ncnt + b - 1) / b, (rows + QD_REDUCE_ROWS - 1) / QD_REDUCE_ROWS), dim3(b), 0,
                       st, partial, rows, cnt_stride, ncnt, out);
    return hipGetLastError();
}
