// gfx950 (CDNA4 / MI355X) kernels of the demultiplexing hot path.
//
// Per read pair (reference: src/Quade.py:217-218,246-247 + src/Sample.py:56-91):
//   slice barcode window(s) out of the packed index-read rows, fuse, ASCII-fold, exact match
//   against the sample barcode table, min-phred gate over the barcode positions, slice + fuse the
//   molecular index, emit a uint16 routing code (+ molecular bytes), bump counters.
//
// This is byte/integer work bound by HBM bandwidth; there is no contraction in it, so no MFMA.
// Design (DESIGN.md has the numbers):
//   * demux_fast<>: rows of 8 or 16 bytes are read as 16-byte-per-lane vector loads
//     (global_load_dwordx4, lane i <-> consecutive 16 B: fully coalesced, 1 KiB per wave
//     instruction); all loads of a tile are issued before any is consumed.
//   * the barcode table (open-addressing slots + 16-byte canonical keys) and the per-sample
//     histogram live in LDS, staged once per workgroup; workgroups are persistent (grid = a few per
//     CU) and stride over tiles, so staging is amortised over >= 10^5 pairs.
//   * case fold and quality gate are SWAR on 64-bit registers (8 bases per operation);
//     the undetermined count is reduced across the wavefront with DPP/shuffle before one LDS add.
//   * counters: LDS histogram -> one global atomic per non-zero bin per workgroup into that
//     workgroup's own row of a partial-count matrix (no cross-workgroup contention);
//     rows are summed when the host asks for the counts.
//   * demux_generic: any power-of-two stride, optional per-read lengths (truncated index reads),
//     barcodes up to 32 bytes, table in global memory (L2 resident).  Correctness path.
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

// bytes [off, off+w) of a row (w <= 8) as a little-endian integer. off, w are wave-uniform.
template <int STRIDE>
__device__ __forceinline__ u64 window(u64 lo, u64 hi, int off, u64 mask) {
    u64 v;
    if (STRIDE == 8) {
        v = lo >> (8 * off);
    } else {
        if (off >= 8)
            v = hi >> (8 * (off - 8));
        else if (off == 0)
            v = lo;
        else
            v = (lo >> (8 * off)) | (hi << (64 - 8 * off));
    }
    return v & mask;
}

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
    const u64* bk;   // [S][2]
    uint32_t* hist;  // [2S + 1]
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
            if (t.bk[2 * id] == lo && t.bk[2 * id + 1] == hi) {
                found = id;
                break;
            }
        }
        s = (s + 1) & mask;
    }
    return found;
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

// One pair through the whole path.  Returns the routing code.
template <int SS1, int SS2, bool DUAL>
__device__ __forceinline__ uint32_t do_pair(const DemuxParams& p, const LdsTable& t, u64 s1lo, u64 s1hi,
                                            u64 q1, u64 s2lo, u64 s2hi, u64 q2, u64& mlo, u64& mhi) {
    // a1: barcode slice(s), fused (Quade.py:217 / :246)
    u64 k1 = window<SS1>(s1lo, s1hi, p.idx_off[0], p.idx_mask[0]);
    u64 klo = k1, khi = 0;
    if (DUAL) {
        u64 k2 = window<SS2>(s2lo, s2hi, p.idx_off[1], p.idx_mask[1]);
        fuse(k1, k2, p.idx_w[0], klo, khi);
    }
    // a2: molecular slice(s), fused, raw case (Quade.py:218 / :247)
    if (p.M > 0) {
        u64 m1 = window<SS1>(s1lo, s1hi, p.mol_off[0], p.mol_mask[0]);
        mlo = m1;
        mhi = 0;
        if (DUAL) {
            u64 m2 = window<SS2>(s2lo, s2hi, p.mol_off[1], p.mol_mask[1]);
            fuse(m1, m2, p.mol_w[0], mlo, mhi);
        }
    }
    // a3: fold for the lookup only (Sample.py:65)
    klo = qd_fold8(klo);
    khi = qd_fold8(khi);
    // a4: exact match (Sample.py:65-67)
#ifdef QD_ABLATE_PROBE  // timing-only build: no table probe (wrong results)
    const uint32_t id = (uint32_t)(klo ^ khi) & 63u;
#else
    const uint32_t id = probe_lds(t, klo, khi, (uint32_t)p.K, p.seed, p.slot_mask);
#endif
    if (id == QD_CODE_UNDET) return QD_CODE_UNDET;
    // a5: min-phred gate over the barcode positions (Sample.py:70)
    uint32_t pass = qd_all_ge8(q1, p.thr);
    if (DUAL) pass &= qd_all_ge8(q2, p.thr);
    const uint32_t code = id * 2u + (pass ^ 1u);
    // a6: per-sample counters (Sample.py:71-72,79-80)
#ifndef QD_ABLATE_HIST
    atomicAdd(&t.hist[code], 1u);
#endif
    return code;
}

// ------------------------------------------------------------------------------------------------
// Fast kernel.  Lane handles UNITS x 2 consecutive pairs per tile.  Row strides: seq SS1/SS2 in
// {8,16}, qual rows 8 bytes.  Rows are read once and never again: loads are non-temporal.
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

template <int SS1, int SS2, bool DUAL, int UNITS>
struct Tile {
    U128 s1[UNITS][SS1 / 8], s2[UNITS][SS2 / 8], q1[UNITS], q2[UNITS];
};

// issue every load of a tile (16 B per lane per instruction, coalesced); nothing is consumed here
// FULL: the tile lies entirely inside the batch -- every lane loads unconditionally, so the loads are
// plain straight-line code and the compiler can count them (s_waitcnt vmcnt(N)) when an older tile
// is consumed while these are in flight.  !FULL: the batch's last, partial tile, lane-guarded.
template <bool FULL, int BLOCK, int SS1, int SS2, bool DUAL, int UNITS>
__device__ __forceinline__ void load_tile(Tile<SS1, SS2, DUAL, UNITS>& T, const DemuxParams& p, int64_t base,
                                          uint32_t tid) {
    const int64_t n = p.n;
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;  // first pair of the unit
        if (FULL || p0 + 1 < n) {
#pragma unroll
            for (int j = 0; j < SS1 / 8; ++j) T.s1[u][j] = ld16s(p.seq[0] + p0 * SS1 + 16 * j);
            T.q1[u] = ld16s(p.qual[0] + p0 * 8);
            if (DUAL) {
#pragma unroll
                for (int j = 0; j < SS2 / 8; ++j) T.s2[u][j] = ld16s(p.seq[1] + p0 * SS2 + 16 * j);
                T.q2[u] = ld16s(p.qual[1] + p0 * 8);
            }
        } else if (p0 < n) {  // last, odd pair of the batch: never read past row n-1
#pragma unroll
            for (int j = 0; j < SS1 / 8; ++j) T.s1[u][j] = U128{0, 0};
            T.s1[u][0].lo = ld8(p.seq[0] + p0 * SS1);
            if (SS1 == 16) T.s1[u][0].hi = ld8(p.seq[0] + p0 * SS1 + 8);
            T.q1[u] = U128{ld8(p.qual[0] + p0 * 8), 0};
            if (DUAL) {
#pragma unroll
                for (int j = 0; j < SS2 / 8; ++j) T.s2[u][j] = U128{0, 0};
                T.s2[u][0].lo = ld8(p.seq[1] + p0 * SS2);
                if (SS2 == 16) T.s2[u][0].hi = ld8(p.seq[1] + p0 * SS2 + 8);
                T.q2[u] = U128{ld8(p.qual[1] + p0 * 8), 0};
            }
        }
    }
}

// TAG only makes the copies of this code distinct (an assembler comment), so that the compiler does
// not fold the copy that runs with younger loads in flight into the copy that runs without: a folded
// copy has to assume the younger loads are missing and waits for everything.
template <bool FULL, int TAG, int BLOCK, int SS1, int SS2, bool DUAL, int UNITS>
__device__ __forceinline__ uint32_t compute_tile(const Tile<SS1, SS2, DUAL, UNITS>& T, const DemuxParams& p,
                                                 const LdsTable& t, int64_t base, uint32_t tid) {
    asm volatile("; demux tile copy %0" ::"i"(TAG));
    const int64_t n = p.n;
    uint32_t undet = 0;
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int64_t p0 = base + ((int64_t)u * BLOCK + tid) * 2;
        if (!FULL && p0 >= n) continue;
        const bool two = FULL || (p0 + 1 < n);
        u64 a_lo, a_hi, b_lo, b_hi;                   // rows of pair p0 (a) and p0+1 (b), stream 1
        u64 c_lo = 0, c_hi = 0, d_lo = 0, d_hi = 0;  // stream 2
        if (SS1 == 8) {
            a_lo = T.s1[u][0].lo; a_hi = 0; b_lo = T.s1[u][0].hi; b_hi = 0;
        } else {
            a_lo = T.s1[u][0].lo; a_hi = T.s1[u][0].hi; b_lo = T.s1[u][SS1 / 8 - 1].lo; b_hi = T.s1[u][SS1 / 8 - 1].hi;
        }
        if (DUAL) {
            if (SS2 == 8) {
                c_lo = T.s2[u][0].lo; d_lo = T.s2[u][0].hi;
            } else {
                c_lo = T.s2[u][0].lo; c_hi = T.s2[u][0].hi; d_lo = T.s2[u][SS2 / 8 - 1].lo; d_hi = T.s2[u][SS2 / 8 - 1].hi;
            }
        }
        u64 m0lo = 0, m0hi = 0, m1lo = 0, m1hi = 0;
        const uint32_t c0 = do_pair<SS1, SS2, DUAL>(p, t, a_lo, a_hi, T.q1[u].lo, c_lo, c_hi,
                                                    DUAL ? T.q2[u].lo : 0, m0lo, m0hi);
        uint32_t c1 = 0;
        if (two)
            c1 = do_pair<SS1, SS2, DUAL>(p, t, b_lo, b_hi, T.q1[u].hi, d_lo, d_hi, DUAL ? T.q2[u].hi : 0, m1lo, m1hi);
        undet += (c0 == QD_CODE_UNDET) + (two && c1 == QD_CODE_UNDET);
        // a7: routing codes, 2 x uint16 per lane = one dword store, coalesced
        if (two)
            *reinterpret_cast<uint32_t*>(p.codes + p0) = c0 | (c1 << 16);
        else
            p.codes[p0] = (uint16_t)c0;
        if (p.M > 0) {
            if (two)
                store_mol2(p.mol + p0 * p.M, m0lo, m0hi, m1lo, m1hi, p.M);
            else
                store_mol(p.mol + p0 * p.M, m0lo, m0hi, p.M);
        }
    }
    return undet;
}

template <int BLOCK, int SS1, int SS2, bool DUAL, int UNITS>
__global__ __launch_bounds__(BLOCK) void demux_fast(const DemuxParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t* slots = reinterpret_cast<uint32_t*>(lds_raw);
    u64* bk = reinterpret_cast<u64*>(lds_raw + p.lds_bk_off);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_raw + p.lds_hist_off);
    const uint32_t tid = threadIdx.x;
    const uint32_t S = p.n_samples;
    constexpr int64_t TILE = (int64_t)BLOCK * 2 * UNITS;  // pairs per workgroup iteration
    typedef Tile<SS1, SS2, DUAL, UNITS> TileT;

    // Full tiles [0, nfull) are strided over the grid; the partial last tile (if any) is done by one
    // workgroup after its full tiles.  The first tile's rows are requested before anything else, so
    // HBM latency overlaps the staging of the table.
    const int64_t nfull = p.n / TILE;
    int64_t tile = blockIdx.x;
    TileT A, B;
    if (tile < nfull) load_tile<true, BLOCK>(A, p, tile * TILE, tid);

    // stage the table: global (L2) -> LDS, once per workgroup
    for (uint32_t i = tid; i <= p.slot_mask; i += BLOCK) slots[i] = p.slots[i];
    for (uint32_t i = tid; i < 2 * S; i += BLOCK) bk[i] = p.bk16[i];
    for (uint32_t i = tid; i < 2 * S + 1; i += BLOCK) hist[i] = 0;
    __syncthreads();
    const LdsTable t{slots, bk, hist};

    uint32_t undet = 0;
    // Register double buffering (tile k+1 in flight while tile k is matched) for the 8-byte-row
    // instantiations; 16-byte rows already hold 96 B per lane per tile and run single-buffered.
    // Every "load next, then match current" pair is straight-line code with its own copy of the match
    // (no control-flow join between issuing the younger loads and consuming the older ones): with a
    // join the compiler must assume the younger loads may be missing and waits for them too
    // (s_waitcnt vmcnt(3..0) instead of vmcnt(4+)), which serialises the two tiles.
    constexpr bool PREFETCH = QD_FAST_PREFETCH && (SS1 + (DUAL ? SS2 : 0) <= QD_FAST_PREFETCH_MAXROW);
    const int64_t G = gridDim.x;
    if (tile < nfull) {
        if (PREFETCH) {
            for (;;) {
                const int64_t next = tile + G;
                if (next >= nfull) {
                    undet += compute_tile<true, 0, BLOCK>(A, p, t, tile * TILE, tid);
                    break;
                }
                load_tile<true, BLOCK>(B, p, next * TILE, tid);
                undet += compute_tile<true, 1, BLOCK>(A, p, t, tile * TILE, tid);
                tile = next + G;
                if (tile >= nfull) {
                    undet += compute_tile<true, 2, BLOCK>(B, p, t, next * TILE, tid);
                    break;
                }
                load_tile<true, BLOCK>(A, p, tile * TILE, tid);
                undet += compute_tile<true, 3, BLOCK>(B, p, t, next * TILE, tid);
            }
        } else {
            for (;;) {
                undet += compute_tile<true, 4, BLOCK>(A, p, t, tile * TILE, tid);
                tile += G;
                if (tile >= nfull) break;
                load_tile<true, BLOCK>(A, p, tile * TILE, tid);
            }
        }
    }
    if (nfull * TILE < p.n && (int64_t)blockIdx.x == nfull % G) {  // the partial tile, lane-guarded
        load_tile<false, BLOCK>(A, p, nfull * TILE, tid);
        undet += compute_tile<false, 5, BLOCK>(A, p, t, nfull * TILE, tid);
    }

    // undetermined count: wavefront shuffle-reduce (64 lanes), then one LDS add per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) undet += __shfl_xor(undet, o, 64);
    if ((tid & 63) == 0 && undet) atomicAdd(&hist[2 * S], undet);
    __syncthreads();
    // flush this workgroup's histogram into its own row of the partial-count matrix
    u64* row = p.partial + (size_t)(blockIdx.x % p.partial_rows) * p.cnt_stride;
#ifndef QD_ABLATE_FLUSH
    for (uint32_t i = tid; i < 2 * S + 1; i += BLOCK) {
        const uint32_t v = hist[i];
        if (v) atomicAdd(reinterpret_cast<unsigned long long*>(&row[i]), (unsigned long long)v);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: one pair per lane, byte-granular, per-read lengths honoured (Python slice
// clamping of a short index read: src/Quade.py:217-218 on a read shorter than `end`).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(QD_GEN_BLOCK) void demux_generic(const DemuxParams p) {
    const int64_t stride = (int64_t)gridDim.x * QD_GEN_BLOCK;
    const uint32_t S = p.n_samples;
    u64* row = p.partial + (size_t)(blockIdx.x % p.partial_rows) * p.cnt_stride;
    uint32_t undet = 0;
    for (int64_t r = (int64_t)blockIdx.x * QD_GEN_BLOCK + threadIdx.x; r < p.n; r += stride) {
        // slice lengths after clamping to the read length
        int a[2] = {0, 0}, ma[2] = {0, 0};
        const uint8_t* srow[2] = {nullptr, nullptr};
        const uint8_t* qrow[2] = {nullptr, nullptr};
        for (int k = 0; k < p.n_streams; ++k) {
            const int len = p.len[k] ? (int)p.len[k][r] : 0x7FFFFFFF;
            // columns [start, min(end, len)) -> bytes available
            a[k] = clampi((p.idx_col[k] + p.idx_w[k] < len ? p.idx_col[k] + p.idx_w[k] : len) - p.idx_col[k], 0, p.idx_w[k]);
            ma[k] = clampi((p.mol_col[k] + p.mol_w[k] < len ? p.mol_col[k] + p.mol_w[k] : len) - p.mol_col[k], 0, p.mol_w[k]);
            srow[k] = p.seq[k] + r * p.seq_stride[k];
            qrow[k] = p.qual[k] + r * p.qual_stride[k];
        }
        const int klen = a[0] + a[1];
        // canonical key: fused bytes, little-endian packed, zero padded
        u64 w[QD_KEY_WORDS] = {0, 0, 0, 0};
        uint32_t pass = 1;
#pragma unroll
        for (int i = 0; i < QD_MAX_KEY_BYTES; ++i) {
            if (i < klen) {
                uint8_t b, q;
                if (i < a[0]) {
                    b = srow[0][p.idx_off[0] + i];
                    q = qrow[0][i];
                } else {
                    b = srow[1][p.idx_off[1] + (i - a[0])];
                    q = qrow[1][i - a[0]];
                }
                if (b >= 'a' && b <= 'z') b -= 0x20;      // a3
                w[i >> 3] |= (u64)b << (8 * (i & 7));
                pass &= (q >= p.thr) ? 1u : 0u;            // a5
            }
        }
        // a4: probe the global table
        uint32_t code = QD_CODE_UNDET;
        if (klen <= QD_MAX_KEY_BYTES) {
            const uint32_t h = qd_hash_key(w, (uint32_t)klen, p.seed);
            const uint32_t fp = h >> 16;
            uint32_t s = h & p.slot_mask;
            for (;;) {
                const uint32_t e = p.slots[s];
                if (e == QD_EMPTY_SLOT) break;
                if ((e >> 16) == fp) {
                    const uint32_t id = e & 0xFFFFu;
                    const u64* b = p.bk32 + (size_t)id * QD_KEY_WORDS;
                    if (p.blen[id] == (uint8_t)klen && b[0] == w[0] && b[1] == w[1] && b[2] == w[2] && b[3] == w[3]) {
                        code = id * 2u + (pass ^ 1u);
                        break;
                    }
                }
                s = (s + 1) & p.slot_mask;
            }
        }
        p.codes[r] = (uint16_t)code;
        if (code == QD_CODE_UNDET)
            ++undet;
        else
            atomicAdd(reinterpret_cast<unsigned long long*>(&row[code]), 1ull);
        // a2: molecular bytes, I1 part then I2 part, zero padded to M
        if (p.M > 0) {
            uint8_t* d = p.mol + r * p.M;
            int o = 0;
            for (int k = 0; k < p.n_streams; ++k)
                for (int i = 0; i < ma[k]; ++i) d[o++] = srow[k][p.mol_off[k] + i];
            for (; o < p.M; ++o) d[o] = 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) undet += __shfl_xor(undet, o, 64);
    if ((threadIdx.x & 63) == 0 && undet) atomicAdd(reinterpret_cast<unsigned long long*>(&row[2 * S]), (unsigned long long)undet);
}

// sum the partial rows -> out[ncnt] (out zeroed by the caller).  blockIdx.y = a group of
// QD_REDUCE_ROWS rows, thread = one counter: row reads are coalesced across the threads and
// independent across the rows; one 64-bit atomic per (row group, counter).
#define QD_REDUCE_ROWS 32
__global__ void reduce_partials(const u64* partial, uint32_t rows, uint32_t cnt_stride, uint32_t ncnt, u64* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncnt) return;
    const uint32_t r0 = blockIdx.y * QD_REDUCE_ROWS;
    const uint32_t r1 = r0 + QD_REDUCE_ROWS < rows ? r0 + QD_REDUCE_ROWS : rows;
    u64 s = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; ++r) s += partial[(size_t)r * cnt_stride + i];
    if (s) atomicAdd(reinterpret_cast<unsigned long long*>(&out[i]), (unsigned long long)s);
}

template <int BLOCK, int SS1, int SS2, bool DUAL>
hipError_t launch_fast_t(const DemuxParams& p, int cus, int wg_per_cu, size_t lds, hipStream_t st) {
    auto k = demux_fast<BLOCK, SS1, SS2, DUAL, QD_FAST_UNITS>;
    static bool attr_set = false;  // per instantiation
    static size_t occ_lds = ~(size_t)0;
    static int occ_blocks = 1;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (occ_lds != lds) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLOCK, lds);
        if (e != hipSuccess) return e;
        occ_blocks = nb < 1 ? 1 : nb;
        occ_lds = lds;
    }
    const int64_t tile = (int64_t)BLOCK * 2 * QD_FAST_UNITS;
    const int64_t ntiles = (p.n + tile - 1) / tile;
    // Grid (automatic), from the measurements in profiles/r01_tune*_*.txt:
    //  * small table image (<= 24 KB of LDS): oversubscribe -- up to 64 workgroups per CU, at least 8
    //    tiles each; surplus workgroups start as earlier ones retire, which keeps the streams'
    //    active window compact and evens out the tail; re-staging a few KB per workgroup is free;
    //  * large table image: a persistent grid of at most 2 co-resident workgroups per CU (staging
    //    tens of KB and flushing thousands of counters per workgroup is not free).
    int64_t grid;
    if (wg_per_cu > 0) {
        grid = (int64_t)cus * wg_per_cu;
    } else if (lds > 24 * 1024) {
        grid = (int64_t)cus * (occ_blocks < 2 ? occ_blocks : 2);
    } else {
        grid = ntiles / 8;
        const int64_t lo = (int64_t)cus * (occ_blocks < 2 ? occ_blocks : 2), hi = (int64_t)cus * 64;
        if (grid < lo) grid = lo;
        if (grid > hi) grid = hi;
    }
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(BLOCK), lds, st, p);
    return hipGetLastError();
}

template <int BLOCK>
hipError_t launch_fast_b(const DemuxParams& p, int cus, int wg_per_cu, size_t lds_bytes, hipStream_t st) {
    const int ss1 = p.seq_stride[0], ss2 = p.seq_stride[1];
#ifdef QD_SWEEP_BUILD  // tuning builds instantiate the dual 8+8 kernel only
    return launch_fast_t<BLOCK, 8, 8, true>(p, cus, wg_per_cu, lds_bytes, st);
#endif
    if (p.n_streams == 1) {
        if (ss1 == 8) return launch_fast_t<BLOCK, 8, 8, false>(p, cus, wg_per_cu, lds_bytes, st);
        return launch_fast_t<BLOCK, 16, 8, false>(p, cus, wg_per_cu, lds_bytes, st);
    }
    if (ss1 == 8 && ss2 == 8) return launch_fast_t<BLOCK, 8, 8, true>(p, cus, wg_per_cu, lds_bytes, st);
    if (ss1 == 8 && ss2 == 16) return launch_fast_t<BLOCK, 8, 16, true>(p, cus, wg_per_cu, lds_bytes, st);
    if (ss1 == 16 && ss2 == 8) return launch_fast_t<BLOCK, 16, 8, true>(p, cus, wg_per_cu, lds_bytes, st);
    return launch_fast_t<BLOCK, 16, 16, true>(p, cus, wg_per_cu, lds_bytes, st);
}

}  // namespace

// Workgroup size: 512 threads; 1024 when the LDS image of the table is large (few workgroups fit a
// CU then, and bigger ones keep the wave count up).  block_override: 0 = this rule.
hipError_t qd_launch_fast(const DemuxParams& p, int cus, int wg_per_cu, int block_override, size_t lds_bytes,
                          hipStream_t st) {
    int block = block_override ? block_override : (lds_bytes > QD_FAST_BIG_LDS ? 1024 : QD_FAST_BLOCK);
    if (block == 1024) return launch_fast_b<1024>(p, cus, wg_per_cu, lds_bytes, st);
    if (block == 256) return launch_fast_b<256>(p, cus, wg_per_cu, lds_bytes, st);
    return launch_fast_b<512>(p, cus, wg_per_cu, lds_bytes, st);
}

hipError_t qd_launch_generic(const DemuxParams& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL(demux_generic, dim3(grid), dim3(QD_GEN_BLOCK), 0, st, p);
    return hipGetLastError();
}

hipError_t qd_launch_reduce(const uint64_t* partial, uint32_t rows, uint32_t cnt_stride,
                            uint32_t ncnt, uint64_t* out, hipStream_t st) {
    const int b = 256;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)ncnt * 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(reduce_partials, dim3((ncnt + b - 1) / b, (rows + QD_REDUCE_ROWS - 1) / QD_REDUCE_ROWS), dim3(b), 0,
                       st, partial, rows, cnt_stride, ncnt, out);
    return hipGetLastError();
}
