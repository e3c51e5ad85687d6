// gfx950 (CDNA4 / MI355X): the generic path of the demultiplexing hot path -- any plan inside the envelope, any read length.
//
// One pair per lane, byte-granular slices, per-read lengths honoured (Python slice clamping of a short index read:
// src/Quade.py:217-218 on a read shorter than `end`), barcodes up to 32 bytes, the table of EVERY barcode in global memory (L2)
// or, for the specialised forms, in LDS.  It is the correctness path: plans the fast kernels (quade_kernels.hip) do not take
// (windows > 16 B per index read, K > 16 outside the wide form, tables beyond the LDS budget), batches of mostly short reads,
// and the listed short reads of a batch redone after a fast launch (demux_fixup).  DESIGN.md 4.2 has the numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quade_common.h"
#include "quade_kernels.h"

namespace {

typedef uint64_t u64;

// ------------------------------------------------------------------------------------------------
// Generic path: one pair per lane, byte-granular, per-read lengths honoured (Python slice clamping
// of a short index read: src/Quade.py:217-218 on a read shorter than `end`).  generic_pair() is the
// whole of it for one pair; it serves the generic kernel (any plan inside the envelope) and the
// exception pairs (short reads) redone after a fast launch.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// nbytes (0..32, per lane) bytes at p (any alignment) -> little-endian words w[0..3], zero padded.  wmax = the
// wave-uniform upper bound of nbytes (the slice width of the plan; nbytes is smaller only for a short read).
// Aligned dword loads, and NO per-lane branches around them: the number of loads follows from wmax alone
// (a scalar condition), a lane that needs fewer words re-reads its last needed word instead of skipping --
// with a guard per load every load sat in its own exec-masked block and their latencies added up (~10 us
// per pair and lane).  No word is read that holds no byte of the lane's slice, so the end of an array is
// passed by at most the 1-3 bytes that share a word (and a page) with its last byte.
template <int NW>  // NW 64-bit words of output: nbytes, wmax <= 8 * NW
__device__ __forceinline__ void load_bytes(const uint8_t* p, int nbytes, int wmax, u64 (&w)[NW]) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const int sh = (int)(a & 3), need = sh + nbytes;
    const int ju = wmax > 0 ? (wmax + 6) >> 2 : 0;    // dwords that can hold 3 + wmax bytes (uniform)
    const int jl = need > 0 ? (need - 1) >> 2 : 0;    // the lane's last needed dword
    uint32_t d[2 * NW + 1];
#pragma unroll
    for (int j = 0; j < 2 * NW + 1; ++j) d[j] = (j < ju) ? q[j < jl ? j : jl] : 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const u64 lo = (u64)d[2 * i] | ((u64)d[2 * i + 1] << 32), nx = d[2 * i + 2];
        const u64 v = sh ? (lo >> (8 * sh)) | (nx << (64 - 8 * sh)) : lo;
        const int left = nbytes - 8 * i;  // wanted bytes of this word
        w[i] = left >= 8 ? v : (left <= 0 ? 0 : v & ((1ull << (8 * left)) - 1));
    }
}

// w |= v << (8 * off) over the 64 * NW bits (off = 0 .. 8 * NW bytes; what leaves the top is dropped)
template <int NW>
__device__ __forceinline__ void or_shifted(u64 (&w)[NW], const u64 (&v)[NW], int off) {
    const int ws = off >> 3, bs = (off & 7) * 8;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        u64 cur = 0, prev = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            cur = (j == i - ws) ? v[j] : cur;
            prev = (j == i - ws - 1) ? v[j] : prev;
        }
        w[i] |= bs ? (cur << bs) | (prev >> (64 - bs)) : cur;
    }
}

// Writes codes[r] (and mol[r]); returns the routing code.  Counters are the caller's business.
// len0 / len1: the reads' lengths (0x7FFFFFFF = covers its window).
__device__ __forceinline__ uint32_t generic_pair(const DemuxParams& p, int64_t r, int len0, int len1) {
    // slice lengths after clamping to the read length
    int a[2] = {0, 0}, ma[2] = {0, 0};
    const uint8_t* srow[2] = {nullptr, nullptr};
    const uint8_t* qrow[2] = {nullptr, nullptr};
    for (int k = 0; k < p.n_streams; ++k) {
        const int len = k ? len1 : len0;
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
    // slices as words: aligned dword loads + byte shifts, SWAR fold (a3) and gate (a5) as in the fast kernels
    {
        int at = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= p.n_streams) break;
            u64 v[4], qv[4];
            load_bytes<4>(srow[k] + p.idx_off[k], a[k], p.idx_w[k], v);
            load_bytes<4>(qrow[k], a[k], p.idx_w[k], qv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = qd_fold8(v[i]);
                const int left = a[k] - 8 * i;  // quality bytes of this word that count; the others read as 0xFF
                pass &= qd_all_ge8(left >= 8 ? qv[i] : (left <= 0 ? ~0ull : qv[i] | (~0ull << (8 * left))), p.thr);
            }
            or_shifted<4>(w, v, at);
            at += a[k];
        }
    }
    // a4: probe the global table of every barcode
    uint32_t code = QD_CODE_UNDET;
    if (klen <= QD_MAX_KEY_BYTES) {
        const uint32_t h = qd_hash_key(w, (uint32_t)klen, p.gseed);
        const uint32_t fp = h >> 16;
        uint32_t s = h & p.gmask;
        for (;;) {
            const uint32_t e = p.gslots[s];
            if (e == QD_EMPTY_SLOT) break;
            if ((e >> 16) == fp) {
                const uint32_t id = e & 0xFFFFu;
                const u64* b = p.bk32 + (size_t)id * QD_KEY_WORDS;
                if (p.blen[id] == (uint8_t)klen && b[0] == w[0] && b[1] == w[1] && b[2] == w[2] && b[3] == w[3]) {
                    code = id * 2u + (pass ^ 1u);
                    break;
                }
            }
            s = (s + 1) & p.gmask;
        }
    }
    p.codes[r] = (uint16_t)code;
    // a2: molecular bytes, I1 part then I2 part, zero padded to M
    if (p.M > 0) {
        uint8_t* d = p.mol + r * p.M;
        if (p.M > 32) {  // molecular slices may be as wide as the window (64 B each): bytes
            int o = 0;
            for (int k = 0; k < p.n_streams; ++k)
                for (int i = 0; i < ma[k]; ++i) d[o++] = srow[k][p.mol_off[k] + i];
            for (; o < p.M; ++o) d[o] = 0;
            return code;
        }
        u64 m[4] = {0, 0, 0, 0};
        int at = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= p.n_streams) break;
            u64 v[4];
            load_bytes<4>(srow[k] + p.mol_off[k], ma[k], p.mol_w[k], v);
            or_shifted<4>(m, v, at);
            at += ma[k];
        }
        if ((p.M & 3) == 0) {  // r * M is a multiple of 4 then: dword stores
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (4 * j < p.M) reinterpret_cast<uint32_t*>(d)[j] = (uint32_t)(m[j >> 1] >> (32 * (j & 1)));
        } else {
#pragma unroll
            for (int o = 0; o < 32; ++o)
                if (o < p.M) d[o] = (uint8_t)(m[o >> 3] >> (8 * (o & 7)));
        }
    }
    return code;
}

// lengths from the per-pair len rows (absent: every read covers its window)
__device__ __forceinline__ uint32_t generic_pair(const DemuxParams& p, int64_t r) {
    return generic_pair(p, r, p.len[0] ? (int)p.len[0][r] : 0x7FFFFFFF, p.len[1] ? (int)p.len[1][r] : 0x7FFFFFFF);
}

// ---- the generic path specialised by what a plan fixes for a whole launch --------------------------------------
// Every read covers its window (no len rows), NS index reads, keys of at most 8 * KW bytes, molecular bytes of at
// most 8 * MWORDS (0: none).  Same steps as generic_pair(); what changes is what the compiler knows: the word
// arrays have the size the plan needs (a 16-byte key is two words, not four), the slice widths are wave-uniform (no
// per-lane clamping), and the stream loop has a constant bound -- the catch-all keeps ~160 scalars alive and spills
// them through VGPR lanes (158 SGPR spills, 942 VALU per pair).
// A slice whose address is a multiple of ALIGN (4 or 8) for EVERY pair of the launch -- every stride and offset of the
// plan is: no shift, no spare word, and 8-byte loads where the plan allows them (the one-pair-per-lane kernels are bound
// by the number of vector memory instructions: three dword loads per 8-byte slice, each using half of the lines it
// touches, against one).  nbytes (wave-uniform) <= 8 * NW; bytes beyond it read as zero.
template <int NW, int ALIGN>
__device__ __forceinline__ void load_aligned(const uint8_t* p, int nbytes, u64 (&w)[NW]) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int left = nbytes - 8 * i;  // wanted bytes of this word
        u64 v = 0;
        if (left > 0) {
            if (ALIGN >= 8) {
                v = *reinterpret_cast<const u64*>(p + 8 * i);  // (may read up to 7 bytes behind the slice, inside its aligned word)
            } else {
                v = *reinterpret_cast<const uint32_t*>(p + 8 * i);
                if (left > 4) v |= (u64) * reinterpret_cast<const uint32_t*>(p + 8 * i + 4) << 32;
            }
            if (left < 8) v &= (1ull << (8 * left)) - 1;
        }
        w[i] = v;
    }
}
template <int NW, int ALIGN>
__device__ __forceinline__ void load_slice(const uint8_t* p, int nbytes, u64 (&w)[NW]) {
    if (ALIGN >= 4)
        load_aligned<NW, ALIGN>(p, nbytes, w);
    else
        load_bytes<NW>(p, nbytes, nbytes, w);
}

// tslots / tkeys / tlens: the table of every barcode -- in global memory (keys QD_KEY_WORDS words apart) or, LT, the workgroup's
// copy in LDS (keys KW words apart)
template <int NS, int KW, int MWORDS, int ALIGN, bool LT>
__device__ __forceinline__ uint32_t special_pair(const DemuxParams& p, int64_t r, const uint32_t* tslots, const u64* tkeys,
                                                 const uint8_t* tlens) {
    u64 w[KW];
#pragma unroll
    for (int i = 0; i < KW; ++i) w[i] = 0;
    uint32_t pass = 1;
    int at = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int iw = p.idx_w[k];
        u64 v[KW], qv[KW];
        load_slice<KW, ALIGN>(p.seq[k] + r * p.seq_stride[k] + p.idx_off[k], iw, v);
        load_slice<KW, ALIGN>(p.qual[k] + r * p.qual_stride[k], iw, qv);
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            v[i] = qd_fold8(v[i]);                                                    // a3
            const int left = iw - 8 * i;  // quality bytes of this word that count; the others read as 0xFF
            pass &= qd_all_ge8(left >= 8 ? qv[i] : (left <= 0 ? ~0ull : qv[i] | (~0ull << (8 * left))), p.thr);  // a5
        }
        or_shifted<KW>(w, v, at);                                                     // a1
        at += iw;
    }
    // a4: the table of every barcode (a K-long key can only equal a K-long barcode)
    uint32_t code = QD_CODE_UNDET;
    {
        uint32_t h = qd_hash_init((uint32_t)at, p.gseed);
#pragma unroll
        for (int i = 0; i < KW; ++i)
            if (8 * i < at) h = qd_hash_step(h, w[i]);
        h = qd_hash_fini(h);
        const uint32_t fp = h >> 16;
        uint32_t s = h & p.gmask;
        for (;;) {
            const uint32_t e = tslots[s];
            if (e == QD_EMPTY_SLOT) break;
            if ((e >> 16) == fp) {
                const uint32_t id = e & 0xFFFFu;
                const u64* b = tkeys + (size_t)id * (LT ? KW : QD_KEY_WORDS);
                bool same = tlens[id] == (uint8_t)at;
#pragma unroll
                for (int i = 0; i < KW; ++i) same = same && b[i] == w[i];
                if (same) {
                    code = id * 2u + (pass ^ 1u);
                    break;
                }
            }
            s = (s + 1) & p.gmask;
        }
    }
    p.codes[r] = (uint16_t)code;
    if (MWORDS > 0) {  // a2: molecular bytes, I1 part then I2 part
        constexpr int MW_ = MWORDS > 0 ? MWORDS : 1;
        u64 m[MW_];
#pragma unroll
        for (int i = 0; i < MW_; ++i) m[i] = 0;
        int mat = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            u64 v[MW_];
            load_slice<MW_, ALIGN>(p.seq[k] + r * p.seq_stride[k] + p.mol_off[k], p.mol_w[k], v);
            or_shifted<MW_>(m, v, mat);
            mat += p.mol_w[k];
        }
        uint8_t* d = p.mol + r * p.M;
        if ((p.M & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 2 * MW_; ++j)
                if (4 * j < p.M) reinterpret_cast<uint32_t*>(d)[j] = (uint32_t)(m[j >> 1] >> (32 * (j & 1)));
        } else {
#pragma unroll
            for (int o = 0; o < 8 * MW_; ++o)
                if (o < p.M) d[o] = (uint8_t)(m[o >> 3] >> (8 * (o & 7)));
        }
    }
    return code;
}

// LT: the workgroup stages the table of every barcode in LDS behind its histogram (slots | keys, KW words each | lengths) and
// probes it there: the two dependent trips to global memory per pair (slot, then key) were most of what the one-pair-per-lane
// kernels waited for; the launcher picks it while the copy is small (a few workgroups per CU must still fit).
template <int NS, int KW, int MWORDS, int ALIGN, bool LT>
__global__ __launch_bounds__(QD_GEN_BLOCK) void demux_special(const DemuxParams p, uint32_t hist_entries) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_raw);
    const int64_t stride = (int64_t)gridDim.x * QD_GEN_BLOCK;
    const uint32_t S = p.n_samples;
    qd_row_t* row = p.partial + (size_t)(blockIdx.x % p.partial_rows) * p.cnt_stride;
    for (uint32_t i = threadIdx.x; i < hist_entries; i += QD_GEN_BLOCK) hist[i] = 0;
    const uint32_t* tslots = p.gslots;
    const u64* tkeys = p.bk32;
    const uint8_t* tlens = p.blen;
    if (LT) {
        uint32_t* ls = reinterpret_cast<uint32_t*>(lds_raw + (((size_t)hist_entries * 4 + 15) & ~(size_t)15));
        u64* lk = reinterpret_cast<u64*>(ls + (p.gmask + 1));
        uint8_t* ll = reinterpret_cast<uint8_t*>(lk + (size_t)S * KW);
        for (uint32_t i = threadIdx.x; i <= p.gmask; i += QD_GEN_BLOCK) ls[i] = p.gslots[i];
        for (uint32_t i = threadIdx.x; i < S * KW; i += QD_GEN_BLOCK) lk[i] = p.bk32[(size_t)(i / KW) * QD_KEY_WORDS + (i % KW)];
        for (uint32_t i = threadIdx.x; i < S; i += QD_GEN_BLOCK) ll[i] = p.blen[i];
        tslots = ls;
        tkeys = lk;
        tlens = ll;
    }
    if (hist_entries || LT) __syncthreads();
    uint32_t undet = 0;
    for (int64_t r = (int64_t)blockIdx.x * QD_GEN_BLOCK + threadIdx.x; r < p.n; r += stride) {
        const uint32_t code = special_pair<NS, KW, MWORDS, ALIGN, LT>(p, r, tslots, tkeys, tlens);
        if (code == QD_CODE_UNDET)
            ++undet;
        else if (hist_entries)
            atomicAdd(&hist[code], 1u);
        else
            atomicAdd(&row[code], (qd_row_t)1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) undet += __shfl_xor(undet, o, 64);
    if ((threadIdx.x & 63) == 0 && undet) atomicAdd(&row[2 * S], (qd_row_t)undet);
    if (hist_entries) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < hist_entries; i += QD_GEN_BLOCK) {
            const uint32_t v = hist[i];
            if (v) atomicAdd(&row[i], (qd_row_t)v);
        }
    }
}

// hist_entries = 2S+1 when the per-sample counters fit the workgroup's LDS (dynamic, 4 B each): one LDS
// add per matched pair and one global add per non-zero counter per workgroup; 0 for sample tables too
// large for that (global 64-bit adds per pair, as before).
__global__ __launch_bounds__(QD_GEN_BLOCK) void demux_generic(const DemuxParams p, uint32_t hist_entries) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds_raw);
    const int64_t stride = (int64_t)gridDim.x * QD_GEN_BLOCK;
    const uint32_t S = p.n_samples;
    qd_row_t* row = p.partial + (size_t)(blockIdx.x % p.partial_rows) * p.cnt_stride;
    for (uint32_t i = threadIdx.x; i < hist_entries; i += QD_GEN_BLOCK) hist[i] = 0;
    if (hist_entries) __syncthreads();
    uint32_t undet = 0;
    for (int64_t r = (int64_t)blockIdx.x * QD_GEN_BLOCK + threadIdx.x; r < p.n; r += stride) {
        const uint32_t code = generic_pair(p, r);
        if (code == QD_CODE_UNDET)
            ++undet;
        else if (hist_entries)
            atomicAdd(&hist[code], 1u);
        else
            atomicAdd(&row[code], (qd_row_t)1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) undet += __shfl_xor(undet, o, 64);
    if ((threadIdx.x & 63) == 0 && undet) atomicAdd(&row[2 * S], (qd_row_t)undet);
    if (hist_entries) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < hist_entries; i += QD_GEN_BLOCK) {
            const uint32_t v = hist[i];
            if (v) atomicAdd(&row[i], (qd_row_t)v);
        }
    }
}

// Exception pairs after a fast launch on the same stream: the reads of p.exc[0..n_exc) (unique pair
// indices) are shorter than their window, so the fast kernel matched their zero-padded rows; redo
// them with the generic semantics and move one count from the old code's counter to the new one.
__global__ __launch_bounds__(QD_GEN_BLOCK) void demux_fixup(const DemuxParams p) {
    const uint32_t S = p.n_samples;
    u64* row = p.adjust;  // signed moves go to the 64-bit totals (a -1 in a 32-bit row of its own would not cancel)
    for (uint32_t i = blockIdx.x * QD_GEN_BLOCK + threadIdx.x; i < p.n_exc; i += gridDim.x * QD_GEN_BLOCK) {
        const int64_t r = p.exc[i];
        if (r >= p.n) continue;
        const uint32_t old = p.codes[r];
        const uint32_t code = p.exc_len[0] ? generic_pair(p, r, p.exc_len[0][i], p.exc_len[1] ? (int)p.exc_len[1][i] : 0x7FFFFFFF)
                                           : generic_pair(p, r);
        if (code != old) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&row[old == QD_CODE_UNDET ? 2 * S : old]), ~0ull);  // -1
            atomicAdd(reinterpret_cast<unsigned long long*>(&row[code == QD_CODE_UNDET ? 2 * S : code]), 1ull);
        }
    }
}

}  // namespace

namespace {
template <int NS, int KW, int ALIGN, bool LT>
void launch_special_t(const DemuxParams& p, int grid, size_t lds, uint32_t entries, hipStream_t st) {
    if (p.M == 0) hipLaunchKernelGGL((demux_special<NS, KW, 0, ALIGN, LT>), dim3(grid), dim3(QD_GEN_BLOCK), lds, st, p, entries);
    else if (p.M <= 16) hipLaunchKernelGGL((demux_special<NS, KW, 2, ALIGN, LT>), dim3(grid), dim3(QD_GEN_BLOCK), lds, st, p, entries);
    else hipLaunchKernelGGL((demux_special<NS, KW, 4, ALIGN, LT>), dim3(grid), dim3(QD_GEN_BLOCK), lds, st, p, entries);
}
template <int NS, int KW, int ALIGN>
void launch_special_a(const DemuxParams& p, int grid, size_t lds, uint32_t entries, hipStream_t st) {
    // the table in LDS while histogram + slots + keys + lengths stay within QD_GENERIC_LDS_TABLE bytes (needs the histogram there too)
    const size_t tbl = (((size_t)entries * 4 + 15) & ~(size_t)15) + ((size_t)p.gmask + 1) * 4 + (size_t)p.n_samples * KW * 8 + p.n_samples;
    if (QD_GENERIC_LDS_TABLE && entries && tbl <= (size_t)QD_GENERIC_LDS_TABLE) launch_special_t<NS, KW, ALIGN, true>(p, grid, (tbl + 15) & ~(size_t)15, entries, st);
    else launch_special_t<NS, KW, ALIGN, false>(p, grid, lds, entries, st);
}
// what every slice address of the launch is a multiple of: 8, 4 or nothing in particular (row arrays are 16-byte aligned)
int slice_alignment(const DemuxParams& p) {
    int a = 8;
    for (int k = 0; k < p.n_streams; ++k) {
        const int v[5] = {p.seq_stride[k], p.qual_stride[k], p.idx_w[k] ? p.idx_off[k] : 0, p.mol_w[k] ? p.mol_off[k] : 0, 0};
        for (int x : v)
            while (a > 1 && x % a) a >>= 1;
    }
    return a >= 4 ? a : 1;
}
template <int NS, int KW>
void launch_special_m(const DemuxParams& p, int grid, size_t lds, uint32_t entries, hipStream_t st) {
    const int a = QD_GENERIC_ALIGNED ? slice_alignment(p) : 1;
    if (a == 8) launch_special_a<NS, KW, 8>(p, grid, lds, entries, st);
    else if (a == 4) launch_special_a<NS, KW, 4>(p, grid, lds, entries, st);
    else launch_special_a<NS, KW, 1>(p, grid, lds, entries, st);
}
}  // namespace

hipError_t qd_launch_generic(const DemuxParams& p, int grid, hipStream_t st) {
    const uint32_t entries = 2 * p.n_samples + 1 <= 16000 ? 2 * p.n_samples + 1 : 0;  // <= 64 KB of LDS (the default limit)
    const size_t lds = (size_t)entries * 4;
    // every read covers its window and the plan fits the specialised forms (key <= 32 bytes -- always --, at most 16 /
    // 32 key and 32 molecular bytes, every slice inside the words of its form): NS x KW x molecular words
    bool special = QD_GENERIC_SPECIAL && !p.len[0] && !p.len[1] && p.M <= 32 && p.K <= 32;
    const int kw = p.K <= 16 ? 2 : 4, mwords = p.M == 0 ? 0 : (p.M <= 16 ? 2 : 4);
    for (int k = 0; k < p.n_streams; ++k) special = special && p.idx_w[k] <= 8 * kw && p.mol_w[k] <= 8 * (mwords ? mwords : 1);
    if (special) {
        if (p.n_streams == 1) {
            if (kw == 2) launch_special_m<1, 2>(p, grid, lds, entries, st);
            else launch_special_m<1, 4>(p, grid, lds, entries, st);
        } else {
            if (kw == 2) launch_special_m<2, 2>(p, grid, lds, entries, st);
            else launch_special_m<2, 4>(p, grid, lds, entries, st);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(demux_generic, dim3(grid), dim3(QD_GEN_BLOCK), lds, st, p, entries);
    return hipGetLastError();
}

hipError_t qd_launch_fixup(const DemuxParams& p, hipStream_t st) {
    if (p.n_exc == 0) return hipSuccess;
    const unsigned grid = (p.n_exc + QD_GEN_BLOCK - 1) / QD_GEN_BLOCK;
    hipLaunchKernelGGL(demux_fixup, dim3(grid > 1024 ? 1024 : grid), dim3(QD_GEN_BLOCK), 0, st, p);
    return hipGetLastError();
}

