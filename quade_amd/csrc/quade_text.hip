// Device-side fastq text stages for gfx950 (MI355X): see quade_text.h for what each one replaces in the reference.
//
// All of it is byte / integer work bound by HBM or by latency, no MFMA.  Shapes:
//   line kernels ..... one workgroup per 16 KiB of text, every lane 64 consecutive bytes (SWAR newline test, 4 x 16-byte
//                      loads), counts -> exclusive scan -> positions: the text is read twice, nothing else is
//   record kernels ... one lane per record (4 line ends = one 16-byte load), kept records compacted in text order
//   sort ............. stable LSD radix over the 16-bit destination, one WAVE per 1 024 pairs with its counters in LDS,
//                      peers of a lane found with 8 ballots (no atomics, so the order inside a destination is the input's)
//   format ........... 16 lanes per output record, segment after segment
//   CRC-32 ........... 256 lanes per 64 KiB range, slice-by-4 out of LDS tables, lanes' CRCs combined by x^(8n) mod P
#include <hip/hip_runtime.h>

#include "quade_inflate.h"
#include "quade_text.h"
#include "text_rules.h"

namespace {

constexpr uint32_t TILE = QD_TEXT_TILE;
constexpr uint32_t LINE_BLOCK = 256;  // x 64 bytes = one tile
static_assert(LINE_BLOCK * 64 == TILE, "a lane owns 64 bytes of its tile");

struct __attribute__((packed, aligned(1))) U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ uint32_t nl_bits4(uint32_t w) {  // bit k set: byte k of w is '\n'
    const uint32_t x = w ^ 0x0A0A0A0Au;
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 exactly in the zero bytes
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}

// newline mask of the 64 bytes at b: bit i = text[b + i] is a newline, or b + i == len and the window ends the file with
// an unterminated line (virt)
__device__ __forceinline__ uint64_t nl_mask64(const uint8_t* text, uint32_t b, uint32_t len, bool virt) {
    if (b > len) return 0;
    uint64_t m = 0;
    if (b < len) {
        const uint4* p = reinterpret_cast<const uint4*>(text + b);  // (the buffer is padded to a whole tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = p[q];
            const uint64_t bits = (uint64_t)nl_bits4(v.x) | ((uint64_t)nl_bits4(v.y) << 4) | ((uint64_t)nl_bits4(v.z) << 8) |
                                  ((uint64_t)nl_bits4(v.w) << 12);
            m |= bits << (16 * q);
        }
        if (len - b < 64) m &= (1ull << (len - b)) - 1;
    }
    if (virt && len - b < 64) m |= 1ull << (len - b);
    return m;
}

__device__ __forceinline__ bool virtual_newline(const uint8_t* text, uint32_t len, int at_eof) {
    return at_eof && len > 0 && text[len - 1] != '\n';
}

// inclusive scan over the wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += y;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// exclusive scan of one value per thread over a workgroup of NW waves; *total = the workgroup's sum
template <int NW>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* lds /* NW words */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_incl(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t s = lds[i];
        if ((uint32_t)i < wave) before += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

// ---- lines ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LINE_BLOCK) void line_count(const uint8_t* text, uint32_t len, int at_eof, uint32_t* tile_counts) {
    __shared__ uint32_t part[LINE_BLOCK / 64];
    const bool virt = virtual_newline(text, len, at_eof);
    const uint32_t b = blockIdx.x * TILE + threadIdx.x * 64u;
    const uint32_t c = wave_sum((uint32_t)__popcll(nl_mask64(text, b, len, virt)));
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of n words by one workgroup; out[n] = the total.  in == out is fine.
__global__ __launch_bounds__(1024) void scan_single(const uint32_t* in, uint32_t n, uint32_t* out) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_excl<16>(v, lds, &total);
        const uint32_t c = carry;
        if (i < n) out[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// the small serial kernels for several windows at once (qd_text_scan_many): workgroup w does window w's
struct ScanMany {
    const uint32_t* in[4];
    uint32_t* out[4];
    uint32_t n[4];
};
__global__ __launch_bounds__(1024) void scan_single_many(const ScanMany a) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t carry;
    const uint32_t* in = a.in[blockIdx.x];
    uint32_t* out = a.out[blockIdx.x];
    const uint32_t n = a.n[blockIdx.x];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_excl<16>(v, lds, &total);
        const uint32_t c = carry;
        if (i < n) out[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}
struct DoneMany {
    const uint32_t* a[4];      // lines_done: tile_base; recs_done: rec_tile
    const uint32_t* lines[4];  // (recs_done)
    uint32_t n[4];             // lines_done: n_tiles; recs_done: n_tiles_max
    uint32_t cap[4];           // (lines_done: line_cap)
    qd_scan_result* r[4];
};
__global__ void lines_done_many(const DoneMany d) {
    const uint32_t w = threadIdx.x;
    const uint32_t n = d.a[w][d.n[w]];
    qd_scan_result* r = d.r[w];
    r->n_lines = n;
    r->n_records = n / 4;
    r->overflow = n > d.cap[w] ? 1u : 0u;
    r->n_short = 0;
}
__global__ void recs_done_many(const DoneMany d) {
    const uint32_t w = threadIdx.x;
    qd_scan_result* r = d.r[w];
    const uint32_t n_rec = r->overflow ? 0u : r->n_records;
    const uint32_t n_tiles = (n_rec + 1024u - 1) / 1024u;  // (REC_TILE)
    r->n_kept = n_tiles <= d.n[w] ? d.a[w][n_tiles] : 0u;
    r->tail_start = n_rec ? d.lines[w][4 * n_rec - 1] + 1 : 0u;
}

__global__ void lines_done(const uint32_t* tile_base, uint32_t n_tiles, uint32_t line_cap, qd_scan_result* r) {
    const uint32_t n = tile_base[n_tiles];
    r->n_lines = n;
    r->n_records = n / 4;
    r->overflow = n > line_cap ? 1u : 0u;
    r->n_short = 0;
}

__global__ __launch_bounds__(LINE_BLOCK) void line_write(const uint8_t* text, uint32_t len, int at_eof, const uint32_t* tile_base,
                                                         uint32_t* lines, uint32_t line_cap) {
    __shared__ uint32_t lds[LINE_BLOCK / 64];
    const bool virt = virtual_newline(text, len, at_eof);
    const uint32_t b = blockIdx.x * TILE + threadIdx.x * 64u;
    uint64_t m = nl_mask64(text, b, len, virt);
    uint32_t total;
    uint32_t at = tile_base[blockIdx.x] + block_scan_excl<LINE_BLOCK / 64>((uint32_t)__popcll(m), lds, &total);
    while (m) {
        if (at < line_cap) lines[at] = b + (uint32_t)__builtin_ctzll(m);
        ++at;
        m &= m - 1;
    }
}

// ---- records ----------------------------------------------------------------------------------------------------------
constexpr uint32_t REC_TILE = 1024;  // records per workgroup of 256 lanes, 4 consecutive ones per lane

struct Lines4 {
    uint32_t head, e0, e1, e2, e3;
};
__device__ __forceinline__ Lines4 record_lines(const uint32_t* lines, uint32_t r) {
    const uint4 e = reinterpret_cast<const uint4*>(lines)[r];
    Lines4 l;
    l.head = r ? lines[4 * r - 1] + 1 : 0u;
    l.e0 = e.x;
    l.e1 = e.y;
    l.e2 = e.z;
    l.e3 = e.w;
    return l;
}
// sequence and quality lines without a trailing '\r'; true when they have one length (the record is kept)
__device__ __forceinline__ bool record_kept(const uint8_t* text, const Lines4& l, uint32_t* seq_len) {
    const uint32_t seq = l.e0 + 1, qual = l.e2 + 1;
    const uint32_t seq_end = (l.e1 > seq && text[l.e1 - 1] == '\r') ? l.e1 - 1 : l.e1;
    const uint32_t qual_end = (l.e3 > qual && text[l.e3 - 1] == '\r') ? l.e3 - 1 : l.e3;
    *seq_len = seq_end - seq;
    return seq_end - seq == qual_end - qual;
}

__global__ __launch_bounds__(256) void rec_count(const uint8_t* text, const uint32_t* lines, const qd_scan_result* res, uint32_t need,
                                                 uint32_t* rec_tile, qd_scan_result* out) {
    __shared__ uint32_t part[4], parts[4];
    const uint32_t n_rec = res->overflow ? 0u : res->n_records;
    uint32_t kept = 0, shorts = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = blockIdx.x * REC_TILE + threadIdx.x * 4u + (uint32_t)i;
        if (r < n_rec) {
            uint32_t sl;
            if (record_kept(text, record_lines(lines, r), &sl)) {
                ++kept;
                shorts += sl < need ? 1u : 0u;
            }
        }
    }
    kept = wave_sum(kept);
    shorts = wave_sum(shorts);
    if ((threadIdx.x & 63u) == 0) {
        part[threadIdx.x >> 6] = kept;
        parts[threadIdx.x >> 6] = shorts;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        rec_tile[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        const uint32_t s = parts[0] + parts[1] + parts[2] + parts[3];
        if (s) atomicAdd(&out->n_short, s);
    }
}

__global__ void recs_done(const uint32_t* rec_tile, uint32_t n_tiles_max, const uint32_t* lines, qd_scan_result* r) {
    const uint32_t n_rec = r->overflow ? 0u : r->n_records;
    const uint32_t n_tiles = (n_rec + REC_TILE - 1) / REC_TILE;
    r->n_kept = n_tiles <= n_tiles_max ? rec_tile[n_tiles] : 0u;  // (scan_single left the total behind the last tile it scanned)
    r->tail_start = n_rec ? lines[4 * n_rec - 1] + 1 : 0u;
}

// qd_name_of() for a lane: the token's end is looked for eight bytes per load (the byte loop is one dependent trip to memory per
// character: ~30 per name, most of this kernel's time).  Reads up to 7 bytes behind the line: inside the window's padding.
__device__ __forceinline__ void name_of_wide(const uint8_t* text, uint32_t head, uint32_t line_end, uint32_t* name_off, uint32_t* name_len) {
    uint32_t h = head + (line_end > head ? 1u : 0u);
    while (h < line_end && qd_is_blank(text[h])) ++h;
    uint32_t e = h;
    while (e < line_end) {
        uint64_t v;
        __builtin_memcpy(&v, text + e, 8);  // (one unaligned 8-byte load)
        uint32_t k = 8;
#pragma unroll
        for (int i = 7; i >= 0; --i)
            if (qd_is_blank((uint8_t)(v >> (8 * i)))) k = (uint32_t)i;
        e += k;
        if (k < 8) break;
    }
    if (e > line_end) e = line_end;
    *name_off = h;
    *name_len = e - h;
}

__global__ __launch_bounds__(256) void rec_write(const uint8_t* text, const uint32_t* lines, const qd_scan_result* res, int want_names,
                                                 const uint32_t* rec_tile, qd_rec* recs) {
    __shared__ uint32_t lds[4];
    const uint32_t n_rec = res->overflow ? 0u : res->n_records;
    Lines4 l[4];
    uint32_t sl[4];
    bool keep[4];
    uint32_t kept = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = blockIdx.x * REC_TILE + threadIdx.x * 4u + (uint32_t)i;
        keep[i] = false;
        if (r < n_rec) {
            l[i] = record_lines(lines, r);
            keep[i] = record_kept(text, l[i], &sl[i]);
        }
        kept += keep[i] ? 1u : 0u;
    }
    uint32_t total;
    uint32_t at = rec_tile[blockIdx.x] + block_scan_excl<4>(kept, lds, &total);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!keep[i]) continue;
        qd_rec q;
        q.head = l[i].head;
        q.seq = l[i].e0 + 1;
        q.seq_len = sl[i];
        q.qual = l[i].e2 + 1;
        q.name_off = q.head;
        q.name_len = 0;
        if (want_names) name_of_wide(text, l[i].head, l[i].e0, &q.name_off, &q.name_len);
        recs[at++] = q;
    }
}

__global__ void carry_info(const qd_rec* r0, const qd_rec* r1, const qd_rec* r2, const qd_rec* r3, qd_scan_result* s0, qd_scan_result* s1,
                           qd_scan_result* s2, qd_scan_result* s3, int n_streams, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3) {
    const qd_rec* recs[4] = {r0, r1, r2, r3};
    qd_scan_result* res[4] = {s0, s1, s2, s3};
    const uint32_t ns[4] = {n0, n1, n2, n3};
    const int k = (int)threadIdx.x;
    if (k < n_streams) res[k]->carry_start = ns[k] < res[k]->n_kept ? recs[k][ns[k]].head : res[k]->tail_start;
}

// ---- grains: a stream's text cut at BGZF block boundaries, indexed without knowing where its records start -----------------------
// A chunk that several ranks share is cut into grains; a grain's rank does not know how many lines lie before it in the file, so
// it counts for all four residues: a line whose end is newline i of the window heads a record under the phase that makes
// (lines before the grain + i - first) a multiple of 4.  Every line is the candidate of exactly one phase of its grain.
__global__ void grain_bounds(const uint32_t* lines, const qd_scan_result* res, const uint32_t* grain_start, uint32_t n_grains, qd_grain_index* out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_grains) return;
    const uint32_t n = res->overflow ? 0u : res->n_lines;
    auto lower = [&](uint32_t pos) {  // newlines before `pos`
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (lines[mid] < pos) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    qd_grain_index r;
    r.first_line = lower(grain_start[g]);
    r.n_lines = lower(grain_start[g + 1]) - r.first_line;
    for (int k = 0; k < 4; ++k) {
        r.kept[k] = 0;
        r.first_head[k] = 0xFFFFFFFFu;
        r.incomplete[k] = 0;
    }
    out[g] = r;
}

__global__ __launch_bounds__(256) void grain_index(const uint8_t* text, const uint32_t* lines, const qd_scan_result* res, const uint32_t* grain_start,
                                                   uint32_t n_grains, int at_eof, qd_grain_index* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // newline i ends a candidate header line
    const uint32_t n = res->overflow ? 0u : res->n_lines;
    if (i >= n) return;
    const uint32_t head = i ? lines[i - 1] + 1 : 0u;
    // the grain that owns a record is the one its header line STARTS in
    uint32_t lo = 0, hi = n_grains;  // the last g with grain_start[g] <= head
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (grain_start[mid] <= head) lo = mid;
        else hi = mid;
    }
    const uint32_t g = lo;
    if (head < grain_start[g] || head >= grain_start[g + 1]) return;  // (before the first grain / in the overlap behind the last)
    const uint32_t phase = (out[g].first_line - i) & 3u;  // lines before the grain, mod 4, under which line i is a header
    if (i + 3 >= n) {  // the record's lines are not all in the window
        if (!at_eof) atomicOr(&out[g].incomplete[phase], 1u);
        return;
    }
    Lines4 l;
    l.head = head;
    l.e0 = lines[i];
    l.e1 = lines[i + 1];
    l.e2 = lines[i + 2];
    l.e3 = lines[i + 3];
    uint32_t sl;
    if (record_kept(text, l, &sl)) {
        atomicAdd(&out[g].kept[phase], 1u);
        atomicMin(&out[g].first_head[phase], head);
    }
}

// ---- index rows -------------------------------------------------------------------------------------------------------
struct PackParams {
    qd_pack_args a;
    int32_t n_streams;
    int32_t seq_off[2], seq_width[2], seq_stride[2];
    int32_t qual_off[2], qual_width[2], qual_stride[2];
};

__global__ __launch_bounds__(256) void pack_rows(PackParams p, uint32_t n) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    bool is_short = false;
    for (int k = 0; k < p.n_streams; ++k) {
        const qd_rec r = p.a.recs[k][j];
        const uint8_t* t = p.a.text[k];
        const uint32_t len = r.seq_len;
        {
            const uint32_t so = (uint32_t)p.seq_off[k], sw = (uint32_t)p.seq_width[k], ss = (uint32_t)p.seq_stride[k];
            const uint32_t avail = len > so ? min(len - so, sw) : 0u;
            uint16_t* row = reinterpret_cast<uint16_t*>(p.a.seq[k] + (size_t)j * ss);  // (strides are even, the arrays 16-byte aligned)
            const uint8_t* src = t + r.seq + so;
            for (uint32_t i = 0; i < ss; i += 2) {
                const uint32_t lo = i < avail ? src[i] : 0u, hi = i + 1 < avail ? src[i + 1] : 0u;
                row[i >> 1] = (uint16_t)(lo | (hi << 8));
            }
            is_short = is_short || len < so + sw;
        }
        {
            const uint32_t qo = (uint32_t)p.qual_off[k], qw = (uint32_t)p.qual_width[k], qs = (uint32_t)p.qual_stride[k];
            const uint32_t avail = len > qo ? min(len - qo, qw) : 0u;
            uint16_t* row = reinterpret_cast<uint16_t*>(p.a.qual[k] + (size_t)j * qs);
            const uint8_t* src = t + r.qual + qo;
            for (uint32_t i = 0; i < qs; i += 2) {
                const uint32_t lo = i < avail ? src[i] : 0xFFu, hi = i + 1 < avail ? src[i + 1] : 0xFFu;
                row[i >> 1] = (uint16_t)(lo | (hi << 8));
            }
        }
        p.a.len[k][j] = (uint8_t)(len > 255u ? 255u : len);
    }
    if (is_short) {
        const uint32_t at = atomicAdd(p.a.n_short, 1u);
        if (at < p.a.short_cap) p.a.short_idx[at] = j;
    }
}

// ---- destinations and output lengths ------------------------------------------------------------------------------------
struct PlanParams {
    int32_t n_streams;
    int32_t is[2], ie[2], ms[2], me[2];
    uint32_t n_samples;
    int32_t write_pass, write_fail, write_undet;
};
__host__ PlanParams plan_params(const qd_plan& P, uint32_t S, int wp, int wf, int wu) {
    PlanParams q;
    q.n_streams = P.dual ? 2 : 1;
    q.is[0] = P.idx1_start;
    q.ie[0] = P.idx1_end;
    q.ms[0] = P.mol1_start;
    q.me[0] = P.mol1_end;
    q.is[1] = P.dual ? P.idx2_start : 0;
    q.ie[1] = P.dual ? P.idx2_end : 0;
    q.ms[1] = P.dual ? P.mol2_start : 0;
    q.me[1] = P.dual ? P.mol2_end : 0;
    q.n_samples = S;
    q.write_pass = wp;
    q.write_fail = wf;
    q.write_undet = wu;
    return q;
}
__device__ __forceinline__ bool dest_enabled(const PlanParams& p, uint32_t d) {
    if (d == 2 * p.n_samples) return p.write_undet != 0;
    return (d & 1u) ? p.write_fail != 0 : p.write_pass != 0;
}
// ":IDX" or ":IDX:MOL" (src/FastqWriter.py:61-66): lengths of the two parts for the reads' lengths
__device__ __forceinline__ void tag_parts(const PlanParams& p, const uint32_t len[2], uint32_t* idx_bytes, uint32_t* mol_bytes) {
    uint32_t ib = 0, mb = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (k < p.n_streams) {
            ib += qd_slice_len(p.is[k], p.ie[k], len[k]);
            mb += qd_slice_len(p.ms[k], p.me[k], len[k]);
        }
    *idx_bytes = ib;
    *mol_bytes = mb;
}

__global__ __launch_bounds__(256) void dest_lens(PlanParams p, qd_route_args a, uint32_t n) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint32_t c = a.codes[j];
    const uint32_t d = c == QD_CODE_UNDETERMINED ? 2 * p.n_samples : min(c, 2 * p.n_samples);
    a.dest[j] = (uint16_t)d;
    uint32_t l1 = 0, l2 = 0;
    if (dest_enabled(p, d)) {
        uint32_t len[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < p.n_streams) len[k] = a.idx[k][j].seq_len;
        uint32_t ib, mb;
        tag_parts(p, len, &ib, &mb);
        const uint32_t tag = 1 + ib + (mb ? 1 + mb : 0);
        const qd_rec r1 = a.r1[j], r2 = a.r2[j];
        l1 = r1.name_len + tag + 2 * r1.seq_len + 6;  // '@' name tag '\n' seq '\n' '+' '\n' qual '\n'
        l2 = r2.name_len + tag + 2 * r2.seq_len + 6;
    }
    a.len1[j] = l1;
    a.len2[j] = l2;
}

// ---- stable radix sort by destination -------------------------------------------------------------------------------------
constexpr uint32_t SORT_SUB = 1024;  // pairs per wave

__device__ __forceinline__ uint32_t sort_key(const uint16_t* dest, const uint32_t* src, uint32_t i, int shift) {
    return ((uint32_t)dest[src ? src[i] : i] >> shift) & 0xFFu;
}

__global__ __launch_bounds__(256) void radix_hist(const uint16_t* dest, const uint32_t* src, uint32_t n, int shift, uint32_t n_sub,
                                                  uint32_t* hist) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = blockIdx.x * 4u + wave;
    for (uint32_t i = lane; i < 256; i += 64) cnt[wave][i] = 0;
    if (sub < n_sub) {
        for (uint32_t step = 0; step < SORT_SUB / 64; ++step) {
            const uint32_t i = sub * SORT_SUB + step * 64u + lane;
            if (i < n) atomicAdd(&cnt[wave][sort_key(dest, src, i, shift)], 1u);
        }
        for (uint32_t dgt = lane; dgt < 256; dgt += 64) hist[(size_t)dgt * n_sub + sub] = cnt[wave][dgt];
    }
}

__global__ __launch_bounds__(256) void radix_scatter(const uint16_t* dest, const uint32_t* src, uint32_t n, int shift, uint32_t n_sub,
                                                     const uint32_t* hist, uint32_t* out) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = blockIdx.x * 4u + wave;
    if (sub >= n_sub) return;  // (no workgroup barrier below: a wave is on its own)
    for (uint32_t dgt = lane; dgt < 256; dgt += 64) cnt[wave][dgt] = hist[(size_t)dgt * n_sub + sub];
    const uint64_t below = (1ull << lane) - 1;
    for (uint32_t step = 0; step < SORT_SUB / 64; ++step) {
        const uint32_t i = sub * SORT_SUB + step * 64u + lane;
        const bool valid = i < n;
        const uint32_t dgt = valid ? sort_key(dest, src, i, shift) : 0u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = (dgt >> bit) & 1u;
            const uint64_t b = __ballot(valid && one);
            peers &= one ? b : ~b;
        }
        if (valid) {
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            const uint32_t base = cnt[wave][dgt];
            out[base + rank] = src ? src[i] : i;
            if (rank == 0) cnt[wave][dgt] = base + (uint32_t)__popcll(peers);  // (the wave's LDS accesses complete in order)
        }
    }
}

// ---- exclusive scans over many elements: tile sums, one workgroup over the sums, tiles again ----------------------------------
constexpr uint32_t SCAN_TILE = 4096;  // 256 lanes x 16

template <bool GATHER>
__global__ __launch_bounds__(256) void scan_tile_sums(const uint32_t* in, const uint32_t* perm, uint32_t n, uint32_t* tiles) {
    __shared__ uint32_t part[4];
    uint32_t s = 0;
    const uint32_t b = blockIdx.x * SCAN_TILE + threadIdx.x * 16u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t k = b + (uint32_t)i;
        if (k < n) s += in[GATHER ? perm[k] : k];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tiles[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

template <bool GATHER>
__global__ __launch_bounds__(256) void scan_tile_apply(const uint32_t* in, const uint32_t* perm, uint32_t n, const uint32_t* tiles,
                                                       uint32_t n_tiles, uint32_t* out, const uint16_t* dest, uint16_t* sdest) {
    __shared__ uint32_t lds[4];
    uint32_t v[16];
    uint32_t s = 0;
    const uint32_t b = blockIdx.x * SCAN_TILE + threadIdx.x * 16u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t k = b + (uint32_t)i;
        v[i] = 0;
        if (k < n) {
            const uint32_t j = GATHER ? perm[k] : k;
            v[i] = in[j];
            if (sdest) sdest[k] = dest[j];
        }
        s += v[i];
    }
    uint32_t total;
    uint32_t at = tiles[blockIdx.x] + block_scan_excl<4>(s, lds, &total);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t k = b + (uint32_t)i;
        if (k < n) out[k] = at;
        at += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tiles[n_tiles];
}

__global__ __launch_bounds__(256) void dest_bounds(const uint16_t* sdest, const uint32_t* g1, const uint32_t* g2, uint32_t n, uint32_t* first,
                                                   uint32_t* g1_first, uint32_t* g2_first) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint32_t d = sdest[k];
    if (k == 0 || sdest[k - 1] != d) {
        first[d] = k;
        g1_first[d] = g1[k];
        g2_first[d] = g2[k];
    }
}

// ---- format ---------------------------------------------------------------------------------------------------------------
// 16 lanes per output record: '@' name ':' IDX [':' MOL] '\n' seq "\n+\n" qual '\n' (src/FastqWriter.py:61-69)
// (eight bytes per lane and step where the piece is long enough -- unaligned 8-byte loads and stores are the hardware's business --
//  and the rest byte by byte: a read's 150 bases are two steps and six bytes instead of ten steps)
__device__ __forceinline__ void put(uint8_t* out, const uint8_t* src, uint32_t len, uint32_t sub) {
    const uint32_t whole = len & ~7u;
    for (uint32_t i = sub * 8u; i < whole; i += 128) {
        uint64_t v;
        __builtin_memcpy(&v, src + i, 8);
        __builtin_memcpy(out + i, &v, 8);
    }
    for (uint32_t i = whole + sub; i < len; i += 16) out[i] = src[i];
}

__global__ __launch_bounds__(256) void format_records(PlanParams p, qd_format_args a, uint32_t n) {
    const uint32_t item = blockIdx.x * 16u + (threadIdx.x >> 4), sub = threadIdx.x & 15u;
    const uint32_t k = item >> 1, read = item & 1u;
    if (k >= n) return;
    const uint32_t d = a.sdest[k];
    if (!dest_enabled(p, d)) return;
    const uint32_t j = a.perm[k];
    const qd_rec r = read ? a.r2[j] : a.r1[j];
    const uint8_t* text = read ? a.text2 : a.text1;
    uint8_t* o = (read ? a.out2 : a.out1) + ((read ? a.base2[d] : a.base1[d]) + (int64_t)(read ? a.g2[k] : a.g1[k]));
    if (sub == 0) o[0] = '@';
    ++o;
    put(o, text + r.name_off, r.name_len, sub);
    o += r.name_len;
    // the tag: raw-case slices of the index reads, clamped to their lengths
    uint32_t len[2] = {0, 0}, iseq[2] = {0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (s < p.n_streams) {
            const qd_rec ir = a.idx[s][j];
            len[s] = ir.seq_len;
            iseq[s] = ir.seq;
        }
    if (sub == 0) o[0] = ':';
    ++o;
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (s < p.n_streams) {
            const uint32_t w = qd_slice_len(p.is[s], p.ie[s], len[s]);
            put(o, a.itext[s] + iseq[s] + p.is[s], w, sub);
            o += w;
        }
    uint32_t ib, mb;
    tag_parts(p, len, &ib, &mb);
    if (mb) {
        if (sub == 0) o[0] = ':';
        ++o;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (s < p.n_streams) {
                const uint32_t w = qd_slice_len(p.ms[s], p.me[s], len[s]);
                put(o, a.itext[s] + iseq[s] + p.ms[s], w, sub);
                o += w;
            }
    }
    if (sub == 0) o[0] = '\n';
    ++o;
    put(o, text + r.seq, r.seq_len, sub);
    o += r.seq_len;
    if (sub < 3) o[sub] = sub == 1 ? '+' : '\n';
    o += 3;
    put(o, text + r.qual, r.seq_len, sub);
    o += r.seq_len;
    if (sub == 0) o[0] = '\n';
}

// ---- CRC-32 -----------------------------------------------------------------------------------------------------------------
constexpr uint32_t CRC_SLICE = 256;  // bytes per lane: 256 lanes cover a 64 KiB range

// the workgroup's CRC-32 of text[off .. off + len), len <= 64 KiB; the result is valid in thread 0
// The range comes into LDS first, by loads that neighbouring lanes make from neighbouring dwords; every lane then takes its 256-byte
// slice from there (rows of 64 dwords padded to 65: lane L's k-th dword lies in bank L + k).  The first form had every lane read its
// slice from global memory a dword at a time: one load instruction touched 64 cache lines for 256 useful bytes, the lines were
// gone from the CU's cache before their other 31 dwords were asked for, and the kernel moved ~30 bytes from HBM per byte of text
// (profiles/r05_e2e_gz_timeline_before.txt: 6.1 ms for 1.5 GB).
__device__ __forceinline__ uint32_t crc32_range(const uint8_t* text, uint64_t off, uint32_t len_in) {
    __shared__ uint32_t T[4][256];
    __shared__ uint32_t X8[32];
    __shared__ uint32_t part[4];
    __shared__ uint32_t D[256 * 65 + 8];
    const uint32_t tid = threadIdx.x;
    {
        uint32_t c = tid;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        T[0][tid] = c;
    }
    if (tid == 0) qd_crc_pow_table(X8);
    const uint32_t len = len_in > 256u * CRC_SLICE ? 256u * CRC_SLICE : len_in;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(text + off) & 3u);  // the range starts `sh` bytes into its first dword
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(text + off - sh);
        const uint32_t nd = len ? (sh + len + 3u) >> 2 : 0u;
        for (uint32_t d = tid; d < nd; d += 256u) D[d + (d >> 6)] = src[d];
    }
    __syncthreads();
    {
        uint32_t c = T[0][tid];
        for (int t = 1; t < 4; ++t) {
            c = (c >> 8) ^ T[0][c & 0xFFu];
            T[t][tid] = c;
        }
    }
    __syncthreads();
    const uint32_t beg = min(tid * CRC_SLICE, len), end = min(beg + CRC_SLICE, len);
    uint32_t term = 0;
    if (end > beg) {
        auto byte_at = [&](uint32_t j) {
            const uint32_t b = sh + j, d = b >> 2;
            return (D[d + (d >> 6)] >> (8u * (b & 3u))) & 0xFFu;
        };
        uint32_t j = beg;
        uint32_t c = 0xFFFFFFFFu;
        while (j < end && ((sh + j) & 3u)) c = (c >> 8) ^ T[0][(c ^ byte_at(j++)) & 0xFFu];
        for (; j + 4 <= end; j += 4) {
            const uint32_t d = (sh + j) >> 2;
            c ^= D[d + (d >> 6)];
            c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
        }
        while (j < end) c = (c >> 8) ^ T[0][(c ^ byte_at(j++)) & 0xFFu];
        // crc(A || B) = crc(A) * x^(8 |B|) + crc(B) mod P: this lane's share of the range's CRC
        term = qd_crc_mulmod(~c, qd_crc_xpow8(X8, len - end));
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) term ^= __shfl_xor(term, dd, 64);
    if ((tid & 63u) == 0) part[tid >> 6] = term;
    __syncthreads();
    return part[0] ^ part[1] ^ part[2] ^ part[3];
}

__global__ __launch_bounds__(256) void crc32_ranges(const uint8_t* text, const qd_crc_range* ranges, uint32_t* crc) {
    const qd_crc_range rg = ranges[blockIdx.x];
    const uint32_t c = crc32_range(text, rg.off, rg.len);
    if (threadIdx.x == 0) crc[blockIdx.x] = c;
}
// the text of inflated BGZF blocks: block i's range is out[out_off .. out_off + out_len) of its table entry
__global__ __launch_bounds__(256) void crc32_blocks(const uint8_t* out, const qd_inflate_block* blocks, uint32_t* crc) {
    const qd_inflate_block b = blocks[blockIdx.x];
    const uint32_t c = crc32_range(out, b.out_off, b.out_len);
    if (threadIdx.x == 0) crc[blockIdx.x] = c;
}

__global__ __launch_bounds__(64) void crc32_combine(const qd_crc_range* ranges, const uint32_t* crc, const uint32_t* first, uint32_t n_pieces,
                                                    uint32_t* piece_crc, uint32_t stride_words) {
    __shared__ uint32_t X8[32];
    if (threadIdx.x == 0) qd_crc_pow_table(X8);
    __syncthreads();
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_pieces) return;
    uint32_t c = 0;
    for (uint32_t r = first[i]; r < first[i + 1]; ++r) c = qd_crc_mulmod(c, qd_crc_xpow8(X8, ranges[r].len)) ^ crc[r];
    piece_crc[(size_t)i * stride_words] = c;
}

__global__ __launch_bounds__(256) void check_blocks(const int32_t* status, const uint32_t* crc, const uint32_t* expect, uint32_t n,
                                                    uint32_t base_index, uint32_t* first_bad) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && (status[i] != 0 || crc[i] != expect[i])) atomicMin(first_bad, base_index + i);
}

// ---- members -> one packed byte stream ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void member_offsets(const uint32_t* len, uint32_t n, uint64_t* offsets) {
    __shared__ uint32_t lds[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t total;
        const uint32_t ex = block_scan_excl<16>(i < n ? len[i] : 0u, lds, &total);  // (1 024 members of < 4 MiB each)
        const uint64_t c = carry;
        if (i < n) offsets[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}

constexpr uint32_t COPY_PART = 65536;
__global__ __launch_bounds__(256) void member_copy(const uint8_t* slots, int64_t stride, const uint32_t* len, const uint64_t* offsets,
                                                   uint8_t* packed) {
    const uint32_t i = blockIdx.x;
    const uint32_t L = len[i];
    const uint32_t b0 = blockIdx.y * COPY_PART;
    if (b0 >= L) return;
    const uint32_t b1 = min(b0 + COPY_PART, L);
    const uint8_t* s = slots + (int64_t)i * stride;
    uint8_t* d = packed + offsets[i];
    uint32_t b = b0 + threadIdx.x * 16u;
    for (; b + 16 <= b1; b += 256 * 16) *reinterpret_cast<U4*>(d + b) = *reinterpret_cast<const U4*>(s + b);
    if (b < b1)
        for (uint32_t q = b; q < b1; ++q) d[q] = s[q];
}

}  // namespace

// ---- launches -------------------------------------------------------------------------------------------------------------------
hipError_t qd_text_scan(const uint8_t* text, uint32_t len, int at_eof, int want_names, uint32_t need, const qd_scan_scratch& s,
                        qd_scan_result* result, hipStream_t st) {
    const uint32_t n_tiles = len / TILE + 1;  // covers byte `len` too (the unterminated last line of a file)
    hipLaunchKernelGGL(line_count, dim3(n_tiles), dim3(LINE_BLOCK), 0, st, text, len, at_eof, s.tile_counts);
    hipLaunchKernelGGL(scan_single, dim3(1), dim3(1024), 0, st, s.tile_counts, n_tiles, s.tile_base);
    hipLaunchKernelGGL(lines_done, dim3(1), dim3(1), 0, st, s.tile_base, n_tiles, s.line_cap, result);
    hipLaunchKernelGGL(line_write, dim3(n_tiles), dim3(LINE_BLOCK), 0, st, text, len, at_eof, s.tile_base, s.lines, s.line_cap);
    // the record kernels size themselves from the device's line count: launch for the most the table can hold
    const uint32_t rec_cap = s.line_cap / 4;
    const uint32_t rec_tiles = (rec_cap + REC_TILE - 1) / REC_TILE;
    if (rec_tiles) {
        hipLaunchKernelGGL(rec_count, dim3(rec_tiles), dim3(256), 0, st, text, s.lines, result, need, s.rec_tile, result);
        hipLaunchKernelGGL(scan_single, dim3(1), dim3(1024), 0, st, s.rec_tile, rec_tiles, s.rec_tile);
        hipLaunchKernelGGL(recs_done, dim3(1), dim3(1), 0, st, s.rec_tile, rec_tiles, s.lines, result);
        hipLaunchKernelGGL(rec_write, dim3(rec_tiles), dim3(256), 0, st, text, s.lines, result, want_names, s.rec_tile, s.recs);
    } else {
        hipLaunchKernelGGL(recs_done, dim3(1), dim3(1), 0, st, s.rec_tile, 0u, s.lines, result);
    }
    return hipGetLastError();
}

// The scans of several windows (a batch's four streams) stage by stage: the big kernels of all windows back to back, the small serial
// ones -- a one-workgroup scan of the tile counts, a one-thread kernel that reads its total -- once for all windows with a workgroup
// (a thread) per window.  Window by window they were 2 x 0.4 ms of one workgroup per large stream with the device idle beside it
// (profiles/r05_e2e_bgzf_timeline.txt: ~8 ms of scans per batch of 2 M pairs).
hipError_t qd_text_scan_many(int n_windows, const qd_scan_job* jobs, hipStream_t st) {
    if (n_windows < 1 || n_windows > 4) return hipErrorInvalidValue;
    static_assert(REC_TILE == 1024, "recs_done_many");
    ScanMany sm{}, sr{};
    DoneMany dl{}, dr{};
    uint32_t n_tiles[4], rec_tiles[4];
    bool any_rec = false;
    for (int w = 0; w < n_windows; ++w) {
        const qd_scan_job& j = jobs[w];
        n_tiles[w] = j.len / TILE + 1;
        rec_tiles[w] = (j.s.line_cap / 4 + REC_TILE - 1) / REC_TILE;
        any_rec = any_rec || rec_tiles[w];
        sm.in[w] = j.s.tile_counts, sm.out[w] = j.s.tile_base, sm.n[w] = n_tiles[w];
        sr.in[w] = j.s.rec_tile, sr.out[w] = j.s.rec_tile, sr.n[w] = rec_tiles[w];
        dl.a[w] = j.s.tile_base, dl.n[w] = n_tiles[w], dl.cap[w] = j.s.line_cap, dl.r[w] = j.result;
        dr.a[w] = j.s.rec_tile, dr.lines[w] = j.s.lines, dr.n[w] = rec_tiles[w], dr.r[w] = j.result;
    }
    for (int w = 0; w < n_windows; ++w)
        hipLaunchKernelGGL(line_count, dim3(n_tiles[w]), dim3(LINE_BLOCK), 0, st, jobs[w].text, jobs[w].len, jobs[w].at_eof, jobs[w].s.tile_counts);
    hipLaunchKernelGGL(scan_single_many, dim3(n_windows), dim3(1024), 0, st, sm);
    hipLaunchKernelGGL(lines_done_many, dim3(1), dim3(n_windows), 0, st, dl);
    for (int w = 0; w < n_windows; ++w)
        hipLaunchKernelGGL(line_write, dim3(n_tiles[w]), dim3(LINE_BLOCK), 0, st, jobs[w].text, jobs[w].len, jobs[w].at_eof, jobs[w].s.tile_base, jobs[w].s.lines, jobs[w].s.line_cap);
    for (int w = 0; w < n_windows; ++w)
        if (rec_tiles[w])
            hipLaunchKernelGGL(rec_count, dim3(rec_tiles[w]), dim3(256), 0, st, jobs[w].text, jobs[w].s.lines, jobs[w].result, jobs[w].need, jobs[w].s.rec_tile, jobs[w].result);
    if (any_rec) hipLaunchKernelGGL(scan_single_many, dim3(n_windows), dim3(1024), 0, st, sr);  // (a window without record tiles: n = 0, its total is rec_tile[0] = 0 ... written)
    hipLaunchKernelGGL(recs_done_many, dim3(1), dim3(n_windows), 0, st, dr);
    for (int w = 0; w < n_windows; ++w)
        if (rec_tiles[w])
            hipLaunchKernelGGL(rec_write, dim3(rec_tiles[w]), dim3(256), 0, st, jobs[w].text, jobs[w].s.lines, jobs[w].result, jobs[w].want_names, jobs[w].s.rec_tile, jobs[w].s.recs);
    return hipGetLastError();
}

hipError_t qd_text_grain_index(const uint8_t* text, const uint32_t* lines, uint32_t line_cap, const qd_scan_result* res, const uint32_t* grain_start,
                               uint32_t n_grains, int at_eof, qd_grain_index* out, hipStream_t st) {
    if (!n_grains) return hipSuccess;
    hipLaunchKernelGGL(grain_bounds, dim3((n_grains + 63) / 64), dim3(64), 0, st, lines, res, grain_start, n_grains, out);
    if (line_cap) hipLaunchKernelGGL(grain_index, dim3((line_cap + 255) / 256), dim3(256), 0, st, text, lines, res, grain_start, n_grains, at_eof, out);
    return hipGetLastError();
}

hipError_t qd_text_carry_info(const qd_rec* const recs[4], qd_scan_result* const results[4], int n_streams, const uint32_t n[4], hipStream_t st) {
    hipLaunchKernelGGL(carry_info, dim3(1), dim3(64), 0, st, recs[0], recs[1], recs[2], recs[3], results[0], results[1], results[2], results[3],
                       n_streams, n[0], n[1], n[2], n[3]);
    return hipGetLastError();
}

hipError_t qd_text_pack_rows(const qd_layout& L, uint32_t n, const qd_pack_args& a, hipStream_t st) {
    if (!n) return hipSuccess;
    PackParams p;
    p.a = a;
    p.n_streams = L.n_streams;
    for (int k = 0; k < 2; ++k) {
        p.seq_off[k] = L.seq_off[k];
        p.seq_width[k] = L.seq_width[k];
        p.seq_stride[k] = L.seq_stride[k];
        p.qual_off[k] = L.qual_off[k];
        p.qual_width[k] = L.qual_width[k];
        p.qual_stride[k] = L.qual_stride[k];
    }
    hipLaunchKernelGGL(pack_rows, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
    return hipGetLastError();
}

hipError_t qd_text_dest_lens(const qd_plan& P, uint32_t n_samples, int write_pass, int write_fail, int write_undet, uint32_t n,
                             const qd_route_args& a, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(dest_lens, dim3((n + 255) / 256), dim3(256), 0, st, plan_params(P, n_samples, write_pass, write_fail, write_undet), a, n);
    return hipGetLastError();
}

hipError_t qd_text_sort_by_dest(const uint16_t* dest, uint32_t n, uint32_t n_dest, uint32_t* hist, uint32_t* tmp, uint32_t* perm,
                                hipStream_t st) {
    if (!n) return hipSuccess;
    const uint32_t n_sub = (n + SORT_SUB - 1) / SORT_SUB, grid = (n_sub + 3) / 4;
    const uint32_t n_hist = 256u * n_sub;
    const bool two = n_dest > 256;
    const uint32_t scan_tiles = (n_hist + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* tiles = hist + n_hist + 1;  // the scan's tile sums live behind the histograms (and their total)
    for (int pass = 0; pass < (two ? 2 : 1); ++pass) {
        const uint32_t* src = pass == 0 ? nullptr : tmp;
        uint32_t* out = (two && pass == 0) ? tmp : perm;
        const int shift = 8 * pass;
        hipLaunchKernelGGL(radix_hist, dim3(grid), dim3(256), 0, st, dest, src, n, shift, n_sub, hist);
        hipLaunchKernelGGL(scan_tile_sums<false>, dim3(scan_tiles), dim3(256), 0, st, hist, (const uint32_t*)nullptr, n_hist, tiles);
        hipLaunchKernelGGL(scan_single, dim3(1), dim3(1024), 0, st, tiles, scan_tiles, tiles);
        hipLaunchKernelGGL(scan_tile_apply<false>, dim3(scan_tiles), dim3(256), 0, st, hist, (const uint32_t*)nullptr, n_hist, tiles, scan_tiles,
                           hist, (const uint16_t*)nullptr, (uint16_t*)nullptr);
        hipLaunchKernelGGL(radix_scatter, dim3(grid), dim3(256), 0, st, dest, src, n, shift, n_sub, hist, out);
    }
    return hipGetLastError();
}

hipError_t qd_text_scan_gathered(const uint32_t* in, const uint32_t* perm, uint32_t n, uint32_t* tiles, uint32_t* out, const uint16_t* dest,
                                 uint16_t* sdest, hipStream_t st) {
    const uint32_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (!n) return hipMemsetAsync(out, 0, 4, st);
    hipLaunchKernelGGL(scan_tile_sums<true>, dim3(n_tiles), dim3(256), 0, st, in, perm, n, tiles);
    hipLaunchKernelGGL(scan_single, dim3(1), dim3(1024), 0, st, tiles, n_tiles, tiles);
    hipLaunchKernelGGL(scan_tile_apply<true>, dim3(n_tiles), dim3(256), 0, st, in, perm, n, tiles, n_tiles, out, dest, sdest);
    return hipGetLastError();
}

hipError_t qd_text_dest_bounds(const uint16_t* sdest, const uint32_t* g1, const uint32_t* g2, uint32_t n, uint32_t n_dest, uint32_t* first,
                               uint32_t* g1_first, uint32_t* g2_first, hipStream_t st) {
    hipError_t e = hipMemsetAsync(first, 0xFF, (size_t)n_dest * 4, st);
    if (e != hipSuccess || !n) return e;
    hipLaunchKernelGGL(dest_bounds, dim3((n + 255) / 256), dim3(256), 0, st, sdest, g1, g2, n, first, g1_first, g2_first);
    return hipGetLastError();
}

hipError_t qd_text_format(const qd_plan& P, uint32_t n_samples, int write_pass, int write_fail, int write_undet, uint32_t n,
                          const qd_format_args& a, hipStream_t st) {
    if (!n) return hipSuccess;
    const uint64_t items = 2ull * n;
    hipLaunchKernelGGL(format_records, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, st,
                       plan_params(P, n_samples, write_pass, write_fail, write_undet), a, n);
    return hipGetLastError();
}

hipError_t qd_text_crc32(const uint8_t* text, const qd_crc_range* ranges, uint32_t n, uint32_t* crc, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(crc32_ranges, dim3(n), dim3(256), 0, st, text, ranges, crc);
    return hipGetLastError();
}

hipError_t qd_text_crc32_blocks(const uint8_t* out, const qd_inflate_block* blocks, uint32_t n, uint32_t* crc, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(crc32_blocks, dim3(n), dim3(256), 0, st, out, blocks, crc);
    return hipGetLastError();
}

hipError_t qd_text_crc32_combine(const qd_crc_range* ranges, const uint32_t* crc, const uint32_t* first, uint32_t n_pieces, uint32_t* piece_crc,
                                 uint32_t stride_words, hipStream_t st) {
    if (!n_pieces) return hipSuccess;
    hipLaunchKernelGGL(crc32_combine, dim3((n_pieces + 63) / 64), dim3(64), 0, st, ranges, crc, first, n_pieces, piece_crc, stride_words);
    return hipGetLastError();
}

hipError_t qd_text_check_blocks(const int32_t* status, const uint32_t* crc, const uint32_t* expect, uint32_t n, uint32_t base_index,
                                uint32_t* first_bad, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(check_blocks, dim3((n + 255) / 256), dim3(256), 0, st, status, crc, expect, n, base_index, first_bad);
    return hipGetLastError();
}

hipError_t qd_text_pack_members(const uint8_t* slots, int64_t stride, const uint32_t* len, uint32_t n, uint64_t* offsets, uint8_t* packed,
                                hipStream_t st) {
    if (!n) return hipMemsetAsync(offsets, 0, 8, st);
    hipLaunchKernelGGL(member_offsets, dim3(1), dim3(1024), 0, st, len, n, offsets);
    const uint32_t parts = (uint32_t)((stride + COPY_PART - 1) / COPY_PART);
    hipLaunchKernelGGL(member_copy, dim3(n, parts), dim3(256), 0, st, slots, stride, len, offsets, packed);
    return hipGetLastError();
}
