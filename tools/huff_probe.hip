// Go / no-go probe for a GPU output stage: gzip members holding ONE dynamic-Huffman block of literals
// (what quade_io.cpp's huffman_member() writes on a host thread for `gzip_level : -1`) made on the MI355X.
// One workgroup (256 threads) per piece of formatted fastq text (~2 MB, the sink's JOB_BYTES):
//   1. byte histogram: coalesced 16-byte loads, per-wave x 4 replica LDS histograms
//   2. code lengths (<= 15 bits): one lane -- insertion sort of the used symbols, two-queue Huffman merge, the
//      Kraft-excess repair of huffman_lengths(); canonical codes, bit-reversed, into an LDS table (len | code << 8)
//   3. encode, tile by tile (4 KiB of text): each lane codes its 16 bytes into <= 240 bits, a workgroup scan gives
//      its bit offset, the bits are OR-ed into an LDS word buffer, whole words leave coalesced, the partial last
//      word is carried into the next tile
//   4. end-of-block code, byte alignment, CRC-32 (made on the host: libdeflate's PCLMUL CRC costs 0.1 core-s per GB)
//      and ISIZE
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/libhuff_probe.so tools/huff_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <vector>

namespace {
constexpr int BLOCK = 256, TILE = BLOCK * 16, HEADER_BITS = 80 + 3 + 14 + 19 * 3 + 259 * 4;  // gzip header + block header

__device__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// lengths (<= 15) of a Huffman code for the used ones of 257 symbols; same construction as huffman_lengths()
__device__ void code_lengths(const uint32_t* freq, uint8_t* len, uint16_t* order /*257*/, uint32_t* w /*513*/, int16_t* parent /*513*/) {
    int m = 0;
    for (int s = 0; s < 257; ++s) {
        len[s] = 0;
        if (freq[s]) order[m++] = (uint16_t)s;
    }
    for (int s = 0; m < 2 && s < 257; ++s) {
        bool used = false;
        for (int i = 0; i < m; ++i) used |= order[i] == s;
        if (!used) order[m++] = (uint16_t)s;
    }
    auto f = [&](int s) -> uint32_t { return freq[s] ? freq[s] : 1u; };
    for (int i = 1; i < m; ++i) {  // insertion sort by (frequency, symbol)
        const uint16_t s = order[i];
        int j = i - 1;
        while (j >= 0 && (f(order[j]) > f(s) || (f(order[j]) == f(s) && order[j] > s))) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = s;
    }
    for (int i = 0; i < m; ++i) w[i] = f(order[i]);
    for (int i = 0; i < 2 * m - 1; ++i) parent[i] = -1;
    int leaf = 0, inner = m, next = m;
    while (next < 2 * m - 1) {
        int pick[2];
        for (int k = 0; k < 2; ++k) pick[k] = (leaf < m && (inner >= next || w[leaf] <= w[inner])) ? leaf++ : inner++;
        w[next] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = parent[pick[1]] = (int16_t)next;
        ++next;
    }
    int bl[16] = {0};
    // depth of every node from the root down (w is reused for the depths)
    w[2 * m - 2] = 0;
    for (int k = 2 * m - 3; k >= 0; --k) w[k] = w[parent[k]] + 1;
    for (int i = 0; i < m; ++i) ++bl[w[i] < 15 ? w[i] : 15];
    uint32_t kraft = 0;
    for (int d = 1; d <= 15; ++d) kraft += (uint32_t)bl[d] << (15 - d);
    for (uint32_t excess = kraft - (1u << 15); excess > 0; --excess) {
        int bits = 14;
        while (bl[bits] == 0) --bits;
        --bl[bits];
        bl[bits + 1] += 2;
        --bl[15];
    }
    int at = 0;
    for (int bits = 15; bits >= 1; --bits)
        for (int c = 0; c < bl[bits]; ++c) len[order[at++]] = (uint8_t)bits;
}

__global__ __launch_bounds__(BLOCK) void huff_pieces(const uint8_t* text, const int64_t* off, const uint32_t* crc, uint8_t* out,
                                                     int64_t out_stride, uint32_t* out_bytes) {
    __shared__ uint32_t hist[4][4][256];
    __shared__ uint32_t freq[257];
    __shared__ uint32_t lut[257];  // len | code << 8 (code bit-reversed: DEFLATE sends Huffman codes MSB first)
    __shared__ uint8_t len[257];
    __shared__ uint16_t order[257];
    __shared__ uint32_t wtmp[513];
    __shared__ int16_t parent[513];
    __shared__ uint32_t scan[BLOCK];
    __shared__ uint32_t words[TILE * 15 / 32 + 16];
    __shared__ uint32_t carry_word, carry_bits;
    const int tid = threadIdx.x, wave = tid >> 6, rep = tid & 3;
    const int64_t a = off[blockIdx.x], L = off[blockIdx.x + 1] - a;
    const uint8_t* src = text + a;
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (int64_t)blockIdx.x * out_stride);
    // 1. histogram
    for (int i = tid; i < 4 * 4 * 256; i += BLOCK) (&hist[0][0][0])[i] = 0;
    __syncthreads();
    for (int64_t t = 0; t < L; t += TILE) {
        const int64_t p = t + (int64_t)tid * 16;
        if (p + 16 <= L) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + p);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) atomicAdd(&hist[wave][rep][(q[j >> 2] >> (8 * (j & 3))) & 0xff], 1u);
        } else {
            for (int64_t i = p; i < L; ++i) atomicAdd(&hist[wave][rep][src[i]], 1u);
        }
    }
    __syncthreads();
    for (int s = tid; s < 256; s += BLOCK) {
        uint32_t c = 0;
        for (int i = 0; i < 16; ++i) c += (&hist[0][0][0])[i * 256 + s];
        freq[s] = c;
    }
    if (tid == 0) freq[256] = 1;
    __syncthreads();
    // 2. the code
    if (tid == 0) {
        code_lengths(freq, len, order, wtmp, parent);
        int blc[16] = {0}, nxt[16] = {0};
        for (int s = 0; s < 257; ++s) ++blc[len[s]];
        blc[0] = 0;
        for (int b = 1, c = 0; b <= 15; ++b) {
            c = (c + blc[b - 1]) << 1;
            nxt[b] = c;
        }
        for (int s = 0; s < 257; ++s) lut[s] = len[s] ? ((uint32_t)len[s] | (rev_bits((uint32_t)nxt[len[s]]++, len[s]) << 8)) : 0;
        // gzip header + block header: 10 bytes, then BFINAL, dynamic, HLIT 0, HDIST 1, HCLEN 15, 19 x 3 bits, 259 x 4 bits
        uint64_t acc = 0;
        int cnt = 0, wi = 0;
        auto put = [&](uint32_t v, int n) {
            acc |= (uint64_t)v << cnt;
            cnt += n;
            if (cnt >= 32) {
                dst[wi++] = (uint32_t)acc;
                acc >>= 32;
                cnt -= 32;
            }
        };
        const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
        for (int i = 0; i < 10; ++i) put(head[i], 8);
        put(1, 1);
        put(2, 2);
        put(0, 5);
        put(1, 5);
        put(15, 4);
        const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < 19; ++k) put(ord[k] < 16 ? 4 : 0, 3);
        for (int s = 0; s < 257; ++s) put(rev_bits(len[s], 4), 4);
        put(rev_bits(1, 4), 4);
        put(rev_bits(1, 4), 4);
        carry_word = (uint32_t)acc;
        carry_bits = (uint32_t)cnt;
        scan[0] = (uint32_t)wi;  // words written so far
    }
    __syncthreads();
    uint32_t out_word = scan[0];
    __syncthreads();
    // 3. encode
    for (int64_t t = 0; t < L; t += TILE) {
        const int64_t p = t + (int64_t)tid * 16;
        uint32_t q[4] = {0, 0, 0, 0};
        int nbytes = 0;
        if (p + 16 <= L) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + p);
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            nbytes = 16;
        } else if (p < L) {
            nbytes = (int)(L - p);
            for (int i = 0; i < nbytes; ++i) q[i >> 2] |= (uint32_t)src[p + i] << (8 * (i & 3));
        }
        uint64_t b[4] = {0, 0, 0, 0};  // this lane's bits, LSB first
        uint32_t nb = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < nbytes) {
                const uint32_t e = lut[(q[j >> 2] >> (8 * (j & 3))) & 0xff];
                const uint64_t c = e >> 8;
                const uint32_t l = e & 0xff, wi = nb >> 6, sh = nb & 63;
                b[wi] |= c << sh;
                if (sh + l > 64 && wi < 3) b[wi + 1] |= c >> (64 - sh);
                nb += l;
            }
        }
        // exclusive scan of nb over the workgroup: shuffles inside a wave, the four wave totals through LDS
        uint32_t x = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if ((tid & 63) >= d) x += y;
        }
        if ((tid & 63) == 63) scan[wave] = x;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int i = 0; i < BLOCK / 64; ++i) {
            const uint32_t v = scan[i];
            if (i < wave) base += v;
            total += v;
        }
        const uint32_t mine = base + x - nb;
        const uint32_t cb = carry_bits, cw = carry_word;
        const uint32_t nwords = (cb + total + 31) >> 5;
        for (uint32_t i = tid; i <= nwords; i += BLOCK) words[i] = i == 0 ? cw : 0;
        __syncthreads();
        if (nb) {
            const uint32_t pos = cb + mine, w0 = pos >> 5, sh = pos & 31;
            // 256 bits shifted left by sh (< 32) -> up to 9 dwords
            uint32_t prev = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t d = (uint32_t)(b[i >> 1] >> (32 * (i & 1)));
                const uint32_t o = sh ? (d << sh) | (prev >> (32 - sh)) : d;
                if (o && 32u * i < nb + sh) atomicOr(&words[w0 + i], o);
                prev = d;
            }
            const uint32_t o = sh ? prev >> (32 - sh) : 0;
            if (o) atomicOr(&words[w0 + 8], o);
        }
        __syncthreads();
        const uint32_t full = (cb + total) >> 5;
        for (uint32_t i = tid; i < full; i += BLOCK) dst[out_word + i] = words[i];
        __syncthreads();
        if (tid == 0) {
            carry_word = words[full];
            carry_bits = (cb + total) & 31;
        }
        out_word += full;
        __syncthreads();
    }
    // 4. end of block, alignment, trailer
    if (tid == 0) {
        uint64_t acc = carry_word;
        int cnt = (int)carry_bits;
        const uint32_t e = lut[256];
        acc |= (uint64_t)(e >> 8) << cnt;
        cnt += (int)(e & 0xff);
        uint8_t* bytes = reinterpret_cast<uint8_t*>(dst) + (size_t)out_word * 4;
        int nby = 0;
        while (cnt > 0) {
            bytes[nby++] = (uint8_t)acc;
            acc >>= 8;
            cnt -= 8;
        }
        const uint32_t c = crc[blockIdx.x], isz = (uint32_t)L;
        for (int i = 0; i < 4; ++i) bytes[nby++] = (uint8_t)(c >> (8 * i));
        for (int i = 0; i < 4; ++i) bytes[nby++] = (uint8_t)(isz >> (8 * i));
        out_bytes[blockIdx.x] = out_word * 4 + (uint32_t)nby;
    }
}

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return -1;                                                              \
        }                                                                           \
    } while (0)
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// text[0..n): pageable host memory; piece: text bytes per member; crc[npieces]: CRC-32 of every piece (host made).
// out: room for npieces * out_stride bytes (host), sizes[npieces].  ms[0] = kernel only (HIP events, average of reps),
// ms[1] = H2D + kernel + D2H of the used bytes through pinned buffers, pieces in 3 groups on 3 streams (wall clock).
extern "C" int huff_probe(const uint8_t* text, int64_t n, int64_t piece, const uint32_t* crc, int reps, uint8_t* out,
                          int64_t out_stride, uint32_t* sizes, double* ms) {
    const int64_t np = (n + piece - 1) / piece;
    std::vector<int64_t> off((size_t)np + 1);
    for (int64_t i = 0; i <= np; ++i) off[(size_t)i] = i * piece < n ? i * piece : n;
    uint8_t *h_text, *h_out, *d_text, *d_out;
    int64_t* d_off;
    uint32_t *d_crc, *d_sizes, *h_sizes;
    CK(hipHostMalloc(&h_text, (size_t)n + 64));
    CK(hipHostMalloc(&h_out, (size_t)np * out_stride));
    CK(hipHostMalloc(&h_sizes, (size_t)np * 4));
    memcpy(h_text, text, (size_t)n);
    CK(hipMalloc(&d_text, (size_t)n + 64));
    CK(hipMalloc(&d_out, (size_t)np * out_stride));
    CK(hipMalloc(&d_off, (size_t)(np + 1) * 8));
    CK(hipMalloc(&d_crc, (size_t)np * 4));
    CK(hipMalloc(&d_sizes, (size_t)np * 4));
    CK(hipMemcpy(d_off, off.data(), (size_t)(np + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_crc, crc, (size_t)np * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_text, h_text, (size_t)n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // kernel only, text resident
    hipLaunchKernelGGL(huff_pieces, dim3((unsigned)np), dim3(BLOCK), 0, 0, d_text, d_off, d_crc, d_out, out_stride, d_sizes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(huff_pieces, dim3((unsigned)np), dim3(BLOCK), 0, 0, d_text, d_off, d_crc, d_out, out_stride, d_sizes);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float k = 0;
    CK(hipEventElapsedTime(&k, e0, e1));
    ms[0] = k / reps;
    // whole trip: pinned text -> device -> members -> pinned host, three groups of pieces on three streams
    const int G = 3;
    hipStream_t st[G];
    for (int g = 0; g < G; ++g) CK(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now();
        for (int g = 0; g < G; ++g) {
            const int64_t p0 = np * g / G, p1 = np * (g + 1) / G;
            if (p1 == p0) continue;
            CK(hipMemcpyAsync(d_text + off[(size_t)p0], h_text + off[(size_t)p0], (size_t)(off[(size_t)p1] - off[(size_t)p0]), hipMemcpyHostToDevice, st[g]));
            hipLaunchKernelGGL(huff_pieces, dim3((unsigned)(p1 - p0)), dim3(BLOCK), 0, st[g], d_text, d_off + p0, d_crc + p0,
                               d_out + p0 * out_stride, out_stride, d_sizes + p0);
            CK(hipMemcpyAsync(h_sizes + p0, d_sizes + p0, (size_t)(p1 - p0) * 4, hipMemcpyDeviceToHost, st[g]));
        }
        for (int g = 0; g < G; ++g) {  // sizes known -> fetch the used bytes of every member
            const int64_t p0 = np * g / G, p1 = np * (g + 1) / G;
            CK(hipStreamSynchronize(st[g]));
            for (int64_t i = p0; i < p1; ++i)
                CK(hipMemcpyAsync(h_out + i * out_stride, d_out + i * out_stride, h_sizes[i], hipMemcpyDeviceToHost, st[g]));
        }
        for (int g = 0; g < G; ++g) CK(hipStreamSynchronize(st[g]));
        const double dt = now() - t0;
        if (dt < best) best = dt;
    }
    ms[1] = best * 1e3;
    memcpy(sizes, h_sizes, (size_t)np * 4);
    for (int64_t i = 0; i < np; ++i) memcpy(out + i * out_stride, h_out + i * out_stride, h_sizes[i]);
    (void)hipFree(d_text); (void)hipFree(d_out); (void)hipFree(d_off); (void)hipFree(d_crc); (void)hipFree(d_sizes);
    (void)hipHostFree(h_text); (void)hipHostFree(h_out); (void)hipHostFree(h_sizes);
    return (int)np;
}
