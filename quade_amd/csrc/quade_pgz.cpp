// Parallel inflate of one gzip stream on host threads: see quade_pgz.h for what it replaces and how.
// Own DEFLATE decoder (RFC 1951) because the scheme needs three things no library offers together: start at a bit
// offset, stop at a chosen block boundary, and write 16-bit symbols over a window of markers.
#include "quade_pgz.h"

#include <zlib.h>  // crc32_combine

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

uint32_t qd_io_crc32(const uint8_t* p, size_t n);  // quade_io.cpp: libdeflate's when loaded, else zlib's

namespace qdpgz {
namespace {

constexpr size_t WIN = 32768;
constexpr int LB = 11, DB = 8;  // primary table widths (literal/length, distance)
constexpr uint32_t LMASK = (1u << LB) - 1, DMASK = (1u << DB) - 1;
// literal/length entries: flag | payload << 16 | extra or subtable bits << 8 | bits to drop
constexpr uint32_t F_LIT = 0x80000000u, F_LEN = 0x40000000u, F_EOB = 0x20000000u, F_SUB = 0x10000000u;
// distance entries: flag | base or subtable index << 15 | extra or subtable bits << 8 | bits to drop
constexpr uint32_t D_SUB = 0x80000000u, D_OK = 0x40000000u;
constexpr int LIT_CAP = (1 << LB) + 1024, DIST_CAP = (1 << DB) + 512;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
    uint32_t lit[LIT_CAP];
    uint32_t dist[DIST_CAP];
};

inline uint32_t bitrev(uint32_t v, int n) {  // the low n bits of v, reversed
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0f0fu) << 4) | ((v >> 4) & 0x0f0fu);
    v = ((v & 0x00ffu) << 8) | ((v >> 8) & 0x00ffu);
    return v >> (16 - n);
}

inline uint32_t lit_entry(int sym, int drop) {
    if (sym < 256) return F_LIT | ((uint32_t)sym << 16) | (uint32_t)drop;
    if (sym == 256) return F_EOB | (uint32_t)drop;
    if (sym > 285) return 0;  // 286, 287: in the code space, never valid in data
    return F_LEN | ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | (uint32_t)drop;
}
inline uint32_t dist_entry(int sym, int drop) {
    if (sym > 29) return 0;
    return D_OK | ((uint32_t)DIST_BASE[sym] << 15) | ((uint32_t)DIST_EXTRA[sym] << 8) | (uint32_t)drop;
}

// Canonical Huffman code -> lookup table indexed by the next PB bits of the stream (LSB first); longer codes go
// through subtables behind the primary entries.  Accepts what zlib's inflate_table accepts: an over-subscribed code
// is an error, an incomplete one is an error unless it is a single code of length 1 (or no code at all).
template <bool LITLEN>
bool build_table(const uint8_t* lens, int n, uint32_t* tab) {
    constexpr int PB = LITLEN ? LB : DB, CAP = LITLEN ? LIT_CAP : DIST_CAP, SHIFT = LITLEN ? 16 : 15;
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    int maxl = 15;
    while (maxl > 0 && !count[maxl]) --maxl;
    memset(tab, 0, sizeof(uint32_t) << PB);
    if (maxl == 0) return true;  // no symbol: every lookup is an error (a block of literals needs no distance code)
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && maxl != 1) return false;
    uint16_t offs[16], sorted[288];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;
    uint32_t code = 0;
    int at = 0, next_free = 1 << PB;
    // short codes: replicate over the primary table
    int l = 1;
    for (; l <= PB && l <= maxl; ++l) {
        for (int c = 0; c < count[l]; ++c, ++code) {
            const uint32_t e = LITLEN ? lit_entry(sorted[at], l) : dist_entry(sorted[at], l);
            ++at;
            for (uint32_t i = bitrev(code, l); i < (1u << PB); i += 1u << l) tab[i] = e;
        }
        code <<= 1;
    }
    if (maxl <= PB) return true;
    // long codes: codes that share their first PB bits are consecutive in canonical order and their lengths do not
    // decrease, so a subtable is sized by the last code that falls into it
    const int first_long = at;
    uint8_t sub_bits[1 << PB];
    memset(sub_bits, 0, sizeof sub_bits);
    {
        uint32_t c2 = code;
        int a2 = at;
        for (int l2 = l; l2 <= maxl; ++l2) {
            for (int c = 0; c < count[l2]; ++c, ++c2, ++a2) sub_bits[bitrev(c2, l2) & ((1u << PB) - 1)] = (uint8_t)(l2 - PB);
            c2 <<= 1;
        }
    }
    (void)first_long;
    for (; l <= maxl; ++l) {
        for (int c = 0; c < count[l]; ++c, ++code) {
            const uint32_t rev = bitrev(code, l), prefix = rev & ((1u << PB) - 1);
            uint32_t p = tab[prefix];
            const uint32_t sb = sub_bits[prefix];
            if (!(p & (LITLEN ? F_SUB : D_SUB))) {
                if (next_free + (1 << sb) > CAP) return false;
                p = (LITLEN ? F_SUB : D_SUB) | ((uint32_t)next_free << SHIFT) | (sb << 8) | (uint32_t)PB;
                memset(tab + next_free, 0, sizeof(uint32_t) << sb);
                tab[prefix] = p;
                next_free += 1 << sb;
            }
            const uint32_t base = (p >> SHIFT) & (LITLEN ? 0xfffu : 0x7fffu);
            const uint32_t e = LITLEN ? lit_entry(sorted[at], l - PB) : dist_entry(sorted[at], l - PB);
            ++at;
            for (uint32_t i = rev >> PB; i < (1u << sb); i += 1u << (l - PB)) tab[base + i] = e;
        }
        code <<= 1;
    }
    return true;
}

const Tables& fixed_tables() {
    static const Tables* T = [] {
        Tables* t = new Tables();
        uint8_t l[288];
        for (int s = 0; s < 144; ++s) l[s] = 8;
        for (int s = 144; s < 256; ++s) l[s] = 9;
        for (int s = 256; s < 280; ++s) l[s] = 7;
        for (int s = 280; s < 288; ++s) l[s] = 8;
        build_table<true>(l, 288, t->lit);
        uint8_t d[32];
        for (int s = 0; s < 32; ++s) d[s] = 5;
        build_table<false>(d, 32, t->dist);
        return t;
    }();
    return *T;
}

// ---- bit reader: LSB first, 64-bit buffer ---------------------------------------------------------------------
struct Bits {
    const uint8_t* base = nullptr;
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    uint64_t buf = 0;
    int cnt = 0;  // bits of buf that count as read from the input (buf may hold valid bits beyond them)
    void seek(const uint8_t* b, size_t size, int64_t bitpos) {
        base = b;
        end = b + size;
        p = b + (bitpos >> 3);
        if (p > end) p = end;
        buf = 0;
        cnt = 0;
        fill_safe();
        drop((int)(bitpos & 7));
    }
    int64_t pos() const { return (int64_t)(p - base) * 8 - cnt; }
    void fill_safe() {
        while (cnt < 56 && p < end) {  // (not <=: fill_fast shifts by cnt, which must stay below 64)
            buf |= (uint64_t)*p++ << cnt;
            cnt += 8;
        }
    }
    void fill_fast() {  // needs p + 8 <= end; afterwards 56 <= cnt <= 63
        uint64_t w;
        memcpy(&w, p, 8);
        buf |= w << cnt;
        p += (63 - cnt) >> 3;
        cnt |= 56;
    }
    void drop(int n) {
        buf >>= n;
        cnt -= n;
    }
    // back to whole bytes: the bits up to the next byte boundary are skipped, fetched bytes are un-read
    void to_bytes() {
        drop(cnt & 7);
        p -= cnt >> 3;
        buf = 0;
        cnt = 0;
    }
};

enum Rc { RC_BLOCK_END = 0, RC_NEED_OUT = 1, RC_ERR_DATA = 2, RC_ERR_TRUNC = 3 };

template <class T>
inline void copy_match(T* out, uint32_t dist, uint32_t len) {  // may write up to 16 bytes past out + len
    constexpr uint32_t W = 16 / sizeof(T);
    const T* src = out - dist;
    T* const end = out + len;
    if (dist >= W) {
        do {
            memcpy(out, src, 16);
            out += W;
            src += W;
        } while (out < end);
    } else if (dist == 1) {
        T pat[W];
        for (uint32_t i = 0; i < W; ++i) pat[i] = *src;
        do {
            memcpy(out, pat, 16);
            out += W;
        } while (out < end);
    } else {
        do {
            *out++ = *src++;
        } while (out < end);
    }
}

// The symbols of one block, up to its end-of-block code.  `out` may be written up to `hard_end`; RC_NEED_OUT asks
// for a larger buffer (the call is then repeated: nothing is lost, the loop only stops between symbols).
// `lowest`: the first element a match may reach back to.
template <class T>
Rc block_body(Bits& b, const Tables& t, T*& outp, T* hard_end, const T* lowest) {
    T* out = outp;
    T* const fast_lim = hard_end - 320;  // a round of the fast loop writes at most 2 literals + 258 + 15
    Rc rc = RC_NEED_OUT;
    if (b.end - b.base >= 16) {
        const uint8_t* const in_lim = b.end - 16;
        while (b.p <= in_lim && out < fast_lim) {
            b.fill_fast();
            uint32_t e = t.lit[b.buf & LMASK];
            if (e & F_LIT) {
                *out++ = (T)((e >> 16) & 0xff);
                b.drop((int)(e & 31));
                e = t.lit[b.buf & LMASK];
                if (e & F_LIT) {
                    *out++ = (T)((e >> 16) & 0xff);
                    b.drop((int)(e & 31));
                    e = t.lit[b.buf & LMASK];
                    if (e & F_LIT) {
                        *out++ = (T)((e >> 16) & 0xff);
                        b.drop((int)(e & 31));
                        continue;
                    }
                }
            }
            if (e & F_SUB) {
                b.drop(LB);
                e = t.lit[((e >> 16) & 0xfff) + ((uint32_t)b.buf & ((1u << ((e >> 8) & 15)) - 1))];
                if (e & F_LIT) {
                    *out++ = (T)((e >> 16) & 0xff);
                    b.drop((int)(e & 31));
                    continue;
                }
            }
            if (!(e & F_LEN)) {
                outp = out;
                if (e & F_EOB) {
                    b.drop((int)(e & 31));
                    return RC_BLOCK_END;
                }
                return RC_ERR_DATA;
            }
            uint32_t cl = e & 31, eb = (e >> 8) & 15;
            const uint32_t len = ((e >> 16) & 0x1ff) + ((uint32_t)(b.buf >> cl) & ((1u << eb) - 1));
            b.drop((int)(cl + eb));
            b.fill_fast();
            uint32_t d = t.dist[b.buf & DMASK];
            if (d & D_SUB) {
                b.drop(DB);
                d = t.dist[((d >> 15) & 0x7fff) + ((uint32_t)b.buf & ((1u << ((d >> 8) & 15)) - 1))];
            }
            if (!(d & D_OK)) {
                outp = out;
                return RC_ERR_DATA;
            }
            cl = d & 31;
            eb = (d >> 8) & 15;
            const uint32_t dist = ((d >> 15) & 0x7fff) + ((uint32_t)(b.buf >> cl) & ((1u << eb) - 1));
            b.drop((int)(cl + eb));
            if ((size_t)(out - lowest) < dist) {
                outp = out;
                return RC_ERR_DATA;
            }
            copy_match(out, dist, len);
            out += len;
        }
    }
    // the careful loop: the last bytes of the input, and the last elements of the buffer
    for (;;) {
        if (hard_end - out < 258 + 16) {
            rc = RC_NEED_OUT;
            break;
        }
        b.fill_safe();
        uint32_t e = t.lit[b.buf & LMASK];
        if (e & F_SUB) {
            b.drop(LB);
            e = t.lit[((e >> 16) & 0xfff) + ((uint32_t)b.buf & ((1u << ((e >> 8) & 15)) - 1))];
        }
        if (e & F_LIT) {
            b.drop((int)(e & 31));
            if (b.cnt < 0) {
                rc = RC_ERR_TRUNC;
                break;
            }
            *out++ = (T)((e >> 16) & 0xff);
            continue;
        }
        if (!(e & F_LEN)) {
            if (e & F_EOB) {
                b.drop((int)(e & 31));
                rc = b.cnt < 0 ? RC_ERR_TRUNC : RC_BLOCK_END;
            } else {
                rc = b.p >= b.end && b.cnt < 15 ? RC_ERR_TRUNC : RC_ERR_DATA;
            }
            break;
        }
        uint32_t cl = e & 31, eb = (e >> 8) & 15;
        const uint32_t len = ((e >> 16) & 0x1ff) + ((uint32_t)(b.buf >> cl) & ((1u << eb) - 1));
        b.drop((int)(cl + eb));
        b.fill_safe();
        uint32_t d = t.dist[b.buf & DMASK];
        if (d & D_SUB) {
            b.drop(DB);
            d = t.dist[((d >> 15) & 0x7fff) + ((uint32_t)b.buf & ((1u << ((d >> 8) & 15)) - 1))];
        }
        if (!(d & D_OK)) {
            rc = b.p >= b.end && b.cnt < 15 ? RC_ERR_TRUNC : RC_ERR_DATA;
            break;
        }
        cl = d & 31;
        eb = (d >> 8) & 15;
        const uint32_t dist = ((d >> 15) & 0x7fff) + ((uint32_t)(b.buf >> cl) & ((1u << eb) - 1));
        b.drop((int)(cl + eb));
        if (b.cnt < 0) {
            rc = RC_ERR_TRUNC;
            break;
        }
        if ((size_t)(out - lowest) < dist) {
            rc = RC_ERR_DATA;
            break;
        }
        const T* src = out - dist;
        for (uint32_t i = 0; i < len; ++i) out[i] = src[i];
        out += len;
    }
    outp = out;
    return rc;
}

// header of a dynamic-Huffman block (behind its 3 type bits) -> tables
Rc read_dynamic(Bits& b, Tables& t) {
    b.fill_safe();
    if (b.cnt < 14) return RC_ERR_TRUNC;
    const int hlit = (int)(b.buf & 31) + 257, hdist = (int)((b.buf >> 5) & 31) + 1, hclen = (int)((b.buf >> 10) & 15) + 4;
    b.drop(14);
    if (hlit > 286 || hdist > 30) return RC_ERR_DATA;
    uint8_t pl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
        if (b.cnt < 3) b.fill_safe();
        if (b.cnt < 3) return RC_ERR_TRUNC;
        pl[PRE_ORDER[i]] = (uint8_t)(b.buf & 7);
        b.drop(3);
    }
    // the code-length code: 7-bit lookup, must be complete (zlib: "invalid code lengths set")
    uint16_t pre[128];
    {
        int count[8] = {0};
        for (int s = 0; s < 19; ++s) ++count[pl[s]];
        int left = 1;
        for (int l = 1; l <= 7; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return RC_ERR_DATA;
        }
        if (left != 0) return RC_ERR_DATA;
        memset(pre, 0, sizeof pre);
        uint32_t code = 0;
        for (int l = 1; l <= 7; ++l) {
            for (int s = 0; s < 19; ++s)
                if (pl[s] == l) {
                    for (uint32_t i = bitrev(code, l); i < 128; i += 1u << l) pre[i] = (uint16_t)(s | (l << 8));
                    ++code;
                }
            code <<= 1;
        }
    }
    uint8_t lens[286 + 30 + 138];
    const int total = hlit + hdist;
    int i = 0;
    while (i < total) {
        b.fill_safe();
        const uint32_t e = pre[b.buf & 127];
        const int sym = (int)(e & 0xff), l = (int)(e >> 8);
        if (!l) return RC_ERR_DATA;
        b.drop(l);
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
        } else {
            int rep, val = 0;
            if (sym == 16) {
                if (i == 0) return RC_ERR_DATA;
                val = lens[i - 1];
                rep = 3 + (int)(b.buf & 3);
                b.drop(2);
            } else if (sym == 17) {
                rep = 3 + (int)(b.buf & 7);
                b.drop(3);
            } else {
                rep = 11 + (int)(b.buf & 127);
                b.drop(7);
            }
            if (i + rep > total) return RC_ERR_DATA;
            memset(lens + i, val, (size_t)rep);
            i += rep;
        }
        if (b.cnt < 0) return RC_ERR_TRUNC;
    }
    if (lens[256] == 0) return RC_ERR_DATA;  // zlib: "missing end-of-block"
    if (!build_table<true>(lens, hlit, t.lit)) return RC_ERR_DATA;
    if (!build_table<false>(lens + hlit, hdist, t.dist)) return RC_ERR_DATA;
    return RC_BLOCK_END;
}

// gzip member header at p (RFC 1952) -> its length; 0 = not a header, -1 = cut off by the end of the input
int64_t gzip_header_len(const uint8_t* p, size_t n) {
    if (n < 10) return (n >= 1 && p[0] != 0x1f) || (n >= 2 && p[1] != 0x8b) ? 0 : -1;
    if (!looks_like_gzip(p, n)) return 0;
    const uint8_t flg = p[3];
    size_t o = 10;
    if (flg & 4) {
        if (o + 2 > n) return -1;
        o += 2 + ((size_t)p[o] | ((size_t)p[o + 1] << 8));
        if (o > n) return -1;
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
            while (o < n && p[o]) ++o;
            if (o >= n) return -1;
            ++o;
        }
    if (flg & 2) o += 2;
    return o > n ? -1 : (int64_t)o;
}

// ---- buffers: kept between uses (a fresh 30 MB buffer per chunk is thousands of page faults) ------------------
struct BufPool {
    std::mutex m;
    std::vector<std::pair<void*, size_t>> v;
    size_t bytes = 0;
};
constexpr size_t BUF_KEEP = 48, BUF_KEEP_BYTES = (size_t)2 << 30;
BufPool& buf_pool() {  // never destroyed (pool threads may still return buffers while the process exits) ...
    static BufPool* P = [] {
        BufPool* p = new BufPool();
        atexit([] {  // ... but its idle buffers go back to the allocator (leak checkers look)
            BufPool& b = buf_pool();
            std::lock_guard<std::mutex> g(b.m);
            for (auto& e : b.v) free(e.first);
            b.v.clear();
            b.bytes = 0;
        });
        return p;
    }();
    return *P;
}

void* buf_get(size_t bytes, size_t* cap) {
    BufPool& b = buf_pool();
    {
        std::lock_guard<std::mutex> g(b.m);
        size_t best = b.v.size();
        for (size_t i = 0; i < b.v.size(); ++i)
            if (b.v[i].second >= bytes && b.v[i].second <= 4 * bytes + (1u << 20) && (best == b.v.size() || b.v[i].second < b.v[best].second))
                best = i;
        if (best != b.v.size()) {
            void* p = b.v[best].first;
            *cap = b.v[best].second;
            b.bytes -= *cap;
            b.v.erase(b.v.begin() + (long)best);
            return p;
        }
    }
    *cap = bytes;
    return malloc(bytes);
}
void buf_put(void* p, size_t cap) {
    if (!p) return;
    BufPool& b = buf_pool();
    {
        std::lock_guard<std::mutex> g(b.m);
        if (b.v.size() < BUF_KEEP && b.bytes + cap <= BUF_KEEP_BYTES) {
            b.v.emplace_back(p, cap);
            b.bytes += cap;
            return;
        }
    }
    free(p);
}

struct MemberEnd {
    size_t out_pos;  // output elements of this run before the member ended
    uint32_t crc, isize;
};

enum Stop { STOP_SYNC = 0, STOP_EOF = 1, STOP_ERROR = 2 };

// one run of the decoder: [WIN elements of history][n elements of output]
template <class T>
struct Run {
    T* buf = nullptr;
    size_t cap_bytes = 0, cap = 0;  // elements
    size_t n = 0;
    std::vector<MemberEnd> ends;
    int64_t start_bit = 0, end_bit = 0;
    Stop stop = STOP_ERROR;
    std::string err;
    ~Run() { buf_put(buf, cap_bytes); }
    bool reserve(size_t elems) {  // keeps the contents
        if (elems <= cap) return true;
        size_t nb = 0;
        T* q = (T*)buf_get(elems * sizeof(T), &nb);
        if (!q) return false;
        if (buf) memcpy(q, buf, (WIN + n) * sizeof(T));
        buf_put(buf, cap_bytes);
        buf = q;
        cap_bytes = nb;
        cap = nb / sizeof(T);
        return true;
    }
};

const char* const ERR_TRUNC = "compressed file ended before the end-of-stream marker";
const char* const ERR_DATA = "not a valid gzip stream (invalid deflate data)";

// Inflates from `start_bit` (a block header; or, with header_first, the byte offset * 8 of a gzip member header)
// until the first dynamic block that begins at or beyond `limit_bit` (the block at start_bit itself never
// stops the run) or the end of the file.  `valid`: how many history elements matches may reach at the start.
// max_out: give up (as an error) beyond this many output elements (trial decodes of candidates).
template <class T>
void inflate_run(const uint8_t* data, size_t size, int64_t start_bit, bool header_first, int64_t limit_bit, size_t valid, Run<T>& r,
                 size_t max_out = ~(size_t)0, int max_blocks = 0x7fffffff) {
    r.start_bit = start_bit;
    r.stop = STOP_ERROR;
    Bits b;
    size_t lowest = WIN - valid;  // index into r.buf
    if (header_first) {
        const size_t at = (size_t)(start_bit >> 3);
        const int64_t h = gzip_header_len(data + at, size - at);
        if (h <= 0) {
            r.err = h < 0 ? ERR_TRUNC : "not a valid gzip stream (no gzip header where a member should begin)";
            return;
        }
        start_bit = (int64_t)(at + (size_t)h) * 8;
        lowest = WIN + r.n;
    }
    b.seek(data, size, start_bit);
    std::unique_ptr<Tables> dyn(new Tables());
    bool first = true;
    for (int blocks = 0;; ++blocks) {
        b.fill_safe();
        if (b.cnt < 3) {
            r.err = ERR_TRUNC;
            return;
        }
        const uint32_t h = (uint32_t)b.buf & 7;
        if (!first && (((h & 6) == 4 && b.pos() >= limit_bit) || blocks >= max_blocks)) {
            r.end_bit = b.pos();
            r.stop = STOP_SYNC;
            return;
        }
        first = false;
        b.drop(3);
        const bool bfinal = h & 1;
        const int type = (int)(h >> 1);
        if (type == 0) {
            b.to_bytes();
            if (b.end - b.p < 4) {
                r.err = ERR_TRUNC;
                return;
            }
            const uint32_t len = b.p[0] | ((uint32_t)b.p[1] << 8), nlen = b.p[2] | ((uint32_t)b.p[3] << 8);
            if ((len ^ nlen) != 0xffff) {
                r.err = "not a valid gzip stream (stored block lengths)";
                return;
            }
            b.p += 4;
            if ((size_t)(b.end - b.p) < len) {
                r.err = ERR_TRUNC;
                return;
            }
            if (r.n + len > max_out || !r.reserve(WIN + r.n + len + 1024)) {
                r.err = "out of memory inflating";
                return;
            }
            T* o = r.buf + WIN + r.n;
            for (uint32_t i = 0; i < len; ++i) o[i] = (T)b.p[i];
            b.p += len;
            r.n += len;
        } else if (type == 3) {
            r.err = "not a valid gzip stream (reserved block type)";
            return;
        } else {
            const Tables* t = &fixed_tables();
            if (type == 2) {
                const Rc rc = read_dynamic(b, *dyn);
                if (rc != RC_BLOCK_END) {
                    r.err = rc == RC_ERR_TRUNC ? ERR_TRUNC : "not a valid gzip stream (invalid code lengths)";
                    return;
                }
                t = dyn.get();
            }
            for (;;) {
                if (r.cap < WIN + r.n + 4096 && !r.reserve(std::max<size_t>(2 * r.cap, WIN + r.n + (1u << 20)))) {
                    r.err = "out of memory inflating";
                    return;
                }
                T* out = r.buf + WIN + r.n;
                const Rc rc = block_body<T>(b, *t, out, r.buf + r.cap, r.buf + lowest);
                r.n = (size_t)(out - (r.buf + WIN));
                if (rc == RC_BLOCK_END) break;
                if (rc == RC_NEED_OUT) {
                    if (r.n > max_out || !r.reserve(r.cap + r.cap / 2 + (1u << 20))) {
                        r.err = "out of memory inflating";
                        return;
                    }
                    continue;
                }
                r.err = rc == RC_ERR_TRUNC ? ERR_TRUNC : ERR_DATA;
                return;
            }
        }
        if (bfinal) {  // member trailer, then the end of the file or the next member
            b.to_bytes();
            if (b.end - b.p < 8) {
                r.err = ERR_TRUNC;
                return;
            }
            MemberEnd me;
            me.out_pos = r.n;
            me.crc = b.p[0] | ((uint32_t)b.p[1] << 8) | ((uint32_t)b.p[2] << 16) | ((uint32_t)b.p[3] << 24);
            me.isize = b.p[4] | ((uint32_t)b.p[5] << 8) | ((uint32_t)b.p[6] << 16) | ((uint32_t)b.p[7] << 24);
            r.ends.push_back(me);
            b.p += 8;
            bool zeros = true;  // zero padding behind the last member is tolerated (as by gzip itself)
            for (const uint8_t* q = b.p; zeros && q < b.end; ++q) zeros = *q == 0;
            if (zeros) {
                r.end_bit = (int64_t)size * 8;
                r.stop = STOP_EOF;
                return;
            }
            const int64_t hl = gzip_header_len(b.p, (size_t)(b.end - b.p));
            if (hl <= 0) {
                r.err = hl < 0 ? ERR_TRUNC : "not a valid gzip stream (garbage behind a gzip member)";
                return;
            }
            b.p += hl;
            lowest = WIN + r.n;  // nothing before the member may be referenced
        }
    }
}

// ---- looking for a block start ---------------------------------------------------------------------------------
inline bool texty(uint32_t c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// complete code-length code?  w = the stream from the candidate's first bit on (>= 74 + bits), hclen + 4 lengths
inline bool precode_complete(uint64_t lo, uint64_t hi, int n) {
    static const uint8_t K[8] = {0, 64, 32, 16, 8, 4, 2, 1};
    uint32_t sum = 0;
    // lengths start at bit 17: fifteen of them lie inside `lo`, the rest straddle into `hi`
    uint64_t w = lo >> 17;
    const int a = n < 15 ? n : 15;
    for (int i = 0; i < a; ++i) {
        sum += K[w & 7];
        w >>= 3;
    }
    if (n > 15) {
        w = (lo >> 62) | (hi << 2);
        for (int i = 15; i < n; ++i) {
            sum += K[w & 7];
            w >>= 3;
        }
    }
    return sum == 128;
}

// Does a dynamic block begin at bit `pos`?  The whole block must decode (to text, into the marker window) and be
// followed by something that parses as a block header -- behind a final block: the member's trailer, then the end
// of the file or a gzip header, then a block header.
bool validate_start(const uint8_t* data, size_t size, int64_t pos, Run<uint16_t>& scratch) {
    scratch.n = 0;
    scratch.ends.clear();
    scratch.err.clear();
    inflate_run<uint16_t>(data, size, pos, false, 0, WIN, scratch, (size_t)8 << 20, 1);
    if (scratch.stop != STOP_SYNC && scratch.stop != STOP_EOF) return false;
    const uint16_t* s = scratch.buf + WIN;
    for (size_t i = 0; i < scratch.n; ++i)
        if (s[i] < 0x8000 && !texty(s[i])) return false;
    if (scratch.stop == STOP_EOF) return true;
    // what follows must look like a block again
    Bits b;
    b.seek(data, size, scratch.end_bit);
    if (b.cnt < 3) return false;
    const uint32_t h = (uint32_t)b.buf & 7;
    b.drop(3);
    const int type = (int)(h >> 1);
    if (type == 3) return false;
    if (type == 2) {
        std::unique_ptr<Tables> t(new Tables());
        return read_dynamic(b, *t) == RC_BLOCK_END;
    }
    if (type == 0) {
        b.to_bytes();
        if (b.end - b.p < 4) return false;
        const uint32_t len = b.p[0] | ((uint32_t)b.p[1] << 8), nlen = b.p[2] | ((uint32_t)b.p[3] << 8);
        return (len ^ nlen) == 0xffff;
    }
    return true;
}

// first bit position in [from_bit, to_bit) at which a dynamic block (validated) begins; -1: none
int64_t find_start(const uint8_t* data, size_t size, int64_t from_bit, int64_t to_bit, Run<uint16_t>& scratch, int64_t* tried) {
    const int64_t last = std::min<int64_t>(to_bit, (int64_t)size * 8 - 100);
    int64_t pos = from_bit;
    while (pos < last) {
        const size_t byte = (size_t)(pos >> 3);
        if (byte + 24 > size) break;
        uint64_t a, c;
        memcpy(&a, data + byte, 8);
        memcpy(&c, data + byte + 8, 8);
        for (int k = (int)(pos & 7); k < 8 && pos < last; ++k, ++pos) {
            const uint64_t lo = k ? (a >> k) | (c << (64 - k)) : a, hi = c >> k;
            // BTYPE = 2 (bits 1..2 = 0 1, LSB first), final or not; HLIT <= 29, HDIST <= 29
            if ((lo & 6) != 4) continue;
            if (((lo >> 3) & 31) > 29 || ((lo >> 8) & 31) > 29) continue;
            if (!precode_complete(lo, hi, (int)((lo >> 13) & 15) + 4)) continue;
            if (validate_start(data, size, pos, scratch)) {
                if (tried) *tried += pos - from_bit + 1;
                return pos;
            }
        }
    }
    if (tried) *tried += std::max<int64_t>(0, pos - from_bit);
    return -1;
}

// ---- symbols -> bytes ----------------------------------------------------------------------------------------
// window: the WIN bytes before the run; marker 0x8000 + i = window[i].  Returns false when a marker reaches below
// `lowest_ok` (a match went back beyond the start of its member: the stream is damaged).
bool translate(const uint16_t* s, size_t n, const uint8_t* window, uint8_t* out, size_t lowest_ok) {
    uint32_t min_marker = 0xffff;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        uint32_t any = 0;
        for (int k = 0; k < 16; ++k) any |= s[i + k];
        if (!(any & 0x8000)) {
            for (int k = 0; k < 16; ++k) out[i + k] = (uint8_t)s[i + k];
        } else {
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = s[i + k];
                if (v & 0x8000) {
                    out[i + k] = window[v & 0x7fff];
                    min_marker = std::min(min_marker, v);
                } else {
                    out[i + k] = (uint8_t)v;
                }
            }
        }
    }
    for (; i < n; ++i) {
        const uint32_t v = s[i];
        if (v & 0x8000) {
            out[i] = window[v & 0x7fff];
            min_marker = std::min(min_marker, v);
        } else {
            out[i] = (uint8_t)v;
        }
    }
    return min_marker == 0xffff || (min_marker & 0x7fff) >= lowest_ok;
}

struct Segment {  // a stretch of a piece that belongs to one member
    size_t len;
    uint32_t crc;
    bool ends_member;
    uint32_t want_crc, want_isize;
};

struct Piece {
    // speculative (16-bit) or exact (8-bit) run
    std::unique_ptr<Run<uint16_t>> r16;
    std::unique_ptr<Run<uint8_t>> r8;
    bool found = false;            // speculative: a start was found
    int64_t searched = 0;
    std::shared_ptr<Text> text;    // the translated (or directly inflated) bytes
    std::vector<Segment> segs;
    bool bad_reach = false;
    bool done1 = false, done2 = false;  // pass 1 (inflate) / pass 2 (translate + CRC) finished
    size_t index = 0;
};

void crc_segments(const uint8_t* text, size_t n, const std::vector<MemberEnd>& ends, std::vector<Segment>& segs) {
    size_t at = 0;
    for (const MemberEnd& e : ends) {
        Segment sg{e.out_pos - at, qd_io_crc32(text + at, e.out_pos - at), true, e.crc, e.isize};
        segs.push_back(sg);
        at = e.out_pos;
    }
    if (at < n) segs.push_back(Segment{n - at, qd_io_crc32(text + at, n - at), false, 0, 0});
}

}  // namespace

struct State {
    const uint8_t* data = nullptr;
    size_t size = 0;
    Options opt;
    Submit submit;
    std::mutex m;
    std::condition_variable cv;
    int outstanding = 0;  // jobs on the pool
    std::string err;
    bool finished = false;
    Stats st;
    // coordinator state (only the thread that calls next() touches it)
    size_t n_chunks = 0, next_chunk = 0;
    std::deque<std::shared_ptr<Piece>> pass1;  // submitted, not yet accepted (file order)
    std::deque<std::shared_ptr<Piece>> pass2;  // accepted, being translated / ready (file order)
    int64_t expect_bit = 0;                    // where the next accepted run must begin
    bool expect_header = true;                 // ... and whether that is a member header
    bool eof = false;
    uint8_t window[WIN];
    size_t hist = 0;                           // bytes of `window` (its tail) that belong to the current member
    uint32_t crc = 0;                          // running CRC-32 / length of the current member
    uint64_t member_len = 0;
    size_t consumed_bytes = 0;
};

namespace {

void job_done(const std::shared_ptr<State>& s, const std::function<void()>& mark) {
    std::lock_guard<std::mutex> g(s->m);
    mark();
    --s->outstanding;
    s->cv.notify_all();
}

void fail(State& s, const std::string& why) {
    if (s.err.empty()) s.err = why;
}

// the window after a run of n bytes whose text is t (already resolved)
void advance_window(State& s, const uint8_t* t, size_t n, const std::vector<MemberEnd>& ends) {
    if (n >= WIN) {
        memcpy(s.window, t + n - WIN, WIN);
    } else if (n) {
        memmove(s.window, s.window + n, WIN - n);
        memcpy(s.window + WIN - n, t, n);
    }
    const size_t since = ends.empty() ? s.hist + n : n - ends.back().out_pos;
    s.hist = std::min(since, WIN);
}

}  // namespace

Gunzip::Gunzip(const uint8_t* data, size_t size, const Options& opt, Submit submit) : s_(std::make_shared<State>()) {
    s_->data = data;
    s_->size = size;
    s_->opt = opt;
    if (s_->opt.chunk_bytes < (64u << 10)) s_->opt.chunk_bytes = 64u << 10;
    if (s_->opt.in_flight < 1) s_->opt.in_flight = 1;
    s_->submit = std::move(submit);
    s_->n_chunks = size ? (size + s_->opt.chunk_bytes - 1) / s_->opt.chunk_bytes : 0;
    if (size == 0) s_->eof = true;  // an empty file is an empty stream (as the reader always treated it)
}

Gunzip::~Gunzip() {
    std::unique_lock<std::mutex> g(s_->m);
    s_->cv.wait(g, [this] { return s_->outstanding == 0; });
}

const std::string& Gunzip::error() const { return s_->err; }
Stats Gunzip::stats() const { return s_->st; }
size_t Gunzip::consumed() const { return s_->consumed_bytes; }

int Gunzip::next(std::shared_ptr<Text>* out) {
    State& s = *s_;
    std::shared_ptr<State> keep = s_;
    if (!s.err.empty()) return -1;
    const size_t C = s.opt.chunk_bytes;
    for (;;) {
        // 1. keep the pool busy: speculative inflate of the next chunks
        while (!s.eof && s.next_chunk < s.n_chunks && (int)(s.pass1.size() + s.pass2.size()) < s.opt.in_flight) {
            std::shared_ptr<Piece> pc = std::make_shared<Piece>();
            pc->index = s.next_chunk++;
            s.pass1.push_back(pc);
            if (pc->index == 0) continue;  // the first chunk has a known (empty) window: inflated below, exactly
            const int64_t from = (int64_t)(pc->index * C) * 8, to = (int64_t)std::min(s.size, (pc->index + 1) * C) * 8;
            const bool last = pc->index + 1 == s.n_chunks;
            {
                std::lock_guard<std::mutex> g(s.m);
                ++s.outstanding;
            }
            s.submit([keep, pc, from, to, last] {
                State& st = *keep;
                pc->r16.reset(new Run<uint16_t>());
                Run<uint16_t>& r = *pc->r16;
                const size_t want = WIN + (size_t)((to - from) / 8) * 4 + (1u << 20);
                if (r.reserve(want)) {
                    for (size_t i = 0; i < WIN; ++i) r.buf[i] = (uint16_t)(0x8000 + i);
                    const int64_t at = find_start(st.data, st.size, from, to, r, &pc->searched);
                    if (at >= 0) {
                        pc->found = true;
                        r.n = 0;
                        r.ends.clear();
                        r.err.clear();
                        inflate_run<uint16_t>(st.data, st.size, at, false, last ? INT64_MAX : to, WIN, r);
                    }
                }
                job_done(keep, [&] { pc->done1 = true; });
            });
        }
        // 2. something to deliver?
        if (!s.pass2.empty()) {
            std::shared_ptr<Piece> pc = s.pass2.front();
            bool ready;
            {
                std::lock_guard<std::mutex> g(s.m);
                ready = pc->done2;
            }
            // deliver when ready; when nothing else can be done meanwhile, wait for it
            const bool can_accept = !s.eof && !s.pass1.empty();
            if (!ready && !can_accept) {
                std::unique_lock<std::mutex> g(s.m);
                s.cv.wait(g, [&] { return pc->done2; });
                ready = true;
            }
            if (ready) {
                s.pass2.pop_front();
                if (pc->bad_reach) {
                    fail(s, "not a valid gzip stream (a match reaches back beyond the start of its member)");
                    return -1;
                }
                for (const Segment& sg : pc->segs) {
                    s.crc = s.member_len ? (uint32_t)crc32_combine(s.crc, sg.crc, (z_off_t)sg.len) : sg.crc;
                    s.member_len += sg.len;
                    if (sg.ends_member) {
                        if (s.crc != sg.want_crc || (uint32_t)s.member_len != sg.want_isize) {
                            fail(s, "not a valid gzip stream (CRC-32 or length of a member does not match its trailer)");
                            return -1;
                        }
                        ++s.st.members;
                        s.crc = 0;
                        s.member_len = 0;
                    }
                }
                ++s.st.chunks;
                if (pc->text && pc->text->len) {
                    *out = pc->text;
                    return 1;
                }
                continue;
            }
        }
        if (s.eof && s.pass2.empty()) {
            // runs that were still speculating beyond the end of the stream are simply dropped
            s.pass1.clear();
            if (s.member_len != 0) {
                fail(s, ERR_TRUNC);
                return -1;
            }
            s.finished = true;
            return 0;
        }
        if (s.pass1.empty()) {
            if (s.next_chunk >= s.n_chunks && !s.eof) {  // every chunk was used up and the stream has not ended
                fail(s, ERR_TRUNC);
                return -1;
            }
            continue;
        }
        // 3. accept the next chunk, in file order
        std::shared_ptr<Piece> pc = s.pass1.front();
        const int64_t chunk_end_bit = (int64_t)std::min(s.size, (pc->index + 1) * C) * 8;
        const bool last = pc->index + 1 == s.n_chunks;
        if (pc->index != 0) {
            std::unique_lock<std::mutex> g(s.m);
            // wait for its speculative run -- unless a translated piece becomes ready first (deliver that)
            s.cv.wait(g, [&] { return pc->done1 || (!s.pass2.empty() && s.pass2.front()->done2); });
            if (!pc->done1) continue;
        }
        s.pass1.pop_front();
        s.st.search_bits += pc->searched;
        if (pc->index != 0 && !last && chunk_end_bit <= s.expect_bit) continue;  // an earlier run already went beyond this chunk
        Run<uint16_t>* r16 = pc->r16.get();
        if (pc->index != 0 && pc->found && !s.expect_header && r16->start_bit == s.expect_bit) {
            // proven: it began exactly where its predecessor stopped
            if (r16->stop == STOP_ERROR) {
                fail(s, r16->err);
                return -1;
            }
            ++s.st.parallel;
            // its last WIN bytes are resolved here (the serial step), everything else by a pool job
            std::shared_ptr<std::vector<uint8_t>> win = std::make_shared<std::vector<uint8_t>>(s.window, s.window + WIN);
            const size_t lowest_ok = WIN - s.hist;
            {
                const size_t n = r16->n, tail = std::min(n, WIN);
                uint8_t tmp[WIN];
                translate(r16->buf + WIN + n - tail, tail, win->data(), tmp, 0);
                advance_window(s, tmp, tail, std::vector<MemberEnd>());  // (n >= WIN: hist = WIN either way)
                if (!r16->ends.empty()) s.hist = std::min(n - r16->ends.back().out_pos, WIN);
            }
            s.expect_bit = r16->end_bit;
            s.expect_header = false;
            s.consumed_bytes = (size_t)(r16->end_bit >> 3);
            if (r16->stop == STOP_EOF) s.eof = true;
            s.pass2.push_back(pc);
            {
                std::lock_guard<std::mutex> g(s.m);
                ++s.outstanding;
            }
            s.submit([keep, pc, win, lowest_ok] {
                Run<uint16_t>& r = *pc->r16;
                std::shared_ptr<Text> t = std::make_shared<Text>();
                size_t cap = 0;
                uint8_t* bytes = (uint8_t*)buf_get(r.n + 64, &cap);
                bool ok = bytes != nullptr;
                if (ok) {
                    t->owner = std::shared_ptr<void>(bytes, [cap](void* q) { buf_put(q, cap); });
                    t->data = bytes;
                    t->len = r.n;
                    // markers before the run's first member end reach into the predecessor's history (as far as
                    // that member goes back); behind a member end there can be none at all
                    const size_t first_part = r.ends.empty() ? r.n : r.ends.front().out_pos;
                    ok = translate(r.buf + WIN, first_part, win->data(), bytes, lowest_ok);
                    if (first_part < r.n) ok = translate(r.buf + WIN + first_part, r.n - first_part, win->data(), bytes + first_part, WIN) && ok;
                    crc_segments(bytes, r.n, r.ends, pc->segs);
                }
                pc->text = t;
                pc->r16.reset();
                job_done(keep, [&] {
                    pc->bad_reach = !ok;
                    pc->done2 = true;
                });
            });
            continue;
        }
        // exact inflate on this thread with the known window: the first chunk, a candidate that was not a block
        // start, a search that found nothing
        ++s.st.serial;
        pc->r16.reset();
        pc->r8.reset(new Run<uint8_t>());
        Run<uint8_t>& r = *pc->r8;
        if (!r.reserve(WIN + C * 4 + (1u << 20))) {
            fail(s, "out of memory inflating");
            return -1;
        }
        memcpy(r.buf, s.window, WIN);
        inflate_run<uint8_t>(s.data, s.size, s.expect_bit, s.expect_header, last ? INT64_MAX : chunk_end_bit, s.hist, r);
        if (r.stop == STOP_ERROR) {
            fail(s, r.err);
            return -1;
        }
        advance_window(s, r.buf + WIN, r.n, r.ends);
        s.expect_bit = r.end_bit;
        s.expect_header = false;
        s.consumed_bytes = (size_t)(r.end_bit >> 3);
        if (r.stop == STOP_EOF) s.eof = true;
        crc_segments(r.buf + WIN, r.n, r.ends, pc->segs);
        {  // the run's buffer is the text (its first WIN bytes are the history)
            std::shared_ptr<Run<uint8_t>> run(pc->r8.release());
            pc->text = std::make_shared<Text>();
            pc->text->data = run->buf + WIN;
            pc->text->len = run->n;
            pc->text->owner = run;
        }
        pc->done2 = true;
        s.pass2.push_back(pc);
    }
}

}  // namespace qdpgz
