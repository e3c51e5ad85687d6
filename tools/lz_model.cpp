// CPU model of the parse of quade_deflate.hip's lz_subblocks (the device's `gzip_level : 1`): same sub-blocks, rounds of
// look-ups (a position sees what was entered up to the round before its own), buckets of last positions, run candidate, 32-byte
// views, the rule for matches inside the sequence lines, lazy walk cut into quarters -- and the size the two Huffman codes
// would give it.  For tuning the parse's choices without a GPU; the defaults are the kernel's.  Its history on 2 MB of
// binned-quality records (zlib level 1: 22.5 % of the text, level 6: 19.0 %):
//   waves=4 look=64 cut=0 ways=1 hash=14 nice=0 lazy=0 dna=0 run=0   31.8 %  four waves on quarters of a sub-block, side by side
//   waves=1 look=64 cut=0 ways=4 hash=12 run=0                         19.5 %  one wave per sub-block (complete history): 10 ms per 64 MB
//   cost=0                                                               18.8 %  the run candidate always compared, rounds of 256
//   (defaults)                                                           18.7 %  + matches priced against their literals (the benchmarks' records with random qualities: 43.0 -> 41.5 %)
// The device's members are checked with zlib by tests/test_gpu_deflate.py.
//   g++ -O2 -o /tmp/lz_model tools/lz_model.cpp && /tmp/lz_model text [key=value ...]
//   keys: sub (65536) look (256: positions looked up before any of them is entered) cut (16384: the walk's regions)
//         hash (11) ways (2) run (1: distance 1 always compared) minlen (4) nice (32) lazy (1)
//         dna (12: least length of a match that covers only ACGTN) waves (1: > 1 = regions looked up side by side, the first form)
//         cost (2: a match must cost less than the literals it replaces, priced by the 8 bytes at its start; 1: without the dna rule; 0: off) cbase (13)
//         hist (0: bytes of history before the sub-block) far (0: a match of < far_len bytes farther than `far` is dropped) far_len (0)
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <vector>

static std::map<std::string, long> opt = {{"sub", 65536}, {"waves", 1},  {"hash", 11},   {"minlen", 4}, {"ways", 2},
                                          {"hist", 0},    {"lazy", 1},   {"far", 0},     {"far_len", 0}, {"maxlen", 256},
                                          {"dna", 12},    {"nice", 32},  {"look", 256}, {"cut", 16384}, {"run", 1},    {"cost", 2},   {"cbase", 13}};

static double huff_bits(const std::vector<uint64_t>& f) {  // total bits of an (unlimited) Huffman code for these counts
    std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> q;
    for (uint64_t v : f)
        if (v) q.push(v);
    if (q.size() == 1) return (double)q.top();
    double bits = 0;
    while (q.size() > 1) {
        uint64_t a = q.top();
        q.pop();
        uint64_t b = q.top();
        q.pop();
        bits += (double)(a + b);
        q.push(a + b);
    }
    return bits;
}
static void len_symbol(uint32_t l, uint32_t& sym, uint32_t& eb) {
    if (l < 8) sym = 257 + l, eb = 0;
    else if (l == 255) sym = 285, eb = 0;
    else {
        uint32_t n = 31 - __builtin_clz(l);
        eb = n - 2;
        sym = 257 + 4 * (n - 1) + ((l >> eb) & 3);
    }
}
static void dist_symbol(uint32_t d, uint32_t& sym, uint32_t& eb) {
    if (d < 4) sym = d, eb = 0;
    else {
        uint32_t n = 31 - __builtin_clz(d);
        eb = n - 1;
        sym = 2 * n + ((d >> eb) & 1);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    for (int i = 2; i < argc; ++i) {
        char* e = strchr(argv[i], '=');
        if (e) opt[std::string(argv[i], e - argv[i])] = atol(e + 1);
    }
    FILE* f = fopen(argv[1], "rb");
    std::vector<uint8_t> text;
    uint8_t buf[1 << 16];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) text.insert(text.end(), buf, buf + r);
    fclose(f);
    const long SUB = opt["sub"], W = opt["waves"], HB = opt["hash"], MINLEN = opt["minlen"], WAYS = opt["ways"], HIST = opt["hist"],
               MAXLEN = opt["maxlen"], LAZY = opt["lazy"], FAR = opt["far"], FAR_LEN = opt["far_len"], DNA = opt["dna"], NICE = opt["nice"], LOOK = opt["look"], CUT = opt["cut"], RUN = opt["run"], COST = opt["cost"], CBASE = opt["cbase"];
    const size_t N = text.size();
    text.resize(N + 512, 0);
    double total_bits = 0;
    uint64_t nlit = 0, nmatch = 0, mbytes = 0;
    uint64_t len_hist[8] = {0};
    for (size_t s0 = 0; s0 < N; s0 += SUB) {
        const long L = (long)std::min<size_t>(SUB, N - s0);
        const long H = (long)std::min<size_t>(HIST, s0);  // history bytes before the sub-block (same piece)
        const uint8_t* t = text.data() + s0;              // t[-H .. L)
        std::vector<int32_t> table((size_t)WAYS << HB, -0x40000000);
        auto hash = [&](long p) {
            uint32_t u;
            memcpy(&u, t + p, 4);
            return (u * 2654435761u) >> (32 - HB);
        };
        auto insert = [&](long p) {
            int32_t* b = &table[(size_t)hash(p) * WAYS];
            for (long k = WAYS - 1; k > 0; --k) b[k] = b[k - 1];
            b[0] = (int32_t)p;
        };
        for (long p = -H; p + 4 <= 0 && p < 0; ++p) insert(p);  // (the device would enter the history in parallel)
        // cost = 1: literal costs from the sub-block's byte histogram (bits, rounded), a match is taken when the literals it
        // replaces (priced by the 8 bytes at its start) cost more than cbase + log2(distance) bits
        int bcost[256];
        {
            uint64_t h[256] = {0};
            for (long i = 0; i < L; ++i) ++h[t[i]];
            for (int b = 0; b < 256; ++b) bcost[b] = h[b] ? std::max(1, std::min(12, (int)std::lround(std::log2((double)L / (double)h[b])))) : 12;
        }
        std::vector<uint64_t> lf(286, 0), df(30, 0);
        uint64_t extra = 0;
        lf[256] = 1;
        const long REG = SUB / W;
        std::vector<long> base(W), carry(W, 0);
        for (long w = 0; w < W; ++w) base[w] = w * REG;
        auto match_len = [&](long p, long c, long maxlen) {
            long n = 0;
            while (n < maxlen && t[p + n] == t[c + n]) ++n;
            return n;
        };
        // look > 64: the candidates of `look` positions are looked up before any of them is entered (the kernel's rounds of
        // several waves); cut > 0: the walk is cut every `cut` bytes (regions walked side by side after the look-ups)
        std::vector<std::array<long, 8>> pre;
        if (LOOK > 64) {
            pre.assign((size_t)L + 64, std::array<long, 8>{-1, -1, -1, -1, -1, -1, -1, -1});
            for (long r0 = 0; r0 < L; r0 += LOOK) {
                for (long p = r0; p < std::min(r0 + LOOK, L); ++p) {
                    if (p + 4 > L) continue;
                    const int32_t* bk = &table[(size_t)hash(p) * WAYS];
                    long nc = 0;
                    for (long k = 0; k < WAYS; ++k) {
                        const long c = bk[k];
                        if (c < p && p - c <= 32768 && c >= -H && memcmp(t + c, t + p, 4) == 0) pre[p][nc++] = c;
                    }
                    if ((nc == 0 || (RUN && nc < 8)) && p - 1 >= -H && memcmp(t + p - 1, t + p, 4) == 0) {
                        if (RUN) {  // the nearest candidate: first
                            for (long k = nc; k > 0; --k) pre[p][k] = pre[p][k - 1];
                            pre[p][0] = p - 1;
                            ++nc;
                        } else
                            pre[p][nc++] = p - 1;
                    }
                }
                for (long p = r0; p < std::min(r0 + LOOK, L); ++p)
                    if (p + 4 <= L) insert(p);
            }
        }
        bool any = true;
        while (any) {
            any = false;
            for (long w = 0; w < W; ++w) {  // the waves advance in lockstep, one stretch each
                long rend = std::min((w + 1) * REG, L);
                if (CUT) rend = std::min(rend, (base[w] / CUT + 1) * CUT);  // (with waves=1: regions of `cut` bytes one after the other)
                if (base[w] >= rend) continue;
                any = true;
                const long b0 = base[w], limit = std::min<long>(64, rend - b0);
                long cand[64][8];
                for (long i = 0; i < 64; ++i) {  // look-ups of the whole stretch first
                    const long p = b0 + i;
                    for (long k = 0; k < 8; ++k) cand[i][k] = -1;
                    if (LOOK > 64) {
                        if (p < L) for (long k = 0; k < 8; ++k) cand[i][k] = pre[p][k];
                        continue;
                    }
                    if (p + 4 > L) continue;
                    const int32_t* bk = &table[(size_t)hash(p) * WAYS];
                    long nc = 0;
                    for (long k = 0; k < WAYS; ++k) {
                        const long c = bk[k];
                        if (c < p && p - c <= 32768 && c >= -H && memcmp(t + c, t + p, 4) == 0) cand[i][nc++] = c;
                    }
                    if ((nc == 0 || (RUN && nc < 8)) && p - 1 >= -H && memcmp(t + p - 1, t + p, 4) == 0) {
                        if (RUN) {
                            for (long k = nc; k > 0; --k) cand[i][k] = cand[i][k - 1];
                            cand[i][0] = p - 1;
                            ++nc;
                        } else
                            cand[i][nc++] = p - 1;
                    }
                }
                for (long i = 0; i < 64 && LOOK <= 64; ++i)
                    if (b0 + i + 4 <= L) insert(b0 + i);
                // capped: the lane's own view (candidates compared up to `nice` bytes); the walk extends the chosen one fully
                auto best_at = [&](long s, long& blen, long& bdist, bool capped) {
                    blen = 0;
                    bdist = 0;
                    if (s >= 64) return;
                    const long ps = b0 + s, maxlen = std::min<long>(MAXLEN, rend - ps);
                    long bcap = 0;
                    for (long k = 0; k < 8 && cand[s][k] != -1; ++k) {
                        const long n = match_len(ps, cand[s][k], NICE ? std::min(NICE, maxlen) : maxlen);
                        if (n > bcap) bcap = n, bdist = ps - cand[s][k];
                    }
                    blen = bcap;
                    if (bcap && !capped) blen = match_len(ps, ps - bdist, maxlen);
                    long need = MINLEN;
                    if (DNA) {  // literals that are cheap (ACGTN) make a short match a loss
                        long cheap = 0;
                        for (long k = 0; k < 8; ++k) cheap += t[ps + k] && strchr("ACGTN", t[ps + k]) != nullptr;
                        if (cheap == 8) need = DNA;
                    }
                    if (COST) {
                        long c8 = 0;
                        for (long k = 0; k < 8; ++k) c8 += bcost[t[ps + k]];
                        const long mbits = CBASE + (bdist > 1 ? 63 - __builtin_clzl((unsigned long)bdist) : 0);
                        if (COST == 1) need = MINLEN;  // (cost = 2: the rule for bases stays as well)
                        if (blen * c8 < mbits * 8) blen = 0;  // (length x average literal cost of the first 8 bytes) against the match's own cost
                    }
                    if (blen < need || (FAR && bdist > FAR && blen < FAR_LEN)) blen = 0;
                };
                long s = carry[w];
                while (s < limit) {
                    long blen, bdist;
                    best_at(s, blen, bdist, false);
                    if (blen && LAZY && s + 1 < limit) {
                        long l1, d1, l2, d2;
                        best_at(s, l1, d1, true);
                        best_at(s + 1, l2, d2, true);
                        if (l2 > l1 + (LAZY - 1) && (!NICE || l1 < NICE)) blen = 0;  // a literal now, the longer match next
                    }
                    if (blen) {
                        uint32_t sym, eb;
                        len_symbol((uint32_t)blen - 3, sym, eb);
                        ++lf[sym];
                        extra += eb;
                        dist_symbol((uint32_t)bdist - 1, sym, eb);
                        ++df[sym];
                        extra += eb;
                        ++nmatch;
                        mbytes += blen;
                        ++len_hist[std::min<long>(7, blen / 4)];
                        s += blen;
                    } else {
                        ++lf[t[b0 + s]];
                        ++nlit;
                        ++s;
                    }
                }
                carry[w] = s - 64;
                base[w] += 64;
            }
        }
        int nl = 286, nd = 30;
        while (nl > 257 && !lf[nl - 1]) --nl;
        while (nd > 1 && !df[nd - 1]) --nd;
        // header: the code lengths run-length coded with the kernel's fixed code-length code -- 5 bits per used symbol (equal
        // neighbours would repeat for 7: not modelled), 2 per lone zero, 6 / 10 per run of 3-10 / 11-138 zeros
        double hdr = 3 + 14 + 57;
        {
            int run = 0;
            auto flush = [&] {
                while (run >= 11) hdr += 10, run -= std::min(run, 138);
                if (run >= 3) hdr += 6, run = 0;
                hdr += 2 * run;
                run = 0;
            };
            for (int i = 0; i < nl + nd; ++i) {
                const bool used = i < nl ? lf[i] != 0 : df[i - nl] != 0;
                if (used) flush(), hdr += 5;
                else ++run;
            }
            flush();
        }
        total_bits += huff_bits(lf) + huff_bits(df) + extra + hdr + 3 + 7 + 32;
    }
    printf("%s:", argv[1]);
    for (int i = 2; i < argc; ++i) printf(" %s", argv[i]);
    printf("\n  %.0f bytes = %.2f %% of %zu; %llu literals, %llu matches covering %.1f %% (avg %.1f); match lengths 4-7 %llu, 8-11 %llu, 12-15 %llu, 16-19 %llu, 20-27 %llu / %llu, 28+ %llu\n",
           total_bits / 8 + 20, 100.0 * (total_bits / 8 + 20) / N, N, (unsigned long long)nlit, (unsigned long long)nmatch, 100.0 * mbytes / N,
           nmatch ? (double)mbytes / nmatch : 0.0, (unsigned long long)len_hist[1], (unsigned long long)len_hist[2], (unsigned long long)len_hist[3],
           (unsigned long long)len_hist[4], (unsigned long long)len_hist[5], (unsigned long long)len_hist[6], (unsigned long long)len_hist[7]);
    return 0;
}
