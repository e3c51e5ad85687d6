// The third inflater's lane decoder (quade_amd/csrc/inflate3_lane.h) on the host, against zlib: raw deflate streams of every block
// type, several levels and strategies, texts that stress the tables (many symbols with long codes, one symbol, runs), damaged
// input, and a stream cut into units at its block boundaries (what the gzip path does).  Build: g++ -O1 -std=c++17 ... -lz
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../quade_amd/csrc/inflate3_lane.h"

static int g_fail = 0, g_redone = 0;
#define CHECK(x)                                                     \
    do {                                                             \
        if (!(x)) {                                                  \
            printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #x);      \
            ++g_fail;                                                \
        }                                                            \
    } while (0)

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& text, int level, int strategy, int mem_level = 8) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, -15, mem_level, strategy);
    std::vector<uint8_t> out(deflateBound(&z, text.size()) + 64);
    z.next_in = const_cast<uint8_t*>(text.data());
    z.avail_in = (uInt)text.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    if (rc != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// tokens -> text, one after the other
static bool expand(const std::vector<uint16_t>& tok, uint32_t n_slots, std::vector<uint8_t>& text) {
    for (uint32_t i = 0; i < n_slots; ++i) {
        const uint32_t s = tok[i];
        if (!(s & qd3::TOK_MATCH)) {
            if (s > 255) return false;
            text.push_back((uint8_t)s);
            continue;
        }
        if (i + 1 >= n_slots) return false;
        const uint32_t len = (s & 0xFF) + 3, dist = (uint32_t)tok[++i] + 1;
        if ((s & 0x7F00) || dist > 32768 || dist > text.size()) return false;
        for (uint32_t k = 0; k < len; ++k) text.push_back(text[text.size() - dist]);
    }
    return true;
}

template <class C>
struct Run {
    qd3::Result res{};
    std::vector<uint16_t> tok;
    std::vector<uint64_t> headers;  // bit positions of the block headers the lane passed
};

template <class C>
static Run<C> run_unit(const std::vector<uint8_t>& comp, uint64_t bit_start, uint64_t bit_stop, uint32_t tok_cap, const std::vector<uint64_t>* cands = nullptr) {
    std::vector<uint32_t> words((comp.size() + 3) / 4 + 80, 0);  // (a lane's ring holds the 256 bytes behind its position)
    memcpy(words.data(), comp.data(), comp.size());
    std::vector<uint16_t> lds(2 * C::LANE_DW);
    std::vector<uint32_t> ring(C::RING_DW, 0xABABABAB), lens(qd3::LENS_DW, 0xDEADBEEF);
    Run<C> r;
    r.tok.assign(tok_cap + 8, 0xEEEE);
    qd3::Unit u{words.data(), bit_start, bit_stop, (uint64_t)comp.size() * 8, 0, tok_cap, (uint32_t)((comp.size() + 3) / 4 + 80), cands ? cands->data() : nullptr,
                cands ? (uint32_t)cands->size() : 0u, 0};
    qd3::Lane<C> L;
    const uint32_t lane = 0;
    qd3::lane_init(L, u, lens.data());
    qd3::topup(L, ring.data(), lane, true);
    qd3::ring_wait();
    qd3::landed_all(L);
    qd3::prime(L, ring.data(), lane);
    uint64_t guard = 0;
    // the kernel's loop for one lane: headers with the ring full, rounds of turns between two top-ups -- a dword is only ever taken
    // from a chunk that has landed (checked here; the ring's sizing argument in inflate3_lane.h)
    while (L.state != qd3::ST_DONE && ++guard < (1ull << 28)) {
        if (L.state == qd3::ST_HEADER) {
            qd3::topup(L, ring.data(), lane, true);
            qd3::landed_all(L);
            r.headers.push_back(qd3::bitpos(L));
            qd3::header<C>(L, lds.data(), ring.data(), lane);
            qd3::topup(L, ring.data(), lane, L.state <= qd3::ST_STORED);
            qd3::landed_all(L);
            continue;
        }
        qd3::landed_all(L);
        const uint32_t landed = L.landed;
        qd3::topup(L, ring.data(), lane, true);
        // (what this top-up requests counts as not there yet: poison it, so that a turn that took it would decode garbage)
        uint32_t saved[64];
        int n_saved = 0;
        for (uint32_t c = landed; c < L.fetched; ++c)
            for (int k = 0; k < 4; ++k) {
                uint32_t& w = ring[qd3::ring_at<C>(4 * c + k, lane)];
                saved[n_saved++] = w;
                w = 0x5A5A5A5A;
            }
        for (int t = 0; t < C::ROUND_TURNS && L.state <= qd3::ST_STORED; ++t) {
            if (L.state == qd3::ST_STORED) qd3::turn_stored<C>(L, ring.data(), lane, r.tok.data());
            else qd3::turn<C>(L, lds.data(), ring.data(), lane, r.tok.data());
            if (L.rd > 4 * landed) {
                printf("FAIL: a turn took dword %u, landed %u chunks\n", L.rd, landed);
                ++g_fail;
            }
        }
        n_saved = 0;
        for (uint32_t c = landed; c < L.fetched; ++c)
            for (int k = 0; k < 4; ++k) ring[qd3::ring_at<C>(4 * c + k, lane)] = saved[n_saved++];
    }
    qd3::lane_finish(L, &r.res);
    return r;
}

template <class C>
static void check_stream(const char* what, const std::vector<uint8_t>& text, int level, int strategy, bool allow_space = false) {
    const std::vector<uint8_t> comp = deflate_raw(text, level, strategy);
    const uint32_t cap = (uint32_t)((text.size() + 16 + 3) & ~3u) + 8;
    Run<C> r = run_unit<C>(comp, 0, ~0ull, cap);
    if (allow_space && r.res.status == QD_INFLATE_TABLE_SPACE) return;
    // a configuration too small for the fixed code (264 symbols behind 7 bits, 112 behind 8) says "table space" at a fixed block: the
    // launch decodes such a unit again with the large configuration (quade_inflate3.hip: launch_tokens) -- so does the test
    constexpr bool fixed_fits = C::LB >= 9 || (C::LB == 8 && C::NLONG >= 112);
    if (!fixed_fits && r.res.status == QD_INFLATE_TABLE_SPACE) {
        ++g_redone;
        Run<qd3::Cfg<8, 7, 112>> again = run_unit<qd3::Cfg<8, 7, 112>>(comp, 0, ~0ull, cap);
        CHECK(again.res.status == 0 && again.res.final_seen && again.res.text_len == text.size());
        std::vector<uint8_t> got;
        CHECK(expand(again.tok, again.res.n_slots, got));
        CHECK(got == text);
        return;
    }
    if (r.res.status != 0 || !r.res.final_seen || r.res.text_len != text.size()) {
        printf("FAIL %s level %d strategy %d: status %u final %u text %u of %zu\n", what, level, strategy, r.res.status, r.res.final_seen, r.res.text_len, text.size());
        ++g_fail;
        return;
    }
    std::vector<uint8_t> got;
    CHECK(expand(r.tok, r.res.n_slots, got));
    if (got != text) {
        printf("FAIL %s level %d strategy %d: text differs\n", what, level, strategy);
        ++g_fail;
    }
    // the decode ends exactly behind the last block: at most 7 bits of padding to the stream's end
    CHECK(r.res.bit_next <= comp.size() * 8 && comp.size() * 8 - r.res.bit_next < 8);
    // the same stream cut into units at its block boundaries: every unit stops where the next one starts, the texts add up
    if (r.headers.size() >= 3) {
        std::vector<uint8_t> whole;
        std::vector<std::vector<uint16_t>> parts;
        uint64_t total = 0;
        bool ok = true;
        std::vector<uint16_t> all;
        for (size_t k = 0; k < r.headers.size(); k += 2) {
            const uint64_t stop = k + 2 < r.headers.size() ? r.headers[k + 2] : ~0ull;
            Run<C> u = run_unit<C>(comp, r.headers[k], stop, cap);
            ok = ok && u.res.status == 0 && (stop == ~0ull ? u.res.final_seen == 1 : (u.res.bit_next == stop && !u.res.final_seen));
            total += u.res.text_len;
            all.insert(all.end(), u.tok.begin(), u.tok.begin() + u.res.n_slots);
        }
        CHECK(ok);
        CHECK(total == text.size());
        std::vector<uint8_t> got2;
        CHECK(expand(all, (uint32_t)all.size(), got2));
        CHECK(got2 == text);
        // a stop position that is no block boundary: the unit goes on to the next candidate that is one (here: the third block's
        // start, among candidates that are none), or to the stream's end without any
        const std::vector<uint64_t> cands = {r.headers[1] + 1, r.headers[1] + 5, r.headers[2], r.headers[2] + 3};
        Run<C> on = run_unit<C>(comp, r.headers[0], r.headers[1] + 1, cap, &cands);
        CHECK(on.res.status == 0 && on.res.bit_next == r.headers[2] && !on.res.final_seen);
        Run<C> through = run_unit<C>(comp, r.headers[0], r.headers[1] + 1, cap);
        CHECK(through.res.status == 0 && through.res.final_seen && through.res.text_len == text.size());
    }
}

static std::vector<uint8_t> fastq(std::mt19937& g, size_t n_bytes, int n_qual) {
    std::string out;
    size_t i = 0;
    while (out.size() < n_bytes) {
        const int L = 30 + (int)(g() % 121);
        char head[96];
        snprintf(head, sizeof head, "@SIM:1:FC:%zu:%zu 1:N:0:\n", i, i * 7);
        out += head;
        for (int k = 0; k < L; ++k) out += "ACGTN"[g() % 100 == 0 ? 4 : g() % 4];
        out += "\n+\n";
        for (int k = 0; k < L; ++k) out += (char)(33 + 2 + g() % n_qual);
        out += "\n";
        ++i;
    }
    out.resize(n_bytes);
    return std::vector<uint8_t>(out.begin(), out.end());
}

template <class C>
static void suite(const char* name) {
    std::mt19937 g(12345);
    const std::vector<uint8_t> fq = fastq(g, 600000, 11), fq40 = fastq(g, 400000, 41);
    std::vector<uint8_t> rnd(200000), runs(300000, 'A'), period, two(70000), skew(300000), empty, one(1, 'x');
    for (auto& b : rnd) b = (uint8_t)g();
    for (size_t i = 0; i < 250001; ++i) period.push_back("ACGTTGCA"[i % 8]);
    for (auto& b : two) b = "AB"[g() & 1];
    for (auto& b : skew) {  // many byte values, geometrically rarer: a literal code with a long tail of long codes
        uint32_t v = 0;
        while (v < 200 && (g() & 3) != 0) ++v;
        b = (uint8_t)(32 + v);
    }
    const int levels[] = {0, 1, 6, 9};
    for (int lv : levels) {
        check_stream<C>("fastq", fq, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("fastq 41 qualities", fq40, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("random", rnd, lv, Z_DEFAULT_STRATEGY, true);
        check_stream<C>("runs", runs, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("period", period, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("two symbols", two, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("skewed bytes", skew, lv, Z_DEFAULT_STRATEGY, true);
        check_stream<C>("empty", empty, lv, Z_DEFAULT_STRATEGY);
        check_stream<C>("one byte", one, lv, Z_DEFAULT_STRATEGY);
    }
    const int strategies[] = {Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
    for (int st : strategies) {
        check_stream<C>("fastq", fq, 6, st);
        check_stream<C>("random", rnd, 6, st, true);
        check_stream<C>("skewed bytes", skew, 6, st, true);
        check_stream<C>("runs", runs, 6, st);
    }
    // how often the long-code table is too small on byte soup (reported, not a failure: another inflater takes such blocks)
    {
        const std::vector<uint8_t> comp = deflate_raw(rnd, 6, Z_HUFFMAN_ONLY);
        Run<C> r = run_unit<C>(comp, 0, ~0ull, 400000);
        printf("%s: random bytes, Huffman only: status %u (10 = table space)\n", name, r.res.status);
    }
    // damage: every outcome is a status or a text, never a crash or an endless loop; a cut stream is "truncated"
    {
        std::vector<uint8_t> comp = deflate_raw(fq, 6, Z_DEFAULT_STRATEGY);
        for (int t = 0; t < 300; ++t) {
            std::vector<uint8_t> bad = comp;
            bad[g() % bad.size()] ^= (uint8_t)(1u << (g() % 8));
            Run<C> r = run_unit<C>(bad, 0, ~0ull, 1u << 20);
            (void)r;
        }
        for (size_t cut : {(size_t)1, (size_t)5, comp.size() / 2, comp.size() - 1}) {
            std::vector<uint8_t> bad(comp.begin(), comp.begin() + cut);
            Run<C> r = run_unit<C>(bad, 0, ~0ull, 1u << 20);
            CHECK(r.res.status != 0);
        }
        // too little room for the tokens
        Run<C> r = run_unit<C>(comp, 0, ~0ull, 1024);
        CHECK(r.res.status == QD_INFLATE_TOKEN_SPACE);
    }
    printf("%s: done (%d streams decoded again by the large configuration)\n", name, g_redone);
    g_redone = 0;
}

int main() {
    suite<qd3::Cfg<8, 7, 112>>("LB 8 / DB 7 / 112 long");
    suite<qd3::Cfg<9, 6, 56>>("LB 9 / DB 6 / 56 long");
    suite<qd3::Cfg<10, 7, 96>>("LB 10 / DB 7 / 96 long");
    suite<qd3::Cfg<7, 6, 88, true, 8, 12>>("LB 7 / DB 6 in bytes / 88 long / ring of 8 chunks, 12 turns a round");
    suite<qd3::Cfg<8, 7, 64, true, 4, 6>>("LB 8 / DB 7 in bytes / 64 long / ring of 4 chunks, 6 turns a round");
    if (g_fail) {
        printf("%d checks failed\n", g_fail);
        return 1;
    }
    printf("all checks passed\n");
    return 0;
}
