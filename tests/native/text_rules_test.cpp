// Host-side checks of what the device text stages share with the host (quade_amd/csrc/text_rules.h, crc_lds.h): the CRC-32
// arithmetic against zlib, the name cut against the rule it restates.  Built and run by tests/test_host_text.py (no GPU).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../quade_amd/csrc/crc_lds.h"
#include "../../quade_amd/csrc/text_rules.h"

static int fails = 0;
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                          \
        }                                                     \
    } while (0)

static uint32_t z(const std::vector<uint8_t>& v, size_t a, size_t b) { return (uint32_t)crc32(0L, v.data() + a, (uInt)(b - a)); }

int main() {
    srand(7);
    std::vector<uint8_t> v(300000);
    for (auto& b : v) b = (uint8_t)(rand() >> 7);
    uint32_t pw[32];
    qd_crc_pow_table(pw);
    // crc(A || B) = crc(A) * x^(8 |B|) + crc(B): ranges of many lengths, also empty ones
    const size_t cuts[] = {0, 1, 3, 4, 255, 256, 257, 65535, 65536, 65537, 200001, 300000};
    for (size_t a : cuts)
        for (size_t b : cuts) {
            if (a > b) continue;
            const uint32_t whole = z(v, 0, b), left = z(v, 0, a), right = z(v, a, b);
            CHECK((qd_crc_mulmod(left, qd_crc_xpow8(pw, (uint32_t)(b - a))) ^ right) == whole);
        }
    // the compile-time tables of the LDS form: slice-by-4 over them equals zlib, and the slice shifts equal x^(8 n)
    constexpr qdcrc::Tables T = qdcrc::make_tables();
    for (size_t n : {0u, 4u, 8u, 1024u, 65536u}) {
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = 0; i < n; i += 4) {
            uint32_t w;
            memcpy(&w, v.data() + i, 4);
            c ^= w;
            c = T.t[3][c & 0xFFu] ^ T.t[2][(c >> 8) & 0xFFu] ^ T.t[1][(c >> 16) & 0xFFu] ^ T.t[0][c >> 24];
        }
        CHECK(~c == z(v, 0, n));
    }
    constexpr qdcrc::Shifts<33> S33 = qdcrc::make_shifts<33>();
    constexpr qdcrc::Shifts<17> S17 = qdcrc::make_shifts<17>();
    for (uint32_t j : {0u, 1u, 31u, 32u, 33u, 500u, 1023u}) {
        CHECK(qd_crc_mulmod(S33.lo[j & 31], S33.hi[j >> 5]) == qd_crc_xpow8(pw, 33u * 4u * j));
        CHECK(qd_crc_mulmod(S17.lo[j & 31], S17.hi[j >> 5]) == qd_crc_xpow8(pw, 17u * 4u * j));
    }
    for (uint32_t k = 0; k < 4; ++k) CHECK(S33.tail[k] == qd_crc_xpow8(pw, k));
    // the name of a record: header without its first byte, first blank-delimited token (Python's bytes.split)
    struct {
        const char* head;
        const char* name;
    } cases[] = {{"@SIM:1:FC 1:N:0", "SIM:1:FC"}, {"@", ""}, {"", ""}, {"@ \t lead\ttail", "lead"}, {"@a\r", "a"}, {"@ ", ""}, {"xname rest", "name"},
                 {"@n\x0b" "x", "n"}, {"@n\x0c" "x", "n"}, {"@\ttab", "tab"}};
    for (auto& c : cases) {
        const std::string h = c.head;
        uint32_t off = 0, len = 0;
        qd_name_of((const uint8_t*)h.data(), 0, (uint32_t)h.size(), &off, &len);
        CHECK(h.substr(off, len) == c.name);
    }
    CHECK(qd_slice_len(0, 8, 8) == 8 && qd_slice_len(0, 8, 5) == 5 && qd_slice_len(6, 8, 5) == 0 && qd_slice_len(3, 3, 9) == 0 && qd_slice_len(2, 9, 0) == 0);
    printf(fails ? "FAILED %d\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
