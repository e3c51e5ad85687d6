// Measurement tool (NOT product code, not linked into libquade_hip.so, not a fallback):
// a tuned multi-threaded CPU version of the dual 8+8 hot path on packed rows, to put the GPU rate
// next to what the host's own cores can do.  Same SWAR fold / gate / hash as quade_common.h.
//   g++ -O3 -march=native -std=c++17 -pthread tools/cpu_strong.cpp -o tools/cpu_strong
//   tools/cpu_strong <pairs> <threads> <samples>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../quade_amd/csrc/quade_common.h"

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 50000000;
    const int T = argc > 2 ? atoi(argv[2]) : (int)std::thread::hardware_concurrency();
    const int S = argc > 3 ? atoi(argv[3]) : 96;
    std::mt19937_64 rng(7);
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    std::vector<uint64_t> bc(2 * S);
    for (auto& w : bc) {
        w = 0;
        for (int i = 0; i < 8; ++i) w |= (uint64_t)acgt[rng() & 3] << (8 * i);
    }
    uint32_t mask = 16;
    while (mask < 4u * S) mask <<= 1;
    mask -= 1;
    std::vector<uint32_t> slots(mask + 1, QD_EMPTY_SLOT);
    for (int i = 0; i < S; ++i) {
        uint64_t w[4] = {bc[2 * i], bc[2 * i + 1], 0, 0};
        uint32_t h = qd_hash_key(w, 16, 0), s = h & mask;
        while (slots[s] != QD_EMPTY_SLOT) s = (s + 1) & mask;
        slots[s] = qd_slot_entry(h, i);
    }
    std::vector<uint64_t> s1(n), s2(n), q1(n), q2(n);
    std::vector<uint16_t> codes(n);
    for (int64_t i = 0; i < n; ++i) {
        const int b = rng() % S;
        const bool hit = (rng() % 10) != 0;
        s1[i] = hit ? bc[2 * b] : rng();
        s2[i] = bc[2 * b + 1];
        q1[i] = 0x4949494949494949ull;
        q2[i] = (rng() % 7) ? 0x4949494949494949ull : 0x4949492549494949ull;
    }
    std::vector<std::vector<uint64_t>> hist(T, std::vector<uint64_t>(2 * S + 1, 0));
    auto work = [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        auto& hh = hist[t];
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t klo = qd_fold8(s1[i]), khi = qd_fold8(s2[i]);
            uint32_t h = qd_hash_init(16, 0);
            h = qd_hash_fini(qd_hash_step(qd_hash_step(h, klo), khi));
            uint32_t s = h & mask, code = 0xFFFF;
            for (;;) {
                const uint32_t e = slots[s];
                if (e == QD_EMPTY_SLOT) break;
                if ((e >> 16) == (h >> 16)) {
                    const uint32_t id = e & 0xFFFF;
                    if (bc[2 * id] == klo && bc[2 * id + 1] == khi) {
                        code = id * 2 + ((qd_all_ge8(q1[i], 58) & qd_all_ge8(q2[i], 58)) ^ 1);
                        break;
                    }
                }
                s = (s + 1) & mask;
            }
            codes[i] = (uint16_t)code;
            hh[code == 0xFFFF ? 2 * S : code] += 1;
        }
    };
    for (int rep = 0; rep < 3; ++rep) {
        for (auto& h : hist) std::fill(h.begin(), h.end(), 0);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t und = 0, tot = 0;
        for (auto& h : hist) {
            und += h[2 * S];
            for (auto v : h) tot += v;
        }
        printf("{\"tool\": \"cpu_strong\", \"pairs\": %lld, \"threads\": %d, \"samples\": %d, \"pairs_per_s\": %.3e, "
               "\"GBps_algorithmic\": %.1f, \"undetermined\": %llu, \"total\": %llu}\n",
               (long long)n, T, S, n / dt, n * 34.0 / dt / 1e9, (unsigned long long)und, (unsigned long long)tot);
    }
    return 0;
}
