// Sanitizer driver for the parallel gunzip (quade_amd/csrc/quade_pgz.cpp): built by tests/test_host_gunzip.py with
// -fsanitize=address,undefined or -fsanitize=thread.  Reads a gzip file, then for a number of rounds damages a copy
// (bit flips, truncation, spliced garbage, zeroed stretches), inflates it with the parallel inflater on real threads
// and small chunks, and compares with zlib: whenever zlib accepts the stream the inflater must deliver the same bytes;
// whenever the inflater delivers a whole stream without an error, zlib must agree with every byte of it.
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iterator>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "quade_amd/csrc/quade_pgz.h"

uint32_t qd_io_crc32(const uint8_t* p, size_t n) {
    uLong c = crc32(0L, Z_NULL, 0);
    while (n) {
        const uInt k = (uInt)(n > (1u << 30) ? (1u << 30) : n);
        c = crc32(c, p, k);
        p += k;
        n -= k;
    }
    return (uint32_t)c;
}

namespace {
struct Workers {
    std::vector<std::thread> th;
    std::deque<std::function<void()>> q;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    explicit Workers(int n) {
        for (int i = 0; i < n; ++i)
            th.emplace_back([this] {
                for (;;) {
                    std::function<void()> fn;
                    {
                        std::unique_lock<std::mutex> g(m);
                        cv.wait(g, [this] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        fn = std::move(q.front());
                        q.pop_front();
                    }
                    fn();
                }
            });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void submit(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> g(m);
            q.push_back(std::move(fn));
        }
        cv.notify_one();
    }
};

// zlib on the whole file, concatenated members, zero padding behind the last one tolerated: 0 ok, 1 error
int zlib_all(const std::vector<uint8_t>& comp, std::vector<uint8_t>& out) {
    out.clear();
    size_t at = 0;
    bool any = false;
    while (at < comp.size()) {
        bool zeros = true;
        for (size_t i = at; zeros && i < comp.size(); ++i) zeros = comp[i] == 0;
        if (zeros && any) return 0;
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 31) != Z_OK) return 1;
        zs.next_in = const_cast<Bytef*>(comp.data() + at);
        zs.avail_in = (uInt)(comp.size() - at);
        int r = Z_OK;
        uint8_t buf[1 << 16];
        while (r == Z_OK) {
            zs.next_out = buf;
            zs.avail_out = sizeof buf;
            r = inflate(&zs, Z_NO_FLUSH);
            out.insert(out.end(), buf, buf + (sizeof buf - zs.avail_out));
            if (r == Z_BUF_ERROR) break;
        }
        const size_t used = comp.size() - at - zs.avail_in;
        inflateEnd(&zs);
        if (r != Z_STREAM_END) return 1;
        at += used;
        any = true;
    }
    return any || comp.empty() ? 0 : 1;
}

int ours(const std::vector<uint8_t>& comp, size_t chunk, Workers& w, std::vector<uint8_t>& out, std::string& err) {
    qdpgz::Options o;
    o.chunk_bytes = chunk;
    o.in_flight = 6;
    qdpgz::Gunzip gz(comp.data(), comp.size(), o, [&w](std::function<void()> fn) { w.submit(std::move(fn)); });
    out.clear();
    for (;;) {
        std::shared_ptr<qdpgz::Text> t;
        const int rc = gz.next(&t);
        if (rc == 0) return 0;
        if (rc < 0) {
            err = gz.error();
            return 1;
        }
        out.insert(out.end(), t->data, t->data + t->len);
    }
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    const std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int rounds = atoi(argv[2]);
    const size_t chunk = (size_t)atol(argv[3]);
    Workers w(4);
    std::mt19937_64 rng(12345);
    std::vector<uint8_t> ref, got;
    std::string err;
    if (zlib_all(good, ref) != 0) return 3;
    if (ours(good, chunk, w, got, err) != 0 || got != ref) {
        fprintf(stderr, "undamaged file: %s\n", err.c_str());
        return 4;
    }
    int accepted = 0, refused = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<uint8_t> c = good;
        const int kind = (int)(rng() % 6);
        const size_t at = (size_t)(rng() % c.size());
        if (kind == 0) c[at] ^= (uint8_t)(1u << (rng() % 8));
        else if (kind == 1) c.resize(at);
        else if (kind == 2) for (size_t i = at; i < c.size() && i < at + 1 + rng() % 2000; ++i) c[i] = (uint8_t)rng();
        else if (kind == 3) for (size_t i = at; i < c.size() && i < at + 1 + rng() % 5000; ++i) c[i] = 0;
        else if (kind == 4) c.insert(c.begin() + (long)at, (size_t)(1 + rng() % 64), (uint8_t)rng());
        else c.insert(c.end(), (size_t)(1 + rng() % 300), (uint8_t)(rng() % 2 ? 0 : rng()));
        const int zr = zlib_all(c, ref);
        const int orc = ours(c, chunk, w, got, err);
        if (zr == 0 && (orc != 0 || got != ref)) {
            fprintf(stderr, "round %d kind %d at %zu: zlib accepts %zu bytes, the inflater %s (%zu bytes)\n", r, kind, at, ref.size(),
                    orc ? err.c_str() : "differs", got.size());
            return 5;
        }
        if (orc == 0 && (zr != 0 || got != ref)) {
            fprintf(stderr, "round %d kind %d at %zu: the inflater accepts %zu bytes, zlib %s\n", r, kind, at, got.size(), zr ? "refuses" : "differs");
            return 6;
        }
        if (orc == 0) ++accepted; else ++refused;
    }
    printf("rounds %d: %d accepted, %d refused, all as zlib\n", rounds, accepted, refused);
    return 0;
}
