// The seams of the host I/O that the device chunk pipeline builds on (quade_amd/csrc/quade_io_internal.h), without a GPU: the raw
// text reader on every input flavour (plain, one gzip member, many members, BGZF, BGZF that turns into ordinary members, a start
// offset), host inflate of BGZF members, the sink's file access (lazy creation, ordered appends from several threads), the pool.
// Built with -fsanitize=address,undefined (and thread) and run by tests/test_host_text.py.
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/quade_hip.h"
#include "../../quade_amd/csrc/quade_io_internal.h"

static int fails = 0;
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                          \
        }                                                     \
    } while (0)

static std::string read_raw(const std::string& path, int64_t start, bool* ok) {
    std::string err, out;
    qd_reader* r = qdio::raw_open(path.c_str(), start, &err);
    *ok = r != nullptr;
    if (!r) return out;
    for (;;) {
        const uint8_t* p = nullptr;
        size_t n = 0;
        const int rc = qdio::raw_next(r, &p, &n, &err);
        if (rc < 0) *ok = false;
        if (rc <= 0) break;
        out.append((const char*)p, n);
    }
    qdio::raw_close(r);
    return out;
}

static std::string slurp(const std::string& path) {
    std::string s;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return s;
    char buf[65536];
    size_t g;
    while ((g = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, g);
    fclose(f);
    return s;
}

static std::string gunzip_all(const std::string& comp) {  // every member, with zlib
    std::string out;
    size_t pos = 0;
    while (pos < comp.size()) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return out;
        zs.next_in = (Bytef*)comp.data() + pos;
        zs.avail_in = (uInt)(comp.size() - pos);
        char buf[1 << 16];
        int zr = Z_OK;
        while (zr == Z_OK) {
            zs.next_out = (Bytef*)buf;
            zs.avail_out = sizeof buf;
            zr = inflate(&zs, Z_NO_FLUSH);
            out.append(buf, sizeof buf - zs.avail_out);
        }
        pos = comp.size() - zs.avail_in;
        inflateEnd(&zs);
        if (zr != Z_STREAM_END) break;
    }
    return out;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::string text;
    for (int i = 0; i < 60000; ++i) {
        char b[200];
        snprintf(b, sizeof b, "@r%d:%d d\nACGTACGTACGTACGTACGT%dACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII%dIIIIIIII\n", i, i * 7, i % 10, i % 10);
        text += b;
    }
    const uint8_t* t = (const uint8_t*)text.data();
    bool ok;
    // plain text
    {
        const std::string p = dir + "/a.fastq";
        FILE* f = fopen(p.c_str(), "wb");
        fwrite(text.data(), 1, text.size(), f);
        fclose(f);
        CHECK(read_raw(p, 0, &ok) == text && ok);
    }
    // one member, members of 64 KiB of text, BGZF
    for (int64_t mb : {(int64_t)0, (int64_t)65536, (int64_t)-1}) {
        const std::string p = dir + "/b.fastq.gz";
        CHECK(qd_write_gzip_file(p.c_str(), t, (int64_t)text.size(), 1, mb) == QD_OK);
        CHECK(read_raw(p, 0, &ok) == text && ok);
        if (mb == -1) {
            // BGZF: block sizes from the headers; whole blocks inflate on the host; a start offset at a block boundary reads the rest
            const std::string comp = slurp(p);
            size_t pos = 0, nblocks = 0, half_off = 0, half_text = 0, text_pos = 0;
            while (pos < comp.size()) {
                const size_t bs = qdio::bgzf_block_size((const uint8_t*)comp.data() + pos, comp.size() - pos);
                CHECK(bs >= 26 && pos + bs <= comp.size());
                if (!bs) break;
                uint32_t isz;
                memcpy(&isz, comp.data() + pos + bs - 4, 4);
                if (!half_off && pos > comp.size() / 2) {
                    half_off = pos;
                    half_text = text_pos;
                }
                text_pos += isz;
                pos += bs;
                ++nblocks;
            }
            CHECK(nblocks > 20 && text_pos == text.size());
            std::vector<uint8_t> out(text.size());
            CHECK(qdio::host_inflate_members((const uint8_t*)comp.data(), comp.size(), out.data(), out.size()));
            CHECK(memcmp(out.data(), text.data(), text.size()) == 0);
            CHECK(!qdio::host_inflate_members((const uint8_t*)comp.data(), comp.size() - 9, out.data(), out.size()));  // truncated
            CHECK(read_raw(p, (int64_t)half_off, &ok) == text.substr(half_text) && ok);
            // BGZF that turns into one ordinary member half way
            const std::string q = dir + "/c.fastq.gz", tail_path = dir + "/tail.gz";
            CHECK(qd_write_gzip_file(tail_path.c_str(), t, 100000, 1, 0) == QD_OK);
            const std::string mixed = comp.substr(0, half_off) + slurp(tail_path);
            FILE* f = fopen(q.c_str(), "wb");
            fwrite(mixed.data(), 1, mixed.size(), f);
            fclose(f);
            CHECK(read_raw(q, 0, &ok) == text.substr(0, half_text) + text.substr(0, 100000) && ok);
            // damaged: an error, not a crash
            std::string bad = comp;
            bad[bad.size() / 3] ^= 0x55;
            f = fopen(q.c_str(), "wb");
            fwrite(bad.data(), 1, bad.size(), f);
            fclose(f);
            (void)read_raw(q, 0, &ok);
            CHECK(!ok);
        }
    }
    CHECK(qdio::crc32(t, text.size()) == (uint32_t)crc32(0L, t, (uInt)text.size()));
    {
        std::vector<uint8_t> m;
        for (int level : {-1, 1, 6}) {
            CHECK(qdio::host_gzip_member(t, 300000, level, &m));
            CHECK(gunzip_all(std::string((const char*)m.data(), m.size())) == text.substr(0, 300000));
        }
    }
    // the sink's files: created at the first call for a destination, appends in call order per file, from pool jobs of several files at once
    {
        const char* names[2] = {"A", "B"};
        qd_sink* s = nullptr;
        CHECK(qd_sink_create(dir.c_str(), 2, names, 1, 1, 1, 1, &s) == QD_OK);
        qd_sink_set_quiet(s, 1);
        const qdio::SinkInfo info = qdio::sink_info(s);
        CHECK(info.level == 1 && info.n_samples == 2 && info.write_pass && info.write_fail && info.write_undet);
        void* files[3] = {qdio::sink_file(s, 0, 0), qdio::sink_file(s, 3, 1), qdio::sink_file(s, QD_CODE_UNDETERMINED, 0)};
        CHECK(files[0] && files[1] && files[2] && qdio::sink_file(s, 0, 0) == files[0]);
        std::vector<uint8_t> member;
        CHECK(qdio::host_gzip_member(t, 5000, 1, &member));
        std::atomic<int> left{3};
        for (int k = 0; k < 3; ++k)
            qdio::pool_submit([&, k] {
                for (int i = 0; i < 40; ++i) qdio::sink_append(s, files[k], member.data(), member.size());
                --left;
            }, k == 1);
        while (left.load()) std::this_thread::yield();
        qdio::sink_account(s, 120, 120, 120 * 5000, 120 * (int64_t)member.size());
        int64_t members = 0, bytes_in = 0, bytes_out = 0, nfiles = 0;
        CHECK(qd_sink_stats(s, &members, &bytes_in, &bytes_out, &nfiles) == QD_OK && members == 120 && nfiles == 6);
        CHECK(qd_sink_close(s) == QD_OK);
        std::string want;
        for (int i = 0; i < 40; ++i) want += text.substr(0, 5000);
        CHECK(gunzip_all(slurp(dir + "/A_pass_R1.fastq.gz")) == want);
        CHECK(gunzip_all(slurp(dir + "/B_fail_R2.fastq.gz")) == want);
        CHECK(gunzip_all(slurp(dir + "/Undetermined_R1.fastq.gz")) == want);
        CHECK(slurp(dir + "/A_pass_R2.fastq.gz").empty());  // created with its destination, nothing routed to it
    }
    CHECK(qdio::pool_size() >= 1);
    printf(fails ? "FAILED %d\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
