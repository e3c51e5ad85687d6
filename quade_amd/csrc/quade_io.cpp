// Host-only I/O side of libquade_hip.so: routed records -> per-destination fastq.gz files.
//
// Replaces, for whole batches, what the reference does per read pair after Sample.FINDER has picked a
// destination (src/Sample.py:74-91 -> src/FastqWriter.py:48-90): name tagging + record formatting
// (qd_format_records), buffering, gzip compression and appending to <name>_R1/_R2.fastq.gz.  What the
// reference fixes is kept: file names (FastqWriter.py:29-31), creation of a destination's two files
// at its first routed pair only, truncating what was there (:55-57, :76-81), gzip members appended
// afterwards (:83-90), input order inside every file.  What changes is granularity and machinery:
//   * a batch is scattered by routing code with one counting pass (stable: input order is kept);
//   * every destination's records are cut into pieces of ~2 MB of text; a piece = one job = format
//     + compress (one gzip member) on a thread pool owned by the library;
//   * members are appended to their file strictly in submission order, whichever job finishes
//     first; files are opened for the append only (no descriptor is held between members, as in
//     the reference, so thousands of destinations do not run into RLIMIT_NOFILE);
//   * deflate is libdeflate's when libdeflate.so.0 can be loaded (dlopen; no link-time dependency),
//     zlib's otherwise -- both write standard gzip members, only decompressed bytes are pinned.
// No GPU calls here: this file builds with plain g++ (sanitizer tests).
#include <dlfcn.h>
#include <immintrin.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/quade_hip.h"
#include "fastq_scan.h"
#include "quade_pgz.h"

namespace {

// ---- libdeflate, bound at run time -----------------------------------------------------------------
struct LibDeflate {
    void* handle = nullptr;
    void* (*alloc_compressor)(int) = nullptr;
    size_t (*gzip_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*gzip_compress_bound)(void*, size_t) = nullptr;
    void (*free_compressor)(void*) = nullptr;
    void* (*alloc_decompressor)() = nullptr;
    int (*gzip_decompress_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
    void (*free_decompressor)(void*) = nullptr;
    size_t (*deflate_compress)(void*, const void*, size_t, void*, size_t) = nullptr;  // raw deflate (BGZF blocks)
    size_t (*deflate_compress_bound)(void*, size_t) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        const char* off = getenv("QUADE_NO_LIBDEFLATE");
        if (off && *off && *off != '0') return;
        handle = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!handle) return;
#define QD_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(handle, name))
        QD_SYM(alloc_compressor, "libdeflate_alloc_compressor");
        QD_SYM(gzip_compress, "libdeflate_gzip_compress");
        QD_SYM(gzip_compress_bound, "libdeflate_gzip_compress_bound");
        QD_SYM(free_compressor, "libdeflate_free_compressor");
        QD_SYM(alloc_decompressor, "libdeflate_alloc_decompressor");
        QD_SYM(gzip_decompress_ex, "libdeflate_gzip_decompress_ex");
        QD_SYM(free_decompressor, "libdeflate_free_decompressor");
        QD_SYM(deflate_compress, "libdeflate_deflate_compress");
        QD_SYM(deflate_compress_bound, "libdeflate_deflate_compress_bound");
        QD_SYM(crc32, "libdeflate_crc32");
#undef QD_SYM
        ok = alloc_compressor && gzip_compress && gzip_compress_bound && free_compressor && alloc_decompressor &&
             gzip_decompress_ex && free_decompressor;
    }
};
LibDeflate& deflate_lib() {
    static LibDeflate L;
    return L;
}

// ---- where the host's CPU time goes: thread-CPU seconds per stage, summed over all threads of the library -----------
// (qd_io_stage_seconds; two clock reads per job or per 4 MB piece -- nothing per record)
enum Stage { ST_INFLATE, ST_DEV_INFLATE, ST_READ, ST_SCAN_COPY, ST_SCAN_LINES, ST_SCAN_RECORDS, ST_SCATTER, ST_FORMAT, ST_CRC, ST_DEFLATE, ST_LANE, ST_APPEND,
             ST_LANE_WALL, ST_LANE_BATCHES, ST_LANE_PIECES, ST_NO_BUFFER,  // (the last three are counts, not seconds)
             ST_COUNT };
const char* const STAGE_NAMES[ST_COUNT] = {"inflate (pool jobs)", "reader device lanes: stage, launch, wait, CRC-32", "read + cut input (reader threads)", "scanner: copy into the batch",
                                           "scanner: newlines", "scanner: records + batch hand-over", "sink: scatter by code",
                                           "sink: format records", "sink: CRC-32", "sink: deflate on the host",
                                           "device lanes: launch, wait, copy members", "file appends",
                                           "device lanes: WALL seconds inside qd_deflater_run", "device lanes: launches (count / 1e9)",
                                           "device lanes: pieces (count / 1e9)", "pieces that found no page-locked buffer (count / 1e9)"};
std::atomic<int64_t> g_stage_ns[ST_COUNT];
inline int64_t thread_cpu_ns() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}
thread_local int g_stage_depth = 0;
struct StageTimer {  // the outermost timer of a thread counts; one opened inside it is part of it
    Stage st;
    int64_t t0 = 0;
    bool active;
    explicit StageTimer(Stage s) : st(s), active(g_stage_depth++ == 0) {
        if (active) t0 = thread_cpu_ns();
    }
    void next(Stage s) {  // closes the running stage, opens another
        if (!active) return;
        const int64_t t = thread_cpu_ns();
        g_stage_ns[st].fetch_add(t - t0, std::memory_order_relaxed);
        st = s;
        t0 = t;
    }
    ~StageTimer() {
        --g_stage_depth;
        if (active) g_stage_ns[st].fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed);
    }
};

}  // namespace

// CRC-32 of a buffer (gzip's): libdeflate's slice-by-N when loaded, zlib's otherwise.  Used by the device inflater
// (quade_api.cpp) to check every BGZF block it inflated.
uint32_t qd_io_crc32(const uint8_t* p, size_t n) {
    LibDeflate& L = deflate_lib();
    if (L.ok && L.crc32) return L.crc32(0, p, n);
    uLong c = crc32(0L, Z_NULL, 0);
    while (n) {
        const uInt k = (uInt)std::min<size_t>(n, 1u << 30);
        c = crc32(c, p, k);
        p += k;
        n -= k;
    }
    return (uint32_t)c;
}

namespace {

// byte buffer that is NOT value-initialised (std::vector<uint8_t>::resize writes every byte first)
struct Bytes {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0, cap = 0;
    uint8_t* data() { return p.get(); }
    const uint8_t* data() const { return p.get(); }
    size_t size() const { return n; }
    void clear() { n = 0; }
    void resize(size_t m) {  // contents are kept when growing
        if (m > cap) {
            std::unique_ptr<uint8_t[]> q(new uint8_t[m]);
            if (n) memcpy(q.get(), p.get(), n);
            p.swap(q);
            cap = m;
        }
        n = m;
    }
};

// ---- level -1: Huffman coding only ---------------------------------------------------------------------------
// One gzip member = one dynamic-Huffman DEFLATE block over the whole piece: literals only, no string matching
// (zlib calls the idea Z_HUFFMAN_ONLY).  A byte histogram, a length-limited Huffman code, one table lookup and a
// shift per byte: ~3 x the speed of libdeflate's level 1 per core, and what it gives away is what matching earns
// on fastq text (mostly in the read names): files ~25 % larger on real data, the same on the synthetic records
// here (random bases and qualities have nothing to match).  Any gunzip reads it.
//
// Code lengths (<= 15 bits) for the used ones of n symbols: Huffman by the two-queue merge over the symbols
// sorted by frequency, then zlib's overflow fix (move leaves up until nothing is deeper than the limit) and
// lengths reassigned in frequency order (the least frequent symbols get the longest codes).
void huffman_lengths(const uint64_t* freq, int n, int max_bits, uint8_t* len) {
    std::vector<std::pair<uint64_t, int>> v;
    for (int s = 0; s < n; ++s) {
        len[s] = 0;
        if (freq[s]) v.emplace_back(freq[s], s);
    }
    for (int s = 0; v.size() < 2 && s < n; ++s)  // a prefix code needs two codes: lend one to an unused symbol
        if (!freq[s]) v.emplace_back(1, s);
    std::sort(v.begin(), v.end());
    const int m = (int)v.size();
    std::vector<uint64_t> w(2 * m - 1);
    std::vector<int> parent(2 * m - 1, -1), depth(2 * m - 1, 0);
    for (int i = 0; i < m; ++i) w[i] = v[i].first;
    int leaf = 0, inner = m, next = m;
    auto take = [&]() {  // the lighter of the next unused leaf / inner node (leaves first on a tie: shallower trees)
        if (leaf < m && (inner >= next || w[leaf] <= w[inner])) return leaf++;
        return inner++;
    };
    while (next < 2 * m - 1) {
        const int a = take(), b = take();
        w[next] = w[a] + w[b];
        parent[a] = parent[b] = next;
        ++next;
    }
    for (int k = 2 * m - 3; k >= 0; --k) depth[k] = depth[parent[k]] + 1;
    std::vector<int> bl(max_bits + 1, 0);
    for (int i = 0; i < m; ++i) ++bl[std::min(depth[i], max_bits)];
    // Leaves deeper than the limit were clamped to it, so the code is over-subscribed by
    // excess = sum bl[d] * 2^(max_bits - d) - 2^max_bits code points.  zlib's repair step -- the deepest leaf above
    // the limit becomes an inner node whose children are itself and one leaf taken from the limit -- frees exactly
    // one code point, so it runs `excess` times.  (Counting the overflow by clamped LEAVES only, as this did, stops
    // too early once the unlimited tree is deeper than max_bits + 1: zlib counts clamped inner nodes too.)
    uint64_t kraft = 0;
    for (int d = 1; d <= max_bits; ++d) kraft += (uint64_t)bl[d] << (max_bits - d);
    for (uint64_t excess = kraft - ((uint64_t)1 << max_bits); excess > 0; --excess) {
        int bits = max_bits - 1;
        while (bl[bits] == 0) --bits;
        --bl[bits];
        bl[bits + 1] += 2;
        --bl[max_bits];
    }
    int at = 0;  // v is ascending by frequency: the longest codes first
    for (int bits = max_bits; bits >= 1; --bits)
        for (int c = 0; c < bl[bits]; ++c) len[v[at++].second] = (uint8_t)bits;
}

struct BitOut {  // LSB-first bit packer over a byte buffer with room to spare
    uint8_t* p;
    uint64_t acc = 0;
    int cnt = 0;
    void put(uint32_t v, int n) {  // n <= 32
        acc |= (uint64_t)v << cnt;
        cnt += n;
        if (cnt >= 32) {
            memcpy(p, &acc, 4);
            p += 4;
            acc >>= 32;
            cnt -= 32;
        }
    }
    uint8_t* finish() {
        while (cnt > 0) {
            *p++ = (uint8_t)acc;
            acc >>= 8;
            cnt -= 8;
        }
        return p;
    }
};

uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// text -> one gzip member holding one dynamic-Huffman block of literals
bool huffman_member(const uint8_t* in, size_t n, Bytes& out) {
    uint64_t freq[257] = {0};
    {
        uint32_t h[4][256];
        memset(h, 0, sizeof h);
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {  // four tables: consecutive equal bytes do not wait for each other's increment
            ++h[0][in[i]];
            ++h[1][in[i + 1]];
            ++h[2][in[i + 2]];
            ++h[3][in[i + 3]];
        }
        for (; i < n; ++i) ++h[0][in[i]];
        for (int s = 0; s < 256; ++s) freq[s] = (uint64_t)h[0][s] + h[1][s] + h[2][s] + h[3][s];
    }
    freq[256] = 1;  // end of block
    uint8_t len[257];
    huffman_lengths(freq, 257, 15, len);
    {  // a literal/length code must be complete (inflate refuses over- and under-subscribed sets): never emit another
        uint32_t kraft = 0;
        for (int s = 0; s < 257; ++s)
            if (len[s]) kraft += 1u << (15 - len[s]);
        if (kraft != (1u << 15)) return false;
    }
    uint32_t code[257];  // bit-reversed canonical codes: DEFLATE sends Huffman codes most significant bit first
    {
        int bl[16] = {0}, next[16] = {0};
        for (int s = 0; s < 257; ++s) ++bl[len[s]];
        bl[0] = 0;
        for (int b = 1, c = 0; b <= 15; ++b) {
            c = (c + bl[b - 1]) << 1;
            next[b] = c;
        }
        for (int s = 0; s < 257; ++s) code[s] = len[s] ? reverse_bits((uint32_t)next[len[s]]++, len[s]) : 0;
    }
    uint64_t bits = 3 + 14 + 19 * 3 + (257 + 2) * 4;
    for (int s = 0; s < 257; ++s) bits += freq[s] * len[s];
    out.resize(10 + (size_t)(bits / 8) + 16 + 8);
    uint8_t* p = out.data();
    const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    memcpy(p, head, 10);
    BitOut bo{p + 10};
    bo.put(1, 1);       // BFINAL
    bo.put(2, 2);       // dynamic Huffman
    bo.put(0, 5);       // HLIT: 257 literal/length codes
    bo.put(1, 5);       // HDIST: 2 distance codes (both of length 1: a complete code that is never used)
    bo.put(15, 4);      // HCLEN: all 19 code-length-code lengths follow
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int k = 0; k < 19; ++k) bo.put(order[k] < 16 ? 4 : 0, 3);  // lengths 0..15 as 4-bit codes, no run-length symbols
    for (int s = 0; s < 257; ++s) bo.put(reverse_bits(len[s], 4), 4);
    bo.put(reverse_bits(1, 4), 4);
    bo.put(reverse_bits(1, 4), 4);
    for (size_t i = 0; i < n; ++i) bo.put(code[in[i]], len[in[i]]);
    bo.put(code[256], len[256]);
    p = bo.finish();
    const uint32_t crc = qd_io_crc32(in, n), isize = (uint32_t)n;
    memcpy(p, &crc, 4);
    memcpy(p + 4, &isize, 4);
    out.resize((size_t)(p + 8 - out.data()));
    return true;
}

// one gzip member of `n` bytes at `level` (-1 = Huffman only, 0..9) -> out; false on failure
bool gzip_member(const uint8_t* in, size_t n, int level, Bytes& out) {
    if (level < 0) return huffman_member(in, n, out);
    LibDeflate& L = deflate_lib();
    if (L.ok) {
        thread_local std::map<int, void*> comp;  // one compressor per (thread, level), kept for the thread's life
        void*& c = comp[level];
        if (!c) c = L.alloc_compressor(level);
        if (c) {
            out.resize(L.gzip_compress_bound(c, n));
            const size_t w = L.gzip_compress(c, in, n, out.data(), out.size());
            if (w) {
                out.resize(w);
                return true;
            }
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)n) + 64);
    size_t done_in = 0, done_out = 0;
    int r = Z_OK;
    while (r != Z_STREAM_END) {  // uInt-sized windows: members here are a few MB, but stay correct beyond 4 GB
        const size_t ci = std::min<size_t>(n - done_in, 1u << 30), co = std::min<size_t>(out.size() - done_out, 1u << 30);
        zs.next_in = const_cast<Bytef*>(in + done_in);
        zs.avail_in = (uInt)ci;
        zs.next_out = out.data() + done_out;
        zs.avail_out = (uInt)co;
        r = deflate(&zs, done_in + ci == n ? Z_FINISH : Z_NO_FLUSH);
        if (r != Z_OK && r != Z_STREAM_END && r != Z_BUF_ERROR) {
            deflateEnd(&zs);
            return false;
        }
        done_in += ci - zs.avail_in;
        done_out += co - zs.avail_out;
        if (done_out == out.size()) out.resize(out.size() * 2);
    }
    deflateEnd(&zs);
    out.resize(done_out);
    return true;
}

// ---- thread pool owned by the library --------------------------------------------------------------
class Pool {
  public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void submit(std::function<void()> fn, bool urgent = false) {
        {
            std::lock_guard<std::mutex> g(m_);
            if (urgent)
                uq_.push_back(std::move(fn));  // inflate jobs: everything downstream waits for their text.  Ahead of the
            else                               // other work, but in order among themselves: a reader collects its chunks
                q_.push_back(std::move(fn));   // oldest first, and a newest-first queue starved exactly the one it waited for
        }
        cv_.notify_one();
    }
    int size() const { return (int)threads_.size(); }

  private:
    void run() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || !q_.empty() || !uq_.empty(); });
                if (q_.empty() && uq_.empty()) return;  // stop_ and drained
                std::deque<std::function<void()>>& from = uq_.empty() ? q_ : uq_;
                fn = std::move(from.front());
                from.pop_front();
            }
            fn();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> q_, uq_;  // ordinary jobs; urgent ones (served first, first in first out)
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

std::mutex g_pool_mutex;
std::unique_ptr<Pool> g_pool;
int g_pool_threads = 0;  // 0 = one per core this process may use

// cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container
// on a 256-thread host is often given a few cores' worth of time: 256 runnable gzip threads would
// then starve the reader threads they are fed by)
int available_cores() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && (n < 1 || a < n)) n = a;
    }
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[64];
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");  // cgroup v1
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {
        } else {
            quota = -1;
        }
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) {
        const int c = (int)((quota + period - 1) / period);
        if (c > 0 && c < n) n = c;
    }
    return n < 1 ? 1 : n;
}

Pool& pool() {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    if (!g_pool) g_pool.reset(new Pool(g_pool_threads > 0 ? g_pool_threads : available_cores()));
    return *g_pool;
}

struct Latch {
    std::mutex m;
    std::condition_variable cv;
    int64_t n = 0;
    void done() {
        std::lock_guard<std::mutex> g(m);
        if (--n == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [this] { return n == 0; });
    }
};

struct OutFile {
    std::string path;
    std::mutex m;
    uint64_t next_submit = 0, next_write = 0;
    std::map<uint64_t, Bytes> done;  // finished members waiting for their turn
};

struct Dest {
    OutFile f[2];  // R1, R2
};

}  // namespace

struct qd_sink {
    std::string outdir;
    int level = 6;
    bool write_pass = true, write_fail = true, write_undet = true;
    std::vector<std::string> names;                 // sample names, ordinal order
    std::map<uint32_t, std::unique_ptr<Dest>> dest;  // routing code -> its two files (created lazily)
    std::mutex m;                                    // guards the fields below
    std::condition_variable cv;
    int64_t pending_jobs = 0, pending_bytes = 0;
    std::string err;
    int64_t members = 0, bytes_in = 0, bytes_out = 0;
    int64_t device_members = 0;  // of `members`: made by the GPU (gzip_level -1 with a deflate device)
    int deflate_device = -1;     // >= 0: Huffman-only members are made on that device while page-locked buffers last
    int quiet = 0;
};

namespace {

constexpr int64_t JOB_BYTES = 2 << 20;            // text per gzip member
constexpr int64_t PENDING_LIMIT = (int64_t)1 << 30;  // formatted + compressed bytes allowed in flight per sink

void sink_error(qd_sink* s, const std::string& msg) {
    std::lock_guard<std::mutex> g(s->m);
    if (s->err.empty()) s->err = msg;
}

bool append_file(const std::string& path, const uint8_t* data, size_t size, std::string& why) {
    StageTimer timer(ST_APPEND);
    const int fd = open(path.c_str(), O_WRONLY | O_APPEND | O_CREAT | O_CLOEXEC, 0644);
    if (fd < 0) {
        why = path + ": " + strerror(errno);
        return false;
    }
    size_t off = 0;
    while (off < size) {
        const ssize_t w = write(fd, data + off, size - off);
        if (w < 0) {
            if (errno == EINTR) continue;
            why = path + ": " + strerror(errno);
            close(fd);
            return false;
        }
        off += (size_t)w;
    }
    if (close(fd) != 0) {
        why = path + ": " + strerror(errno);
        return false;
    }
    return true;
}
bool append_file(const std::string& path, const Bytes& data, std::string& why) { return append_file(path, data.data(), data.size(), why); }

// a finished member takes its place in the file's queue; everything that is next in line is written
void deliver(qd_sink* s, OutFile* f, uint64_t seq, Bytes&& member) {
    std::lock_guard<std::mutex> g(f->m);
    f->done.emplace(seq, std::move(member));
    for (auto it = f->done.find(f->next_write); it != f->done.end(); it = f->done.find(f->next_write)) {
        std::string why;
        if (!append_file(f->path, it->second, why)) sink_error(s, why);
        f->done.erase(it);
        ++f->next_write;
    }
}

struct Piece {  // records order[lo..hi) of one read file of one destination
    OutFile* f;
    uint64_t seq;
    const uint8_t* text;
    const int64_t* rec_off;
    const int64_t* sel;
    int64_t n_sel, text_bytes;
};

}  // namespace

// Huffman-only members on the device (quade_api.cpp / quade_deflate.hip); weak: this file also builds without the HIP half
extern "C" {
int qd_deflater_create(int device_id, qd_deflater** out) __attribute__((weak));
int qd_deflater_run(qd_deflater* deflater, int32_t n_pieces, const uint8_t* const* text, const int64_t* text_len, const uint32_t* crc32,
                    int32_t text_pinned, uint8_t* out, int64_t out_stride, int64_t* member_len) __attribute__((weak));
int qd_deflater_set_level(qd_deflater* deflater, int32_t level) __attribute__((weak));
int qd_deflater_destroy(qd_deflater* deflater) __attribute__((weak));
int64_t qd_huffman_member_bound(int64_t text_len) __attribute__((weak));
void* qd_pinned_alloc(int64_t bytes) __attribute__((weak));
void qd_pinned_free(void* p) __attribute__((weak));
}

namespace {

// A formatted piece on its way through the device: the text sits in a page-locked buffer of the service.
struct DevPiece {
    qd_sink* s;
    OutFile* f;
    uint64_t seq;
    int64_t text_bytes;  // what the sink's back-pressure counted for this piece
    uint8_t* text;
    size_t cap;
    int64_t len;
    uint32_t crc;
    int level;  // of its sink: -1 (Huffman only) or 1 (LZ77 + Huffman), the two the device implements
};

// the levels whose members a deflate device makes (qd_deflater_set_level)
inline bool device_level(int level) { return level == -1 || level == 1; }

void finish_piece(qd_sink* s, OutFile* f, uint64_t seq, int64_t text_bytes, int64_t w, Bytes&& member, bool on_device) {
    const int64_t out_bytes = (int64_t)member.size();
    deliver(s, f, seq, std::move(member));
    std::lock_guard<std::mutex> g(s->m);
    s->pending_bytes -= text_bytes;
    --s->pending_jobs;
    ++s->members;
    if (on_device) ++s->device_members;
    s->bytes_in += w > 0 ? w : 0;
    s->bytes_out += out_bytes;
    s->cv.notify_all();
}

// One per device, for the life of the process: two lanes (threads that sleep on the device, each with its own
// deflater: one batch uploads and codes while the other's members come back) and the page-locked text buffers.
// Pool jobs format a piece straight into such a buffer and queue it here; when no buffer is free the job codes its
// piece on its own core as before -- the host and the device share the work by whoever is free.
std::atomic<int64_t> g_test_deflate_fail_after{-1}, g_test_inflate_fail_after{-1};  // qd_io_set_option "test_*_fail_after" (tests only)

class DeflateService {
  public:
    static constexpr size_t BUF_BYTES = (size_t)JOB_BYTES + (JOB_BYTES >> 2) + (256u << 10);  // a piece, its tags, slack
    static constexpr int MAX_BATCH = 32;
    // page-locked buffers at most (made as the lanes come up and as pieces ask for more) and how long a pool job waits for one:
    // the defaults, or QUADE_DEFLATE_BUFFERS / QUADE_DEFLATE_BUFFER_WAIT_MS (measurement knobs)
    const int lanes_n_ = [] {
        const char* e = getenv("QUADE_DEFLATE_LANES");
        const int v = e && *e ? atoi(e) : LANES;
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    const int MAX_BUFS = [this] {
        const char* e = getenv("QUADE_DEFLATE_BUFFERS");
        const int v = e && *e ? atoi(e) : 96 * lanes_n_;  // (what a lane makes when it comes up: 12 + 36 + 48)
        return v < 16 ? 16 : (v > 4096 ? 4096 : v);
    }();
    const int wait_ms_ = [] {
        const char* e = getenv("QUADE_DEFLATE_BUFFER_WAIT_MS");
        const int v = e && *e ? atoi(e) : 0;  // (12 ms: 13 % -> 1 % of the pieces coded by the host, 0.1 core-s per M pairs saved -- and
        return v < 0 ? 0 : (v > 1000 ? 1000 : v);  //  the run no faster: 6.6 vs 7.2 M pairs/s, profiles/r03_e2e_deflate_buffers_ab.txt)
    }();
    explicit DeflateService(int device) : device_(device) {
        for (int i = 0; i < lanes_n_; ++i) lanes_.emplace_back([this] { lane(); });
    }
    // a buffer of BUF_BYTES, or nullptr (none free right now: the caller codes its piece itself).  Buffers are made
    // by the lanes, a slab at a time, off the pool threads' path (page-locking 120 MB takes tens of milliseconds).
    // A routed batch is ~400 pieces at once, more than the lanes drain while the pool formats them: ~10 % of the pieces find
    // no buffer and are coded by libdeflate on their pool thread (the host and the device share the work by whoever is free).
    // A job can wait for a buffer instead (QUADE_DEFLATE_BUFFER_WAIT_MS); measured, that saves CPU and no time.
    uint8_t* take_buffer() {
        std::unique_lock<std::mutex> g(m_);
        if (!failed_ && free_.empty()) {
            if (made_ < MAX_BUFS) {
                want_slab_ = true;
                cv_.notify_one();
            }
            if (wait_ms_ > 0) cv_free_.wait_for(g, std::chrono::milliseconds(wait_ms_), [this] { return failed_ || !free_.empty(); });
        }
        if (failed_ || free_.empty()) return nullptr;
        uint8_t* p = free_.back();
        free_.pop_back();
        if (free_.size() < 8 && made_ < MAX_BUFS) {
            want_slab_ = true;
            cv_.notify_one();
        }
        return p;
    }
    void give_buffer(uint8_t* p) {
        {
            std::lock_guard<std::mutex> g(m_);
            free_.push_back(p);
        }
        cv_free_.notify_one();
    }
    void submit(const DevPiece& d) {
        {
            std::lock_guard<std::mutex> g(m_);
            q_.push_back(d);
        }
        cv_.notify_one();
    }

  private:
    void lane() {
        qd_deflater* def = nullptr;
        bool usable = qd_deflater_create && qd_deflater_run && qd_huffman_member_bound && qd_deflater_create(device_, &def) == QD_OK;
        int64_t batches = 0;
        if (usable) add_slab(12);  // a first small slab; more are made as pieces ask for them (want_slab_), up to MAX_BUFS
        else lane_lost();          // (no deflater on this lane: when none is left the pool's jobs stop asking for buffers)
        for (;;) {
            std::vector<DevPiece> b;
            bool slab = false;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this, usable] { return !q_.empty() || (want_slab_ && usable); });  // (only a lane that can use buffers makes them)
                while (!q_.empty() && (int)b.size() < MAX_BATCH && (b.empty() || q_.front().level == b[0].level)) {  // one level per launch
                    b.push_back(q_.front());
                    q_.pop_front();
                }
                if (b.empty() && want_slab_) {
                    want_slab_ = false;
                    slab = usable;
                }
            }
            if (slab) add_slab();
            if (b.empty()) continue;
            std::vector<const uint8_t*> tp(b.size());
            std::vector<int64_t> tl(b.size()), ml(b.size(), 0);
            std::vector<uint32_t> crc(b.size());
            int64_t longest = 0;
            for (size_t i = 0; i < b.size(); ++i) {
                tp[i] = b[i].text;
                tl[i] = b[i].len;
                crc[i] = b[i].crc;
                longest = std::max(longest, b[i].len);
            }
            const int64_t fail_after = g_test_deflate_fail_after.load();  // test option: the device "fails" after this many batches of a lane
            bool ok = usable && !(fail_after >= 0 && batches >= fail_after);
            int64_t stride = 0;
            StageTimer lane_timer(ST_LANE);
            if (ok && b[0].level != -1) ok = qd_deflater_set_level && qd_deflater_set_level(def, b[0].level) == QD_OK;
            else if (ok && qd_deflater_set_level) (void)qd_deflater_set_level(def, -1);
            std::shared_ptr<MemberBlock> out;
            if (ok) {
                stride = qd_huffman_member_bound(longest);
                out = take_block((size_t)stride * b.size());
                const auto w0 = std::chrono::steady_clock::now();
                ok = out->p && qd_deflater_run(def, (int32_t)b.size(), tp.data(), tl.data(), crc.data(), 1, out->p, stride, ml.data()) == QD_OK;
                g_stage_ns[ST_LANE_WALL].fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count());
                g_stage_ns[ST_LANE_BATCHES].fetch_add(1);
                g_stage_ns[ST_LANE_PIECES].fetch_add((int64_t)b.size());
                ++batches;
                if (!ok) {  // a HIP error: the host takes over from here (pool jobs stop asking for buffers)
                    usable = false;
                    {
                        std::lock_guard<std::mutex> g(m_);
                        failed_ = true;
                    }
                    cv_free_.notify_all();
                    cv_.notify_all();  // (a slab request this lane was about to serve goes to a lane that still can)
                }
            }
            // The members go to their files on the pool's threads (a copy out of the batch's block, the appends in file
            // order): a lane that delivered its 32 members itself spent more of a batch's ~27 ms in write() than waiting
            // for the device, two lanes made 4.8 GB/s of text, and the pieces that found no buffer meanwhile were coded by
            // the host (profiles/r03_e2e_16m_level1_stages.txt: 0.17-0.29 core-s per M pairs of fallback).
            for (size_t i = 0; i < b.size(); ++i) {
                const DevPiece pc = b[i];
                if (ok && ml[i] > 0) {
                    give_buffer(pc.text);
                    const size_t at = (size_t)stride * i, len = (size_t)ml[i];
                    pool().submit([out, at, len, pc] {
                        StageTimer timer(ST_APPEND);
                        Bytes member;
                        member.resize(len);
                        memcpy(member.data(), out->p + at, len);
                        finish_piece(pc.s, pc.f, pc.seq, pc.text_bytes, pc.len, std::move(member), true);
                    });
                } else {  // the device failed, or this member did not fit its slot: the host's coder
                    pool().submit([this, pc] {
                        Bytes member;
                        {
                            StageTimer timer(ST_DEFLATE);
                            if (!gzip_member(pc.text, (size_t)pc.len, pc.level, member)) {
                                sink_error(pc.s, "gzip compression failed");
                                member.clear();
                            }
                        }
                        give_buffer(pc.text);
                        finish_piece(pc.s, pc.f, pc.seq, pc.text_bytes, pc.len, std::move(member), false);
                    });
                }
            }
        }
    }
    void lane_lost() {
        std::lock_guard<std::mutex> g(m_);
        if (++lanes_lost_ >= lanes_n_) failed_ = true;
        cv_free_.notify_all();
    }
    int lanes_lost_ = 0;
    // blocks of members (one per batch), recycled: a fresh 75 MB allocation per batch is 18 000 page faults
    struct MemberBlock {
        uint8_t* p = nullptr;
        size_t cap = 0;
        ~MemberBlock() { free(p); }
    };
    std::shared_ptr<MemberBlock> take_block(size_t need) {
        MemberBlock* mb = nullptr;
        {
            std::lock_guard<std::mutex> g(m_);
            if (!blocks_.empty()) {
                mb = blocks_.back();
                blocks_.pop_back();
            }
        }
        if (!mb) mb = new MemberBlock();
        if (mb->cap < need) {
            free(mb->p);
            mb->p = (uint8_t*)malloc(need);
            mb->cap = mb->p ? need : 0;
        }
        return std::shared_ptr<MemberBlock>(mb, [this](MemberBlock* x) {
            std::lock_guard<std::mutex> g(m_);
            if (blocks_.size() < 8) blocks_.push_back(x);
            else delete x;
        });
    }
    std::vector<MemberBlock*> blocks_;
    // `count` more page-locked buffers, cut from one allocation (kept for the life of the process).  Page-locking costs
    // ~1 ms per MB: a lane's first slab is a small one, so that the first pieces of a run find buffers
    void add_slab(int count = SLAB) {
        {
            std::lock_guard<std::mutex> g(m_);
            if (made_ >= MAX_BUFS) return;
            count = std::min(count, MAX_BUFS - made_);
            made_ += count;
        }
        uint8_t* p = qd_pinned_alloc ? (uint8_t*)qd_pinned_alloc((int64_t)(BUF_BYTES * (size_t)count)) : nullptr;
        std::lock_guard<std::mutex> g(m_);
        if (!p) {
            made_ = MAX_BUFS;  // no more page-locked memory to be had: work with what there is
            return;
        }
        for (int i = 0; i < count; ++i) free_.push_back(p + (size_t)i * BUF_BYTES);
        cv_free_.notify_all();
    }
    static constexpr int SLAB = 48, LANES = 6;  // (with the inflater on the same device the coder's launches take turns with its workgroups:
                                                //  6 lanes against 3 = a third fewer pieces left to the pool, 10-20 % less CPU, the rate level or up
                                                //  to 7 % better: profiles/r03_e2e_deflate_lanes_ab.txt)
    int device_;
    std::mutex m_;
    std::condition_variable cv_, cv_free_;
    std::deque<DevPiece> q_;
    std::vector<uint8_t*> free_;
    int made_ = 0;
    bool failed_ = false, want_slab_ = false;
    std::vector<std::thread> lanes_;
};

std::mutex g_deflate_services_mutex;
std::map<int, DeflateService*> g_deflate_services;  // never destroyed: their lanes run until the process ends
DeflateService* deflate_service(int device) {
    std::lock_guard<std::mutex> g(g_deflate_services_mutex);
    DeflateService*& sv = g_deflate_services[device];
    if (!sv) sv = new DeflateService(device);
    return sv;
}

}  // namespace

extern "C" {

/* Thread-CPU seconds per stage of the host I/O since the process started (or since the last call with reset != 0), summed over
 * all threads of the library: names[i] -> seconds[i], i < the return value (<= cap). */
int qd_io_stage_seconds(const char** names, double* seconds, int32_t cap, int32_t reset) {
    int n = 0;
    for (; n < ST_COUNT && n < cap; ++n) {
        if (names) names[n] = STAGE_NAMES[n];
        if (seconds) seconds[n] = (double)g_stage_ns[n].load(std::memory_order_relaxed) * 1e-9;
    }
    if (reset)
        for (int i = 0; i < ST_COUNT; ++i) g_stage_ns[i].store(0, std::memory_order_relaxed);
    return n;
}

int qd_io_threads(int32_t n_threads) {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    if (n_threads >= 0 && !g_pool) g_pool_threads = n_threads;
    if (g_pool) return g_pool->size();
    return g_pool_threads > 0 ? g_pool_threads : available_cores();
}

int qd_host_cores(void) { return available_cores(); }

int qd_io_backend(void) { return deflate_lib().ok ? 1 : 0; }

int qd_sink_create(const char* outdir, int32_t n_samples, const char* const* names, int32_t gzip_level,
                   int32_t write_pass, int32_t write_fail, int32_t write_undetermined, qd_sink** out) {
    if (!outdir || n_samples < 0 || (n_samples > 0 && !names) || gzip_level < -1 || gzip_level > 9 || !out)
        return QD_ERR_INVALID;
    qd_sink* s = new qd_sink();
    s->outdir = outdir;
    s->level = gzip_level;
    s->write_pass = write_pass != 0;
    s->write_fail = write_fail != 0;
    s->write_undet = write_undetermined != 0;
    for (int i = 0; i < n_samples; ++i) {
        if (!names[i]) {
            delete s;
            return QD_ERR_INVALID;
        }
        s->names.emplace_back(names[i]);
    }
    *out = s;
    return QD_OK;
}

const char* qd_sink_last_error(const qd_sink* s) { return s ? s->err.c_str() : "sink is NULL"; }

int qd_sink_set_device_deflate(qd_sink* s, int32_t device_id) {
    if (!s) return QD_ERR_INVALID;
    if (device_id >= 0 && !(qd_deflater_create && qd_deflater_run && qd_huffman_member_bound && qd_pinned_alloc)) return QD_ERR_NO_DEVICE;
    s->deflate_device = device_id;
    if (device_id >= 0 && device_level(s->level)) (void)deflate_service(device_id);  // its lanes come up (deflaters, first buffers) while the readers start
    return QD_OK;
}

int qd_sink_device_members(qd_sink* s, int64_t* device_members) {
    if (!s || !device_members) return QD_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->m);
    *device_members = s->device_members;
    return QD_OK;
}

int qd_sink_set_quiet(qd_sink* s, int32_t quiet) {
    if (!s) return QD_ERR_INVALID;
    s->quiet = quiet;
    return QD_OK;
}

// The two files of routing code `code`: made (truncated, announced) at the destination's first routed pair
// (src/FastqWriter.py:55-57, 76-81).  nullptr: a file could not be created (the sink's error is set).
static Dest* sink_dest(qd_sink* s, uint32_t code) {
    auto it = s->dest.find(code);
    if (it != s->dest.end()) return it->second.get();
    const bool undet = code == QD_CODE_UNDETERMINED;
    std::unique_ptr<Dest> d(new Dest());
    const std::string base = s->outdir + "/" + (undet ? std::string("Undetermined") : s->names[code >> 1] + ((code & 1) ? "_fail" : "_pass"));
    d->f[0].path = base + "_R1.fastq.gz";
    d->f[1].path = base + "_R2.fastq.gz";
    for (int k = 0; k < 2; ++k) {
        if (!s->quiet) {
            printf("\tCreate %s file\n", d->f[k].path.c_str());
        }
        const int fd = open(d->f[k].path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        if (fd < 0) {
            sink_error(s, d->f[k].path + ": " + strerror(errno));
            return nullptr;
        }
        close(fd);
    }
    if (!s->quiet) fflush(stdout);
    return s->dest.emplace(code, std::move(d)).first->second.get();
}

// what a batch's jobs need until the last of them has formatted its piece
struct RouteRes {
    std::vector<int64_t> order;
    std::vector<uint8_t> tags, tag_len;  // copies (owned mode only)
    void* owned[2] = {nullptr, nullptr};  // text batches of the reader handed over by the caller
    Latch formatted;
    ~RouteRes() {
        for (void* h : owned)
            if (h) qd_text_batch_free(h);
    }
};

// owned == false: returns when every piece has been formatted (the caller's buffers are free again);
// owned == true: the text batches, and copies of the tags, belong to the jobs -- returns after the scatter
static int route_impl(qd_sink* s, int64_t n, const uint16_t* codes, const uint8_t* r1_text, const int64_t* r1_off,
                      const uint8_t* r2_text, const int64_t* r2_off, const uint8_t* tag_rows, int32_t tag_stride,
                      const uint8_t* tag_len, void* own1, void* own2) {
    const bool owned = own1 != nullptr || own2 != nullptr;
    std::shared_ptr<RouteRes> res = std::make_shared<RouteRes>();
    res->owned[0] = own1;  // from here on the handles are freed with `res`, whatever happens
    res->owned[1] = own2;
    if (!s || n < 0) return QD_ERR_INVALID;
    if (n == 0) return QD_OK;
    if (!codes || !r1_text || !r1_off || !r2_text || !r2_off || !tag_rows || !tag_len) return QD_ERR_INVALID;
    {
        std::lock_guard<std::mutex> g(s->m);
        if (!s->err.empty()) return QD_ERR_FORMAT;
    }
    if (owned) {
        res->tags.assign(tag_rows, tag_rows + (size_t)n * (size_t)tag_stride);
        res->tag_len.assign(tag_len, tag_len + n);
        tag_rows = res->tags.data();
        tag_len = res->tag_len.data();
    }
    // 1. counting scatter by routing code (src/Sample.py:74-91 decides per pair; here per batch), stable
    std::unique_ptr<StageTimer> scatter_timer(new StageTimer(ST_SCATTER));
    const uint32_t S = (uint32_t)s->names.size(), nb = 2 * S + 1;
    std::vector<int64_t> start(nb + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t c = codes[i];
        const uint32_t b = c == QD_CODE_UNDETERMINED ? 2 * S : c;
        if (b >= nb) {
            sink_error(s, "routing code beyond the sample table");
            return QD_ERR_INVALID;
        }
        ++start[b + 1];
    }
    for (uint32_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    std::vector<int64_t>& order = res->order;
    order.resize((size_t)n);
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t c = codes[i];
        order[(size_t)fill[c == QD_CODE_UNDETERMINED ? 2 * S : c]++] = i;
    }
    // 2. pieces: per destination and read file, runs of records worth ~JOB_BYTES of text
    std::vector<Piece> pieces;
    for (uint32_t b = 0; b < nb; ++b) {
        const int64_t lo = start[b], hi = start[b + 1];
        if (lo == hi) continue;
        const bool undet = b == 2 * S;
        if (undet ? !s->write_undet : ((b & 1) ? !s->write_fail : !s->write_pass)) continue;  // counters moved on the device
        const uint32_t code = undet ? QD_CODE_UNDETERMINED : b;
        Dest* dest_of_code = sink_dest(s, code);
        if (!dest_of_code) return QD_ERR_FORMAT;
        for (int k = 0; k < 2; ++k) {
            const int64_t* off = k ? r2_off : r1_off;
            int64_t a = lo;
            while (a < hi) {
                int64_t e = a, bytes = 0;
                while (e < hi && bytes < JOB_BYTES) {
                    const int64_t r = order[(size_t)e];
                    bytes += off[r + 1] - off[r] + tag_len[r];
                    ++e;
                }
                OutFile* f = &dest_of_code->f[k];
                pieces.push_back(Piece{f, f->next_submit++, k ? r2_text : r1_text, off, order.data() + a, e - a, bytes});
                a = e;
            }
        }
    }
    scatter_timer.reset();
    if (pieces.empty()) return QD_OK;
    // 3. back-pressure, then one job per piece: format (until then `order`, the texts and the tags
    //    are needed: this call waits for that), compress, deliver in order
    int64_t batch_bytes = 0;
    for (const Piece& p : pieces) batch_bytes += p.text_bytes;
    {
        std::unique_lock<std::mutex> g(s->m);
        s->cv.wait(g, [s] { return s->pending_bytes < PENDING_LIMIT; });
        s->pending_bytes += batch_bytes;
        s->pending_jobs += (int64_t)pieces.size();
    }
    res->formatted.n = (int64_t)pieces.size();
    Pool& P = pool();
    for (const Piece& p : pieces) {
        P.submit([s, p, tag_rows, tag_stride, tag_len, res]() mutable {
            // Huffman-only members with a deflate device: format straight into a page-locked buffer and hand the piece
            // to the device lanes (which deliver it); no buffer free = this core codes the piece itself, as below
            if (device_level(s->level) && s->deflate_device >= 0) {
                DeflateService* sv = deflate_service(s->deflate_device);
                const size_t need = (size_t)p.text_bytes + 8 * (size_t)p.n_sel + 16;
                uint8_t* buf = need <= DeflateService::BUF_BYTES ? sv->take_buffer() : nullptr;
                if (!buf) g_stage_ns[ST_NO_BUFFER].fetch_add(1);
                if (buf) {
                    int64_t w;
                    {
                        StageTimer timer(ST_FORMAT);
                        w = qd_format_records(p.text, p.rec_off, p.sel, p.n_sel, tag_rows, tag_stride, tag_len, buf,
                                              (int64_t)DeflateService::BUF_BYTES);
                    }
                    res->formatted.done();
                    res.reset();
                    if (w >= 0) {
                        uint32_t crc;
                        {
                            StageTimer timer(ST_CRC);
                            crc = qd_io_crc32(buf, (size_t)w);
                        }
                        sv->submit(DevPiece{s, p.f, p.seq, p.text_bytes, buf, DeflateService::BUF_BYTES, w, crc, s->level});
                    } else {
                        sv->give_buffer(buf);
                        sink_error(s, "qd_format_records failed (malformed record text)");
                        finish_piece(s, p.f, p.seq, p.text_bytes, w, Bytes(), false);
                    }
                    return;
                }
            }
            thread_local Bytes text;  // formatted records of this piece: the pool thread's scratch, grown once
            Bytes member;
            text.clear();
            text.resize((size_t)p.text_bytes + 8 * (size_t)p.n_sel + 16);
            int64_t w;
            {
                StageTimer timer(ST_FORMAT);
                w = qd_format_records(p.text, p.rec_off, p.sel, p.n_sel, tag_rows, tag_stride, tag_len, text.data(), (int64_t)text.size());
            }
            res->formatted.done();
            res.reset();  // nothing of the batch is touched after this line: the last job to get here frees it
            bool ok = w >= 0;
            if (!ok) sink_error(s, "qd_format_records failed (malformed record text)");
            if (ok) {
                StageTimer timer(ST_DEFLATE);
                if (!gzip_member(text.data(), (size_t)w, s->level, member)) {
                    sink_error(s, "gzip compression failed");
                    ok = false;
                }
            }
            if (!ok) member.clear();  // keep the file's sequence moving
            finish_piece(s, p.f, p.seq, p.text_bytes, w, std::move(member), false);
        });
    }
    if (!owned) res->formatted.wait();  // the caller's buffers are free again
    return QD_OK;
}

int qd_sink_route(qd_sink* s, int64_t n, const uint16_t* codes, const uint8_t* r1_text, const int64_t* r1_off,
                  const uint8_t* r2_text, const int64_t* r2_off, const uint8_t* tag_rows, int32_t tag_stride,
                  const uint8_t* tag_len) {
    return route_impl(s, n, codes, r1_text, r1_off, r2_text, r2_off, tag_rows, tag_stride, tag_len, nullptr, nullptr);
}

int qd_sink_route_batches(qd_sink* s, int64_t n, const uint16_t* codes, const qd_text_batch* r1, const qd_text_batch* r2,
                          const uint8_t* tag_rows, int32_t tag_stride, const uint8_t* tag_len) {
    if (!r1 || !r2 || !r1->handle || !r2->handle) return QD_ERR_INVALID;
    if (n > r1->n_records || n > r2->n_records) {
        qd_text_batch_free(r1->handle);
        qd_text_batch_free(r2->handle);
        return QD_ERR_INVALID;
    }
    return route_impl(s, n, codes, r1->text, r1->rec_off, r2->text, r2->rec_off, tag_rows, tag_stride, tag_len, r1->handle,
                      r2->handle);
}

int qd_sink_flush(qd_sink* s) {
    if (!s) return QD_ERR_INVALID;
    std::unique_lock<std::mutex> g(s->m);
    s->cv.wait(g, [s] { return s->pending_jobs == 0; });
    return s->err.empty() ? QD_OK : QD_ERR_FORMAT;
}

int qd_sink_stats(qd_sink* s, int64_t* members, int64_t* bytes_in, int64_t* bytes_out, int64_t* files) {
    if (!s) return QD_ERR_INVALID;
    std::lock_guard<std::mutex> g(s->m);
    if (members) *members = s->members;
    if (bytes_in) *bytes_in = s->bytes_in;
    if (bytes_out) *bytes_out = s->bytes_out;
    if (files) *files = 2 * (int64_t)s->dest.size();
    return QD_OK;
}

int qd_sink_close(qd_sink* s) {
    if (!s) return QD_OK;
    const int r = qd_sink_flush(s);
    delete s;
    return r;
}

}  // extern "C"

// ---- native chunk reader -----------------------------------------------------------------------------------
// Replaces what the reference draws from pyFastq.FastqReader one record at a time (src/Quade.py:203-214):
// two threads per open file: one reads and inflates it (gzip members: libdeflate when a whole member fits the
// window, streaming zlib otherwise -- concatenated members are legal, the reference's own writer appends
// them, src/FastqWriter.py:83-90), the other scans the records (a record whose sequence and quality lengths differ
// is dropped inside its own stream) and hands over batches of exactly `batch_records` kept
// records (fewer only at the end of the file): one text block + the record offsets.  The consumer gets
// finished batches; nothing of this runs on its thread.
namespace {

// plain allocation, not value-initialised: a vector would write (and so page in) every byte up front --
// hundreds of MB per reader that are mostly never touched
struct RawBuf {
    uint8_t* p = nullptr;
    size_t n = 0;
    ~RawBuf() { free(p); }
    uint8_t* data() { return p; }
    size_t size() const { return n; }
    void resize(size_t m) {  // grows only; contents are not preserved
        if (m <= n) return;
        free(p);
        p = (uint8_t*)malloc(m);
        n = p ? m : 0;
    }
};

// Text buffers of finished batches are kept for the next ones (a few per process): a fresh 180 MB buffer per batch
// means 45 000 page faults and as many pages zeroed by the kernel before the reader has copied a byte into it.
std::mutex g_text_pool_mutex;
std::vector<std::pair<uint8_t*, int64_t>> g_text_pool;  // (buffer, capacity)
constexpr size_t TEXT_POOL_MAX = 12;
constexpr int64_t TEXT_POOL_MAX_BYTES = (int64_t)3 << 30;  // and never more than this much kept idle
int64_t g_text_pool_bytes = 0;

uint8_t* text_alloc(int64_t need, int64_t& cap) {
    {
        std::lock_guard<std::mutex> g(g_text_pool_mutex);
        for (size_t i = 0; i < g_text_pool.size(); ++i)
            if (g_text_pool[i].second >= need && g_text_pool[i].second <= 3 * need) {
                uint8_t* p = g_text_pool[i].first;
                cap = g_text_pool[i].second;
                g_text_pool_bytes -= cap;
                g_text_pool.erase(g_text_pool.begin() + (long)i);
                return p;
            }
    }
    cap = need;
    return (uint8_t*)malloc((size_t)need);
}
void text_free(uint8_t* p, int64_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_text_pool_mutex);
        if (g_text_pool.size() < TEXT_POOL_MAX && g_text_pool_bytes + cap <= TEXT_POOL_MAX_BYTES) {
            g_text_pool.emplace_back(p, cap);
            g_text_pool_bytes += cap;
            return;
        }
    }
    free(p);
}

struct Batch {
    uint8_t* text = nullptr;
    int64_t cap = 0, text_len = 0, n = 0;
    std::vector<int64_t> off;
    ~Batch() { text_free(text, cap); }
};

constexpr size_t READ_BYTES = 8u << 20;     // compressed bytes per read()
constexpr size_t WINDOW = 32u << 20;        // a member inflated in one piece must lie inside this much input ...
constexpr size_t LOW_WATER = 4u << 20;      // input is topped up when fewer unread bytes than this remain
constexpr size_t MEMBER_OUT = 192u << 20;   // ... and inflate to at most this much
constexpr size_t PIECE = 4u << 20;          // text handed to the scanner at a time

}  // namespace

// device inflate of BGZF runs (quade_api.cpp); weak: this file also builds without the HIP half (sanitizer tests)
extern "C" {
int qd_inflater_create(int device_id, qd_inflater** out) __attribute__((weak));
int qd_inflater_run(qd_inflater* inflater, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len,
                    int32_t* bad_block) __attribute__((weak));
int qd_inflater_destroy(qd_inflater* inflater) __attribute__((weak));
int qd_inflater_set_form(qd_inflater* inflater, int32_t form) __attribute__((weak));
int qd_inflater_run_pinned(qd_inflater* inflater, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len,
                           int32_t* bad_block) __attribute__((weak));
void* qd_pinned_alloc(int64_t bytes) __attribute__((weak));
void qd_pinned_free(void* p) __attribute__((weak));
}

namespace {
struct BgzfRun;
}
struct qd_reader {
    int inflate_device = -1;          // >= 0: BGZF runs go through inflaters on that device
    // device lanes: a few threads of this reader's own, each with its inflater, that mostly sleep on the device
    // (the library's pool is for work that needs a core); runs queue here and are collected in file order
    std::vector<std::thread> dev_threads;
    std::mutex dm;
    std::condition_variable dcv;
    std::deque<std::shared_ptr<BgzfRun>> devq;
    bool dev_stop = false, dev_failed = false;
    std::vector<uint8_t*> pin_pool;   // page-locked text buffers of PIN_BYTES each, recycled between device runs
    int64_t pin_made = 0;
    std::atomic<int64_t> device_runs{0}, host_runs{0};
    std::atomic<int64_t> pgz_parallel{0}, pgz_serial{0};  // ordinary gzip: chunks inflated speculatively / by the coordinator
    std::string path, err;
    int fd = -1;
    bool gz = false;
    // raw mode (qdio::raw_open): no record scanner; the consumer takes the inflater's text pieces itself
    bool raw = false;
    int64_t start_offset = 0;  // raw mode: the file is read from here on
    int raw_slot = 0;          // next hand-over slot the consumer looks at
    bool raw_holding = false;  // the consumer still holds the previous slot
    int64_t B = 0;
    size_t depth = 2;
    std::thread th, th_inflate;
    std::mutex m;
    std::condition_variable cv_room, cv_ready;
    // inflater -> batcher hand-off: two scratch blocks
    RawBuf scratch[2];
    std::mutex hm;
    std::condition_variable hcv;
    int hstate[2] = {0, 0};
    size_t hlen[2] = {0, 0};
    const uint8_t* hptr[2] = {nullptr, nullptr};
    std::shared_ptr<void> held[2];
    bool inflated = false;
    std::deque<Batch*> ready;
    bool done = false, stop = false;
    // producer state
    Batch* cur = nullptr;
    int64_t fill = 0, scan = 0;
    std::vector<int64_t> nls;  // newline positions of cur->text found so far, beyond `scan`
    size_t nl_at = 0;          // next unused entry of nls
    int64_t nl_from = 0;       // cur->text has been searched for newlines up to here
    double avg = 400.0;  // bytes per record, learned
    uint8_t last = '\n';
};

namespace {

Batch* new_batch(qd_reader* r, int64_t need) {
    Batch* b = new Batch();
    const int64_t want = std::max<int64_t>((int64_t)((double)r->B * r->avg * 1.08) + (int64_t)PIECE, need + (int64_t)PIECE);
    b->text = text_alloc(want, b->cap);
    b->off.reserve((size_t)std::min<int64_t>(r->B, 1 << 22) + 1);
    return b;
}

// blocks while the queue is full; false when the reader is being closed
bool push_batch(qd_reader* r, Batch* b) {
    std::unique_lock<std::mutex> g(r->m);
    r->cv_room.wait(g, [r] { return r->stop || r->ready.size() < r->depth; });
    if (r->stop) {
        delete b;
        return false;
    }
    r->ready.push_back(b);
    r->cv_ready.notify_all();
    return true;
}

// Newline positions of t[from, to) appended to out: 32 bytes per step with AVX2 (a fastq line is ~80 bytes: one
// memchr call per line spends more time entering and leaving memchr than searching), memchr elsewhere.
__attribute__((target("avx2"))) void newlines_avx2(const uint8_t* t, int64_t from, int64_t to, std::vector<int64_t>& out) {
    const __m256i nl = _mm256_set1_epi8('\n');
    int64_t i = from;
    for (; i + 32 <= to; i += 32) {
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(t + i)), nl));
        while (m) {
            out.push_back(i + __builtin_ctz(m));
            m &= m - 1;
        }
    }
    for (; i < to; ++i)
        if (t[i] == '\n') out.push_back(i);
}
void newlines(const uint8_t* t, int64_t from, int64_t to, std::vector<int64_t>& out) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return newlines_avx2(t, from, to, out);
    for (int64_t i = from; i < to;) {
        const void* p = memchr(t + i, '\n', (size_t)(to - i));
        if (!p) break;
        out.push_back((const uint8_t*)p - t);
        i = ((const uint8_t*)p - t) + 1;
    }
}

void fail_reader(qd_reader* r, const std::string& msg);

bool scan_records(qd_reader* r) {
    for (;;) {
        Batch* b = r->cur;
        // every newline of the text fed so far, then records four lines at a time (same rules as next_record():
        // a record whose sequence and quality lengths differ is dropped, a trailing '\r' is not part of a line)
        if (r->nl_from < r->fill) {
            StageTimer timer(ST_SCAN_LINES);
            newlines(b->text, r->nl_from, r->fill, r->nls);
            r->nl_from = r->fill;
        }
        StageTimer timer(ST_SCAN_RECORDS);
        const uint8_t* t = b->text;
        while (b->n < r->B && r->nl_at + 4 <= r->nls.size()) {
            const int64_t* e = &r->nls[r->nl_at];
            const int64_t head = r->scan, seq = e[0] + 1, qual = e[2] + 1;
            const int64_t seq_end = (e[1] > seq && t[e[1] - 1] == '\r') ? e[1] - 1 : e[1];
            const int64_t qual_end = (e[3] > qual && t[e[3] - 1] == '\r') ? e[3] - 1 : e[3];
            if (seq_end - seq == qual_end - qual) {
                b->off.push_back(head);
                ++b->n;
            }
            r->scan = e[3] + 1;
            r->nl_at += 4;
        }
        if (b->n < r->B) return true;
        b->off.push_back(r->scan);
        b->text_len = r->scan;
        r->avg = (double)r->scan / (double)b->n;
        const int64_t left = r->fill - r->scan;
        Batch* nb = new_batch(r, left);
        if (!nb->text) {
            fail_reader(r, "out of memory for a text batch");
            delete nb;
            return false;
        }
        if (left) memcpy(nb->text, b->text + r->scan, (size_t)left);
        if (!push_batch(r, b)) {
            delete nb;
            r->cur = nullptr;
            return false;
        }
        // the newlines already found in the text that moves to the next batch move with it
        {
            const int64_t base = r->scan;
            size_t k = 0;
            for (size_t i = r->nl_at; i < r->nls.size(); ++i) r->nls[k++] = r->nls[i] - base;
            r->nls.resize(k);
            r->nl_at = 0;
            r->nl_from -= base;
        }
        r->cur = nb;
        r->fill = left;
        r->scan = 0;
    }
}

void fail_reader(qd_reader* r, const std::string& msg);

bool feed(qd_reader* r, const uint8_t* d, size_t len) {
    while (len) {
        const size_t piece = std::min(len, PIECE);
        Batch* b = r->cur;
        if (r->fill + (int64_t)piece > b->cap) {
            const int64_t cap = std::max<int64_t>(b->cap * 2, r->fill + (int64_t)piece);
            uint8_t* grown = (uint8_t*)realloc(b->text, (size_t)cap);
            if (!grown) {
                fail_reader(r, "out of memory growing a text batch");
                return false;
            }
            b->text = grown;
            b->cap = cap;
        }
        {
            StageTimer timer(ST_SCAN_COPY);
            memcpy(b->text + r->fill, d, piece);
        }
        r->fill += (int64_t)piece;
        r->last = d[piece - 1];
        d += piece;
        len -= piece;
        if (!scan_records(r)) return false;
    }
    return true;
}

void fail_reader(qd_reader* r, const std::string& msg) {
    std::lock_guard<std::mutex> g(r->m);
    if (r->err.empty()) r->err = r->path + ": " + msg;
}

// compressed input with a sliding window: [pos, fill) of buf is unread
struct Input {
    int fd;
    RawBuf buf;
    size_t pos = 0, fill = 0;
    bool eof = false;
    explicit Input(int f) : fd(f) { buf.resize(2 * WINDOW); }
    size_t avail() const { return fill - pos; }
    // Makes at least `want` unread bytes available (or everything up to the end of the file): only when
    // fewer are left is the rest moved to the front (a small move) and the buffer filled to its end, so
    // the moves stay a fraction of the bytes consumed.  false on a read error.
    bool refill(size_t want) {
        if (avail() >= want || eof) return true;
        StageTimer timer(ST_READ);
        if (pos) {
            memmove(buf.data(), buf.data() + pos, avail());
            fill -= pos;
            pos = 0;
        }
        while (fill < buf.size() && !eof) {
            const ssize_t g = read(fd, buf.data() + fill, std::min(READ_BYTES, buf.size() - fill));
            if (g < 0) {
                if (errno == EINTR) continue;
                return false;
            }
            if (g == 0) eof = true;
            fill += (size_t)g;
        }
        return true;
    }
};

// ---- the two threads of a reader: inflate -> (two scratch blocks) -> scan + batch ----------------------
// state[i]: 0 = scratch i is the inflater's, 1 = it holds len[i] bytes of text for the batcher.  Both sides
// walk the blocks in the same order 0, 1, 0, 1 ...
bool hand_over(qd_reader* r, int i, size_t len, const uint8_t* ptr = nullptr) {
    std::unique_lock<std::mutex> g(r->hm);
    r->hptr[i] = ptr ? ptr : r->scratch[i].data();
    r->hlen[i] = len;
    r->hstate[i] = 1;
    r->hcv.notify_all();
    const int nx = i ^ 1;
    r->hcv.wait(g, [r, nx] { return r->stop || r->hstate[nx] == 0; });
    return !r->stop;
}

// BGZF (bgzip, htslib): gzip members of <= 64 KiB whose header carries their own size in an extra
// subfield 'B','C' -- so the file can be cut into blocks WITHOUT inflating it, and the blocks inflated
// in parallel.  Returns the block's total size (header .. ISIZE), 0 when [p, p+avail) does not start with
// a complete BGZF block header.
size_t bgzf_block_size(const uint8_t* p, size_t avail) {
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    if (avail < 12 + xlen) return 0;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
        const size_t slen = p[o + 2] | ((size_t)p[o + 3] << 8);
        if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return (size_t)(p[o + 4] | (p[o + 5] << 8)) + 1;
        o += 4 + slen;
    }
    return 0;
}

constexpr size_t PIN_BYTES = 48u << 20;  // text of one device run (16 MB of blocks inflate to ~30 MB of fastq text)

struct BgzfRun {  // consecutive blocks inflated by one pool job
    std::vector<uint8_t> in;  // the compressed blocks (copied out of the window, which moves on)
    Bytes out;
    // device runs: the text lands in a page-locked buffer of the reader's pool instead (no staging copy, no fresh
    // pages to fault in per run); it goes back to the pool with the run
    qd_reader* owner = nullptr;
    uint8_t* pin = nullptr;
    size_t out_len = 0;
    uint8_t* text() { return pin ? pin : out.data(); }
    size_t text_len() const { return pin ? out_len : out.size(); }
    ~BgzfRun();
    std::mutex m;
    std::condition_variable cv;
    bool done = false, ok = true;
};

constexpr size_t BGZF_RUN_BYTES = 2u << 20;  // compressed bytes per job
std::atomic<int64_t> g_bgzf_in_flight{6};  // runs of blocks a reader keeps with the pool (qd_io_set_option "bgzf_in_flight")
std::atomic<int64_t> g_bgzf_device_run_bytes{8 << 20};  // compressed bytes per device launch (qd_io_set_option "bgzf_device_run_bytes"): ~240 blocks,
                                                         // one workgroup each for the inflater's second form (2 launches in flight x 8 MB measured best
                                                         // end to end: profiles/r03_e2e_device_inflate_form2_ab.txt; the first form wanted 3 x 16 MB)
std::atomic<int64_t> g_bgzf_device_lanes{2};         // launches in flight per reader (qd_io_set_option "bgzf_device_lanes")

// the blocks of a run, one after the other, on this thread
void host_inflate_run(BgzfRun& run, LibDeflate& L) {
    StageTimer timer(ST_INFLATE);
    thread_local void* dec = nullptr;
    if (!dec) dec = L.alloc_decompressor();
    size_t ip = 0, op = 0;
    bool good = dec != nullptr;
    while (good && ip < run.in.size()) {
        size_t ain = 0, aout = 0;
        good = L.gzip_decompress_ex(dec, run.in.data() + ip, run.in.size() - ip, run.out.data() + op, run.out.size() - op,
                                    &ain, &aout) == 0 && ain > 0;
        ip += ain;
        op += aout;
    }
    good = good && op == run.out.size();
    std::lock_guard<std::mutex> g(run.m);
    run.ok = good;
    run.done = true;
    run.cv.notify_all();
}

BgzfRun::~BgzfRun() {
    if (pin && owner) {
        std::lock_guard<std::mutex> g(owner->dm);
        owner->pin_pool.push_back(pin);
    }
}

// Inflaters (a stream + grow-only staging buffers each) and page-locked text buffers outlive the readers that
// used them: a run with many chunks opens four readers per chunk, and setting these up per reader cost more
// than the device saved.  Kept per device for the life of the process (a few hundred MB at most).
std::mutex g_dev_cache_mutex;
std::map<int, std::vector<qd_inflater*>> g_idle_inflaters;
std::vector<uint8_t*> g_idle_pins;

qd_inflater* borrow_inflater(int device) {
    {
        std::lock_guard<std::mutex> g(g_dev_cache_mutex);
        auto& v = g_idle_inflaters[device];
        if (!v.empty()) {
            qd_inflater* f = v.back();
            v.pop_back();
            return f;
        }
    }
    qd_inflater* f = nullptr;
    return qd_inflater_create(device, &f) == QD_OK ? f : nullptr;
}
void return_inflater(int device, qd_inflater* f) {
    if (!f) return;
    std::lock_guard<std::mutex> g(g_dev_cache_mutex);
    g_idle_inflaters[device].push_back(f);
}

// a device lane: takes queued runs, inflates them on the GPU (the thread sleeps meanwhile), falls back to the
// host for a run the device refuses; after a HIP error the lane leaves the rest to the host
void device_lane(qd_reader* r) {
    LibDeflate& L = deflate_lib();
    qd_inflater* inf = borrow_inflater(r->inflate_device);
    bool usable = inf != nullptr;
    if (usable && qd_inflater_set_form) {  // (a borrowed inflater may have been made under another setting)
        const char* fe = getenv("QUADE_INFLATE_FORM");
        (void)qd_inflater_set_form(inf, fe && atoi(fe) == 1 ? 1 : 2);
    }
    // test hook: pretend the device fails once the reader has inflated this many runs on it (the fall-back to the
    // host pool in the middle of a file is otherwise unreachable without breaking a GPU)
    const int64_t fail_after = g_test_inflate_fail_after.load();
    for (;;) {
        std::shared_ptr<BgzfRun> run;
        {
            std::unique_lock<std::mutex> g(r->dm);
            r->dcv.wait(g, [r] { return r->dev_stop || !r->devq.empty(); });
            if (r->devq.empty()) break;
            run = r->devq.front();
            r->devq.pop_front();
        }
        bool taken = false;
        if (usable) {
            StageTimer timer(ST_DEV_INFLATE);
            int32_t bad = -1;
            const int rc = (fail_after >= 0 && r->device_runs >= fail_after) ? QD_ERR_HIP : run->pin ? qd_inflater_run_pinned(inf, run->in.data(), (int64_t)run->in.size(), run->pin, (int64_t)run->out_len, &bad)
                                    : qd_inflater_run(inf, run->in.data(), (int64_t)run->in.size(), run->out.data(), (int64_t)run->out.size(), &bad);
            if (rc == QD_OK) {
                std::lock_guard<std::mutex> g(run->m);
                run->ok = run->done = taken = true;
                run->cv.notify_all();
                ++r->device_runs;
            } else if (rc != QD_ERR_FORMAT) {
                usable = false;
                std::lock_guard<std::mutex> g(r->dm);
                r->dev_failed = true;
            }
        }
        if (!taken) {
            if (run->pin) {  // the host inflates into ordinary memory
                run->out.resize(run->out_len);
                {
                    std::lock_guard<std::mutex> g(r->dm);
                    r->pin_pool.push_back(run->pin);
                }
                run->pin = nullptr;
            }
            host_inflate_run(*run, L);  // (a damaged block shows here as run->ok == false)
            ++r->host_runs;
        }
    }
    if (usable) return_inflater(r->inflate_device, inf);
    else if (inf) qd_inflater_destroy(inf);
}

// The whole file as BGZF: the inflater thread only walks the block headers and hands runs of blocks to the
// pool; it collects the runs in file order.  Returns false when the file stops being BGZF where a block is
// expected (the caller then continues with the sequential member loop from in.pos).
bool inflate_bgzf(qd_reader* r, Input& in, int& cur, bool& ok) {
    LibDeflate& L = deflate_lib();
    std::deque<std::shared_ptr<BgzfRun>> flight;
    // r->held[i]: the run whose text the batcher reads out of slot i; it lives in the reader (not on this
    // stack) so that a close() racing with the batcher's copy cannot free the text under it
    bool still_bgzf = true, more = true;
    auto collect_one = [&]() {  // oldest run -> batcher (in file order)
        std::shared_ptr<BgzfRun> run = flight.front();
        flight.pop_front();
        {
            std::unique_lock<std::mutex> g(run->m);
            run->cv.wait(g, [&] { return run->done; });
        }
        if (!run->ok) {
            fail_reader(r, "damaged BGZF block");
            ok = false;
            return;
        }
        if (run->text_len()) {
            r->held[cur] = run;  // keeps the text alive until this slot is handed over again
            ok = hand_over(r, cur, run->text_len(), run->text());
            cur ^= 1;
        }
    };
    // device mode: a run is inflated by the GPU (one lane per block) while this thread waits for it -- the
    // batcher works on the previous run meanwhile; a run the device refuses is inflated here instead
    bool on_device = r->inflate_device >= 0 && qd_inflater_create && qd_inflater_run && qd_inflater_destroy;
    if (on_device && r->dev_threads.empty())
        for (int64_t i = 0, nl = g_bgzf_device_lanes.load(); i < nl; ++i) r->dev_threads.emplace_back(device_lane, r);
    while (ok && more) {
        // one run: whole blocks up to BGZF_RUN_BYTES
        std::shared_ptr<BgzfRun> run = std::make_shared<BgzfRun>();
        size_t out_bytes = 0;
        while (run->in.size() < (on_device ? (size_t)g_bgzf_device_run_bytes.load() : BGZF_RUN_BYTES)) {
            if (!in.refill(1u << 17)) {
                fail_reader(r, strerror(errno));
                ok = false;
                break;
            }
            if (!in.avail()) {
                more = false;
                break;
            }
            const size_t bs = bgzf_block_size(in.buf.data() + in.pos, in.avail());
            if (!bs || bs > in.avail() || bs < 26) {
                still_bgzf = more = false;  // not (or no longer) BGZF here: the member loop takes over
                break;
            }
            const uint8_t* b = in.buf.data() + in.pos;
            const size_t isize = (size_t)b[bs - 4] | ((size_t)b[bs - 3] << 8) | ((size_t)b[bs - 2] << 16) | ((size_t)b[bs - 1] << 24);
            if (isize > 65536) {  // a BGZF block holds at most 64 KiB of text: whatever this is, its trailer does not size a buffer
                still_bgzf = more = false;
                break;
            }
            out_bytes += isize;
            run->in.insert(run->in.end(), b, b + bs);
            in.pos += bs;
        }
        if (!ok) break;
        if (!run->in.empty()) {
            if (on_device && qd_pinned_alloc && qd_inflater_run_pinned && out_bytes <= PIN_BYTES) {
                std::lock_guard<std::mutex> g(r->dm);
                if (!r->pin_pool.empty()) {
                    run->pin = r->pin_pool.back();
                    r->pin_pool.pop_back();
                } else if (r->pin_made < 12) {  // lanes + queue + the two the batcher holds, with slack
                    {
                        std::lock_guard<std::mutex> c(g_dev_cache_mutex);
                        if (!g_idle_pins.empty()) {
                            run->pin = g_idle_pins.back();
                            g_idle_pins.pop_back();
                        }
                    }
                    if (!run->pin) run->pin = (uint8_t*)qd_pinned_alloc((int64_t)PIN_BYTES);
                    if (run->pin) ++r->pin_made;
                }
                run->owner = r;
                run->out_len = out_bytes;
            }
            if (!run->pin) run->out.resize(out_bytes);
            flight.push_back(run);
            if (on_device) {
                std::lock_guard<std::mutex> g(r->dm);
                if (r->dev_failed) {
                    // the lanes have given the device up: this run (and the rest of the file) is the host pool's.  It
                    // may already hold a page-locked buffer of the reader -- hand that back and give the run ordinary
                    // memory, as device_lane does, or the host would inflate into an empty `out` (and a pool job
                    // holding the last reference to the run would touch the reader after qd_reader_close)
                    on_device = false;
                    if (run->pin) {
                        r->pin_pool.push_back(run->pin);
                        run->pin = nullptr;
                    }
                    run->owner = nullptr;
                } else {
                    r->devq.push_back(run);
                }
            }
            if (!on_device && !run->pin && run->out.size() != out_bytes) run->out.resize(out_bytes);
            if (on_device) {
                r->dcv.notify_one();
            } else {
                ++r->host_runs;
                pool().submit([run, &L] { host_inflate_run(*run, L); }, true);
            }
        }
        while (ok && !flight.empty() && (flight.size() >= (on_device ? r->dev_threads.size() + 1 : (size_t)g_bgzf_in_flight.load()) || !more)) collect_one();
    }
    while (!flight.empty()) {  // stopping early (close / error): the jobs still reference their runs; just wait them out
        std::shared_ptr<BgzfRun> run = flight.front();
        flight.pop_front();
        std::unique_lock<std::mutex> g(run->m);
        run->cv.wait(g, [&] { return run->done; });
    }
    // before anything else reuses the two slots, the batcher must have taken the last runs
    if (ok) {
        std::unique_lock<std::mutex> g(r->hm);
        r->hcv.wait(g, [r] { return r->stop || (r->hstate[0] == 0 && r->hstate[1] == 0); });
    }
    return still_bgzf;
}

// ---- ordinary gzip files (one member or a few, of any size): parallel inflate, quade_pgz.cpp -------------------
std::atomic<int64_t> g_pgz_enabled{1};
std::atomic<int64_t> g_pgz_chunk_bytes{4 << 20};
std::atomic<int64_t> g_pgz_min_file_bytes{8 << 20};  // smaller files: one thread is done before a second could help
std::atomic<int64_t> g_pgz_in_flight{0};             // 0 = half the pool's threads (at least 4) per file

qdpgz::Options pgz_options() {
    qdpgz::Options o;
    o.chunk_bytes = (size_t)std::max<int64_t>(g_pgz_chunk_bytes.load(), 64 << 10);
    const int64_t f = g_pgz_in_flight.load();
    o.in_flight = f > 0 ? (int)f : std::max(4, pool().size() / 2);  // (5, 8 and 16 per file read the same end to end: profiles/r03_e2e_16m_single_*)
    return o;
}

// The whole file through the parallel inflater; false when the file cannot be mapped (a pipe ...): the caller's
// sequential loop takes it then.  *ok = false after an error (the reader has been failed).
bool inflate_parallel(qd_reader* r, int& cur, bool& ok) {
    struct stat sb;
    if (fstat(r->fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < g_pgz_min_file_bytes.load()) return false;
    const size_t size = (size_t)sb.st_size;
    if (size == 0) return false;
    void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, r->fd, 0);
    if (map == MAP_FAILED) return false;
    (void)madvise(map, size, MADV_SEQUENTIAL);
    {
        qdpgz::Gunzip gz((const uint8_t*)map, size, pgz_options(), [](std::function<void()> fn) {
            pool().submit([fn = std::move(fn)] {
                StageTimer timer(ST_INFLATE);
                fn();
            }, true);
        });
        while (ok) {
            std::shared_ptr<qdpgz::Text> t;
            const int rc = gz.next(&t);
            if (rc == 0) break;
            if (rc < 0) {
                fail_reader(r, gz.error());
                ok = false;
                break;
            }
            r->held[cur] = t;  // keeps the text alive until this slot is handed over again
            ok = hand_over(r, cur, t->len, t->data);
            cur ^= 1;
        }
        const qdpgz::Stats st = gz.stats();
        r->pgz_parallel = st.parallel;
        r->pgz_serial = st.serial;
    }  // (the inflater waits for its jobs before the mapping goes)
    munmap(map, size);
    return true;
}

void inflate_thread(qd_reader* r) {
    Input in(r->fd);
    int cur = 0;  // scratch block being filled (block 0 starts free)
    bool ok = true;
    if (!r->gz) {
        r->scratch[0].resize(READ_BYTES);
        r->scratch[1].resize(READ_BYTES);
        while (ok) {
            const ssize_t g = read(r->fd, r->scratch[cur].data(), READ_BYTES);
            if (g < 0) {
                if (errno == EINTR) continue;
                fail_reader(r, strerror(errno));
                break;
            }
            if (g == 0) break;
            ok = hand_over(r, cur, (size_t)g);
            cur ^= 1;
        }
    } else {
        LibDeflate& L = deflate_lib();
        if (L.ok && in.refill(1u << 17) && bgzf_block_size(in.buf.data() + in.pos, in.avail())) {
            // bgzip'd input: blocks are indexed by their headers and inflated in parallel; if the file turns
            // into ordinary gzip members further on, the loop below continues from there
            const bool to_the_end = inflate_bgzf(r, in, cur, ok);
            if (to_the_end) {  // (otherwise both slots are free and the alternation simply goes on at `cur`)
                std::lock_guard<std::mutex> g(r->hm);
                r->inflated = true;
                r->hcv.notify_all();
                return;
            }
        }
        if (in.pos == 0 && r->start_offset == 0 && g_pgz_enabled.load() && inflate_parallel(r, cur, ok)) {
            std::lock_guard<std::mutex> g(r->hm);
            r->inflated = true;
            r->hcv.notify_all();
            return;
        }
        void* dec = L.ok ? L.alloc_decompressor() : nullptr;
        bool whole_members = dec != nullptr;  // until a member turns out not to fit the window
        while (ok) {
            if (!in.refill(whole_members ? LOW_WATER : 1)) {
                fail_reader(r, strerror(errno));
                break;
            }
            if (!in.avail()) break;  // clean end: the last member ended where the file ends
            bool all_zero = in.eof;  // zero padding behind the last member is tolerated
            for (size_t i = in.pos; all_zero && i < in.fill; ++i) all_zero = in.buf.data()[i] == 0;
            if (all_zero) break;
            if (whole_members) {
                if (r->scratch[cur].size() < MEMBER_OUT) r->scratch[cur].resize(MEMBER_OUT);
                size_t ain = 0, aout = 0;
                int res = L.gzip_decompress_ex(dec, in.buf.data() + in.pos, in.avail(), r->scratch[cur].data(),
                                               r->scratch[cur].size(), &ain, &aout);
                if (res != 0 && !in.eof && in.avail() < WINDOW) {  // the member may simply reach beyond what was buffered
                    if (!in.refill(WINDOW)) {
                        fail_reader(r, strerror(errno));
                        break;
                    }
                    res = L.gzip_decompress_ex(dec, in.buf.data() + in.pos, in.avail(), r->scratch[cur].data(),
                                               r->scratch[cur].size(), &ain, &aout);
                }
                if (res == 0) {
                    in.pos += ain;
                    if (aout) {
                        ok = hand_over(r, cur, aout);
                        cur ^= 1;
                    }
                    continue;
                }
                // the member does not end inside the window, or inflates beyond the scratch, or is damaged:
                // the streaming inflater takes it from its first byte (and reports real damage)
                whole_members = false;
            }
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 15 + 16) != Z_OK) {
                fail_reader(r, "inflateInit2 failed");
                break;
            }
            int zr = Z_OK;
            while (ok && zr != Z_STREAM_END) {
                if (!in.avail()) {
                    if (!in.refill(1)) {
                        fail_reader(r, strerror(errno));
                        ok = false;
                        break;
                    }
                    if (!in.avail()) {
                        fail_reader(r, "compressed file ended before the end-of-stream marker");
                        ok = false;
                        break;
                    }
                }
                if (r->scratch[cur].size() < PIECE) r->scratch[cur].resize(PIECE);
                zs.next_in = in.buf.data() + in.pos;
                zs.avail_in = (uInt)std::min<size_t>(in.avail(), 1u << 30);
                zs.next_out = r->scratch[cur].data();
                zs.avail_out = (uInt)PIECE;
                const uInt before = zs.avail_in;
                zr = inflate(&zs, Z_NO_FLUSH);
                if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) {
                    fail_reader(r, std::string("not a valid gzip stream (") + (zs.msg ? zs.msg : "zlib error") + ")");
                    ok = false;
                    break;
                }
                in.pos += before - zs.avail_in;
                const size_t got = PIECE - zs.avail_out;
                if (got) {
                    ok = hand_over(r, cur, got);
                    cur ^= 1;
                }
            }
            inflateEnd(&zs);
            if (ok && dec && !in.eof) whole_members = true;  // the next member may be a small one again
        }
        if (dec) L.free_decompressor(dec);
    }
    std::lock_guard<std::mutex> g(r->hm);
    r->inflated = true;
    r->hcv.notify_all();
}

void batch_thread(qd_reader* r) {
    r->cur = new_batch(r, 0);
    bool ok = r->cur->text != nullptr;
    if (!ok) fail_reader(r, "out of memory for a text batch");
    for (int j = 0; ok; j ^= 1) {
        size_t len;
        {
            std::unique_lock<std::mutex> g(r->hm);
            r->hcv.wait(g, [r, j] { return r->stop || r->hstate[j] == 1 || r->inflated; });
            if (r->stop) {
                ok = false;
                break;
            }
            if (r->hstate[j] != 1) break;  // the inflater is done and this block was never filled
            len = r->hlen[j];
        }
        ok = feed(r, r->hptr[j], len);
        std::lock_guard<std::mutex> g(r->hm);
        r->hstate[j] = 0;
        r->hcv.notify_all();
    }
    bool failed;
    {
        std::lock_guard<std::mutex> g(r->m);
        failed = !r->err.empty();
    }
    if (ok && !failed && r->cur) {
        if (r->fill > 0 && r->last != '\n') {  // a last line without newline still ends a record
            const uint8_t nl = '\n';
            ok = feed(r, &nl, 1);
        }
        if (ok && r->cur) {  // the (possibly empty) last batch
            Batch* b = r->cur;
            r->cur = nullptr;
            b->off.push_back(r->scan);
            b->text_len = r->scan;
            if (b->n > 0)
                push_batch(r, b);
            else
                delete b;
        }
    }
    delete r->cur;
    r->cur = nullptr;
    std::lock_guard<std::mutex> g(r->m);
    r->done = true;
    r->cv_ready.notify_all();
}

thread_local std::string g_reader_open_error;

}  // namespace

extern "C" {

const char* qd_reader_last_error(const qd_reader* r) { return r ? r->err.c_str() : g_reader_open_error.c_str(); }

int qd_reader_open(const char* path, int64_t batch_records, int32_t queue_depth, qd_reader** out) {
    return qd_reader_open_on(path, batch_records, queue_depth, -1, out);
}

int qd_reader_open_on(const char* path, int64_t batch_records, int32_t queue_depth, int32_t device_id, qd_reader** out) {
    if (!path || batch_records < 1 || queue_depth < 1 || queue_depth > 64 || !out) return QD_ERR_INVALID;
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        g_reader_open_error = std::string(path) + ": " + strerror(errno);
        return QD_ERR_FORMAT;
    }
    qd_reader* r = new qd_reader();
    r->path = path;
    r->fd = fd;
    const size_t n = strlen(path);
    r->gz = n >= 3 && (path[n - 3] == '.') && (path[n - 2] == 'g' || path[n - 2] == 'G') && (path[n - 1] == 'z' || path[n - 1] == 'Z');
    r->B = batch_records;
    r->depth = (size_t)queue_depth;
    r->inflate_device = device_id;
    r->th_inflate = std::thread(inflate_thread, r);
    r->th = std::thread(batch_thread, r);
    *out = r;
    return QD_OK;
}

int qd_reader_next(qd_reader* r, qd_text_batch* out) {
    if (!r || !out) return QD_ERR_INVALID;
    memset(out, 0, sizeof *out);
    std::unique_lock<std::mutex> g(r->m);
    r->cv_ready.wait(g, [r] { return !r->ready.empty() || r->done; });
    if (!r->ready.empty()) {
        Batch* b = r->ready.front();
        r->ready.pop_front();
        r->cv_room.notify_all();
        out->text = b->text;
        out->text_len = b->text_len;
        out->rec_off = b->off.data();
        out->n_records = b->n;
        out->handle = b;
        return QD_OK;
    }
    return r->err.empty() ? QD_OK : QD_ERR_FORMAT;  // end of the stream: n_records == 0, handle NULL
}

int qd_text_batch_free(void* handle) {
    delete static_cast<Batch*>(handle);
    return QD_OK;
}

int qd_reader_close(qd_reader* r) {
    if (!r) return QD_OK;
    {
        std::lock_guard<std::mutex> g(r->m);
        std::lock_guard<std::mutex> h(r->hm);
        r->stop = true;
        r->cv_room.notify_all();
        r->hcv.notify_all();
    }
    if (r->th_inflate.joinable()) r->th_inflate.join();
    if (r->th.joinable()) r->th.join();
    for (Batch* b : r->ready) delete b;
    {
        std::lock_guard<std::mutex> g(r->dm);
        r->dev_stop = true;
    }
    r->dcv.notify_all();
    for (auto& t : r->dev_threads) t.join();
    r->held[0].reset();  // runs give their page-locked buffers back to the pool: before the pool goes
    r->held[1].reset();
    {
        std::lock_guard<std::mutex> c(g_dev_cache_mutex);
        for (uint8_t* p : r->pin_pool) {
            if (g_idle_pins.size() < 32) g_idle_pins.push_back(p);
            else if (qd_pinned_free) qd_pinned_free(p);
        }
    }
    r->pin_pool.clear();
    close(r->fd);
    delete r;
    return QD_OK;
}

int qd_io_set_option(const char* name, int64_t value) {
    if (!name) return QD_ERR_INVALID;
    const std::string n(name);
    if (n == "parallel_gunzip") g_pgz_enabled = value != 0;
    else if (n == "gunzip_chunk_bytes" && value >= (64 << 10)) g_pgz_chunk_bytes = value;
    else if (n == "gunzip_min_file_bytes" && value >= 0) g_pgz_min_file_bytes = value;
    else if (n == "gunzip_in_flight" && value >= 0 && value <= 256) g_pgz_in_flight = value;
    else if (n == "test_deflate_fail_after") g_test_deflate_fail_after = value;
    else if (n == "test_inflate_fail_after") g_test_inflate_fail_after = value;
    else if (n == "bgzf_in_flight" && value >= 1 && value <= 256) g_bgzf_in_flight = value;
    else if (n == "bgzf_device_lanes" && value >= 1 && value <= 16) g_bgzf_device_lanes = value;
    else if (n == "bgzf_device_run_bytes" && value >= (1 << 20) && value <= (16 << 20)) g_bgzf_device_run_bytes = value;
    else return QD_ERR_INVALID;
    return QD_OK;
}

int qd_reader_gunzip_stats(const qd_reader* r, int64_t* parallel_chunks, int64_t* serial_chunks) {
    if (!r) return QD_ERR_INVALID;
    if (parallel_chunks) *parallel_chunks = r->pgz_parallel;
    if (serial_chunks) *serial_chunks = r->pgz_serial;
    return QD_OK;
}

static thread_local std::string g_gunzip_error;
const char* qd_gunzip_last_error(void) { return g_gunzip_error.c_str(); }

int qd_gunzip_buffer(const uint8_t* comp, int64_t comp_len, int64_t chunk_bytes, uint8_t* out, int64_t out_cap, int64_t* out_len,
                     int64_t* stats) {
    if ((!comp && comp_len) || comp_len < 0 || (!out && out_cap) || out_cap < 0 || !out_len) return QD_ERR_INVALID;
    qdpgz::Options o = pgz_options();
    if (chunk_bytes > 0) o.chunk_bytes = (size_t)chunk_bytes;
    qdpgz::Gunzip gz(comp, (size_t)comp_len, o, [](std::function<void()> fn) { pool().submit(std::move(fn), true); });
    int64_t at = 0;
    int rc;
    for (;;) {
        std::shared_ptr<qdpgz::Text> t;
        rc = gz.next(&t);
        if (rc != 1) break;
        if (at + (int64_t)t->len > out_cap) {
            g_gunzip_error = "output buffer too small";
            *out_len = at;
            return QD_ERR_INVALID;
        }
        memcpy(out + at, t->data, t->len);
        at += (int64_t)t->len;
    }
    *out_len = at;
    if (stats) {
        const qdpgz::Stats st = gz.stats();
        stats[0] = st.chunks;
        stats[1] = st.parallel;
        stats[2] = st.serial;
        stats[3] = st.members;
        stats[4] = st.search_bits;
    }
    if (rc < 0) {
        g_gunzip_error = gz.error();
        return QD_ERR_FORMAT;
    }
    return QD_OK;
}

int qd_reader_inflate_stats(const qd_reader* r, int64_t* device_runs, int64_t* host_runs) {
    if (!r) return QD_ERR_INVALID;
    if (device_runs) *device_runs = r->device_runs;
    if (host_runs) *host_runs = r->host_runs;
    return QD_OK;
}

}  // extern "C"

// ---- seams for the device chunk pipeline (quade_io_internal.h) -----------------------------------------------------
#include "quade_io_internal.h"
namespace qdio {

qd_reader* raw_open(const char* path, int64_t start_offset, std::string* err) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0 || (start_offset > 0 && lseek(fd, (off_t)start_offset, SEEK_SET) < 0)) {
        if (err) *err = std::string(path) + ": " + strerror(errno);
        if (fd >= 0) close(fd);
        return nullptr;
    }
    qd_reader* r = new qd_reader();
    r->path = path;
    r->fd = fd;
    const size_t n = strlen(path);
    r->gz = n >= 3 && (path[n - 3] == '.') && (path[n - 2] == 'g' || path[n - 2] == 'G') && (path[n - 1] == 'z' || path[n - 1] == 'Z');
    r->raw = true;
    r->start_offset = start_offset;
    r->B = 1;
    r->th_inflate = std::thread(inflate_thread, r);
    return r;
}

int raw_next(qd_reader* r, const uint8_t** ptr, size_t* len, std::string* err) {
    std::unique_lock<std::mutex> g(r->hm);
    if (r->raw_holding) {  // the piece handed out by the previous call goes back to the inflater
        r->hstate[r->raw_slot] = 0;
        r->held[r->raw_slot].reset();
        r->raw_slot ^= 1;
        r->raw_holding = false;
        r->hcv.notify_all();
    }
    const int j = r->raw_slot;
    r->hcv.wait(g, [r, j] { return r->stop || r->hstate[j] == 1 || r->inflated; });
    if (r->hstate[j] == 1) {
        *ptr = r->hptr[j];
        *len = r->hlen[j];
        r->raw_holding = true;
        return 1;
    }
    g.unlock();
    std::lock_guard<std::mutex> e(r->m);
    if (!r->err.empty()) {
        if (err) *err = r->err;
        return -1;
    }
    return 0;
}

void raw_close(qd_reader* r) { qd_reader_close(r); }

size_t bgzf_block_size(const uint8_t* p, size_t avail) { return ::bgzf_block_size(p, avail); }

bool host_inflate_members(const uint8_t* comp, size_t comp_len, uint8_t* out, size_t out_len) {
    LibDeflate& L = deflate_lib();
    size_t ip = 0, op = 0;
    if (L.ok) {
        void* dec = L.alloc_decompressor();
        bool good = dec != nullptr;
        while (good && ip < comp_len) {
            size_t ain = 0, aout = 0;
            good = L.gzip_decompress_ex(dec, comp + ip, comp_len - ip, out + op, out_len - op, &ain, &aout) == 0 && ain > 0;
            ip += ain;
            op += aout;
        }
        if (dec) L.free_decompressor(dec);
        return good && op == out_len;
    }
    while (ip < comp_len) {  // zlib, member by member
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
        zs.next_in = const_cast<Bytef*>(comp + ip);
        zs.avail_in = (uInt)std::min<size_t>(comp_len - ip, 1u << 30);
        zs.next_out = out + op;
        zs.avail_out = (uInt)std::min<size_t>(out_len - op, 1u << 30);
        const uInt in0 = zs.avail_in, out0 = zs.avail_out;
        const int zr = inflate(&zs, Z_FINISH);
        ip += in0 - zs.avail_in;
        op += out0 - zs.avail_out;
        inflateEnd(&zs);
        if (zr != Z_STREAM_END) return false;
    }
    return op == out_len;
}

bool host_gzip_member(const uint8_t* text, size_t n, int level, std::vector<uint8_t>* out) {
    Bytes m;
    if (!gzip_member(text, n, level, m)) return false;
    out->assign(m.data(), m.data() + m.size());
    return true;
}

uint32_t crc32(const uint8_t* p, size_t n) { return qd_io_crc32(p, n); }

SinkInfo sink_info(const qd_sink* s) {
    SinkInfo i;
    i.level = s->level;
    i.write_pass = s->write_pass;
    i.write_fail = s->write_fail;
    i.write_undet = s->write_undet;
    i.n_samples = (uint32_t)s->names.size();
    return i;
}

void* sink_file(qd_sink* s, uint32_t code, int k) {
    Dest* d = sink_dest(s, code);
    return d ? &d->f[k] : nullptr;
}

bool sink_append(qd_sink* s, void* file, const uint8_t* data, size_t n) {
    OutFile* f = static_cast<OutFile*>(file);
    std::string why;
    std::lock_guard<std::mutex> g(f->m);
    if (append_file(f->path, data, n, why)) return true;
    sink_error(s, why);
    return false;
}

void sink_account(qd_sink* s, int64_t members, int64_t device_members, int64_t text_bytes, int64_t gzip_bytes) {
    std::lock_guard<std::mutex> g(s->m);
    s->members += members;
    s->device_members += device_members;
    s->bytes_in += text_bytes;
    s->bytes_out += gzip_bytes;
}

void sink_fail(qd_sink* s, const std::string& msg) { sink_error(s, msg); }

void pool_submit(std::function<void()> fn, bool urgent) { pool().submit(std::move(fn), urgent); }
int pool_size() { return pool().size(); }

}  // namespace qdio

// ---- whole-buffer gzip writer (tooling: synthetic inputs; any binding that holds text in memory) ------------
namespace {

thread_local std::map<int, void*> g_raw_comp;  // libdeflate compressors of this thread, by level

// BGZF blocks (bgzip / htslib layout) of text[0..n): 0xFF00 text bytes per block, raw deflate, 'BC' size field
bool bgzf_blocks(const uint8_t* text, size_t n, int level, Bytes& out) {
    LibDeflate& L = deflate_lib();
    const size_t BLOCK = 0xFF00;
    void* c = nullptr;
    if (L.ok && L.deflate_compress && L.crc32) {
        void*& slot = g_raw_comp[level];
        if (!slot) slot = L.alloc_compressor(level);
        c = slot;
    }
    size_t o = 0;
    out.resize((n / BLOCK + 1) * (BLOCK + 1024));
    for (size_t a = 0; a < n; a += BLOCK) {
        const size_t m = std::min(BLOCK, n - a);
        uint8_t* h = out.data() + o;
        static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(h, head, 16);
        size_t body = 0;
        uint32_t crc;
        if (c) {
            body = L.deflate_compress(c, text + a, m, h + 18, BLOCK + 1024 - 26);
            crc = L.crc32(0, text + a, m);
        } else {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
            zs.next_in = const_cast<Bytef*>(text + a);
            zs.avail_in = (uInt)m;
            zs.next_out = h + 18;
            zs.avail_out = (uInt)(BLOCK + 1024 - 26);
            const int zr = deflate(&zs, Z_FINISH);
            body = zr == Z_STREAM_END ? (BLOCK + 1024 - 26) - zs.avail_out : 0;
            deflateEnd(&zs);
            crc = (uint32_t)::crc32(0L, text + a, (uInt)m);
        }
        if (!body || body + 25 > 0xFFFF) return false;
        const size_t bsize = body + 25;  // total block size - 1
        h[16] = (uint8_t)bsize;
        h[17] = (uint8_t)(bsize >> 8);
        uint8_t* t = h + 18 + body;
        for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i));
        for (int i = 0; i < 4; ++i) t[4 + i] = (uint8_t)((uint32_t)m >> (8 * i));
        o += 18 + body + 8;
    }
    out.resize(o);
    return true;
}

}  // namespace

extern "C" int qd_write_gzip_file(const char* path, const uint8_t* data, int64_t n, int32_t level, int64_t member_bytes) {
    if (!path || (!data && n) || n < 0 || level < -1 || level > 9 || member_bytes < -1) return QD_ERR_INVALID;
    const bool bgzf = member_bytes == -1;
    const int64_t piece = bgzf ? 64 * 0xFF00 : (member_bytes > 0 ? member_bytes : std::max<int64_t>(n, 1));
    const int64_t np = n ? (n + piece - 1) / piece : 0;
    std::vector<Bytes> parts((size_t)np);
    std::vector<char> good((size_t)np, 0);
    std::shared_ptr<Latch> latch = std::make_shared<Latch>();
    latch->n = np;
    for (int64_t i = 0; i < np; ++i) {
        pool().submit([=, &parts, &good] {
            const uint8_t* p = data + i * piece;
            const size_t m = (size_t)std::min<int64_t>(piece, n - i * piece);
            good[(size_t)i] = bgzf ? bgzf_blocks(p, m, level, parts[(size_t)i]) : gzip_member(p, m, level, parts[(size_t)i]);
            latch->done();
        });
    }
    if (np) latch->wait();
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return QD_ERR_FORMAT;
    bool ok = true;
    auto put = [&](const uint8_t* p, size_t m) {
        while (ok && m) {
            const ssize_t w = write(fd, p, m);
            if (w < 0) {
                if (errno == EINTR) continue;
                ok = false;
                break;
            }
            p += w;
            m -= (size_t)w;
        }
    };
    for (int64_t i = 0; i < np && ok; ++i) {
        ok = good[(size_t)i] != 0;
        if (ok) put(parts[(size_t)i].data(), parts[(size_t)i].size());
    }
    if (np == 0 && !bgzf) {  // an empty file is still one (empty) gzip member
        Bytes e;
        ok = gzip_member(data, 0, level, e);
        if (ok) put(e.data(), e.size());
    }
    if (bgzf) {
        static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        put(eof_block, sizeof eof_block);
    }
    if (close(fd) != 0) ok = false;
    return ok ? QD_OK : QD_ERR_FORMAT;
}

// zlib's combination of two CRC-32s (the third inflater's host side: a gzip member's CRC from its units')
uint32_t qd_crc32_combine_host(uint32_t crc1, uint32_t crc2, uint64_t len2) { return (uint32_t)crc32_combine((uLong)crc1, (uLong)crc2, (z_off_t)len2); }
