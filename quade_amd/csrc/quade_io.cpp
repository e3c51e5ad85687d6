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
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

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
#undef QD_SYM
        ok = alloc_compressor && gzip_compress && gzip_compress_bound && free_compressor && alloc_decompressor &&
             gzip_decompress_ex && free_decompressor;
    }
};
LibDeflate& deflate_lib() {
    static LibDeflate L;
    return L;
}

// one gzip member of `n` bytes at `level` (0..9) -> out; false on failure
bool gzip_member(const uint8_t* in, size_t n, int level, std::vector<uint8_t>& out) {
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
    void submit(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> g(m_);
            q_.push_back(std::move(fn));
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
                cv_.wait(g, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // stop_ and drained
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

std::mutex g_pool_mutex;
std::unique_ptr<Pool> g_pool;
int g_pool_threads = 0;  // 0 = one per hardware thread

Pool& pool() {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    if (!g_pool) {
        int n = g_pool_threads > 0 ? g_pool_threads : (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        g_pool.reset(new Pool(n));
    }
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
    std::map<uint64_t, std::vector<uint8_t>> done;  // finished members waiting for their turn
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
    int quiet = 0;
};

namespace {

constexpr int64_t JOB_BYTES = 2 << 20;            // text per gzip member
constexpr int64_t PENDING_LIMIT = (int64_t)1 << 30;  // formatted + compressed bytes allowed in flight per sink

void sink_error(qd_sink* s, const std::string& msg) {
    std::lock_guard<std::mutex> g(s->m);
    if (s->err.empty()) s->err = msg;
}

bool append_file(const std::string& path, const std::vector<uint8_t>& data, std::string& why) {
    const int fd = open(path.c_str(), O_WRONLY | O_APPEND | O_CREAT | O_CLOEXEC, 0644);
    if (fd < 0) {
        why = path + ": " + strerror(errno);
        return false;
    }
    size_t off = 0;
    while (off < data.size()) {
        const ssize_t w = write(fd, data.data() + off, data.size() - off);
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

// a finished member takes its place in the file's queue; everything that is next in line is written
void deliver(qd_sink* s, OutFile* f, uint64_t seq, std::vector<uint8_t>&& member) {
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

extern "C" {

int qd_io_threads(int32_t n_threads) {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    if (n_threads >= 0 && !g_pool) g_pool_threads = n_threads;
    if (g_pool) return g_pool->size();
    if (g_pool_threads > 0) return g_pool_threads;
    const int hw = (int)std::thread::hardware_concurrency();
    return hw < 1 ? 1 : hw;
}

int qd_io_backend(void) { return deflate_lib().ok ? 1 : 0; }

int qd_sink_create(const char* outdir, int32_t n_samples, const char* const* names, int32_t gzip_level,
                   int32_t write_pass, int32_t write_fail, int32_t write_undetermined, qd_sink** out) {
    if (!outdir || n_samples < 0 || (n_samples > 0 && !names) || gzip_level < 0 || gzip_level > 9 || !out)
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

int qd_sink_set_quiet(qd_sink* s, int32_t quiet) {
    if (!s) return QD_ERR_INVALID;
    s->quiet = quiet;
    return QD_OK;
}

int qd_sink_route(qd_sink* s, int64_t n, const uint16_t* codes, const uint8_t* r1_text, const int64_t* r1_off,
                  const uint8_t* r2_text, const int64_t* r2_off, const uint8_t* tag_rows, int32_t tag_stride,
                  const uint8_t* tag_len) {
    if (!s || n < 0) return QD_ERR_INVALID;
    if (n == 0) return QD_OK;
    if (!codes || !r1_text || !r1_off || !r2_text || !r2_off || !tag_rows || !tag_len) return QD_ERR_INVALID;
    {
        std::lock_guard<std::mutex> g(s->m);
        if (!s->err.empty()) return QD_ERR_FORMAT;
    }
    // 1. counting scatter by routing code (src/Sample.py:74-91 decides per pair; here per batch), stable
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
    std::vector<int64_t> order((size_t)n), fill(start.begin(), start.end() - 1);
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
        auto it = s->dest.find(code);
        if (it == s->dest.end()) {  // first routed pair of this destination: create (truncate) its two files
            std::unique_ptr<Dest> d(new Dest());
            const std::string base = s->outdir + "/" + (undet ? std::string("Undetermined") : s->names[b >> 1] + ((b & 1) ? "_fail" : "_pass"));
            d->f[0].path = base + "_R1.fastq.gz";
            d->f[1].path = base + "_R2.fastq.gz";
            for (int k = 0; k < 2; ++k) {
                if (!s->quiet) {
                    printf("\tCreate %s file\n", d->f[k].path.c_str());
                }
                const int fd = open(d->f[k].path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
                if (fd < 0) {
                    sink_error(s, d->f[k].path + ": " + strerror(errno));
                    return QD_ERR_FORMAT;
                }
                close(fd);
            }
            if (!s->quiet) fflush(stdout);
            it = s->dest.emplace(code, std::move(d)).first;
        }
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
                OutFile* f = &it->second->f[k];
                pieces.push_back(Piece{f, f->next_submit++, k ? r2_text : r1_text, off, order.data() + a, e - a, bytes});
                a = e;
            }
        }
    }
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
    Latch formatted;
    formatted.n = (int64_t)pieces.size();
    Pool& P = pool();
    for (const Piece& p : pieces) {
        P.submit([s, p, tag_rows, tag_stride, tag_len, &formatted] {
            std::vector<uint8_t> text((size_t)p.text_bytes + 8 * (size_t)p.n_sel + 16), member;
            const int64_t w = qd_format_records(p.text, p.rec_off, p.sel, p.n_sel, tag_rows, tag_stride, tag_len, text.data(),
                                                (int64_t)text.size());
            formatted.done();  // nothing of the caller's is touched after this line
            bool ok = w >= 0;
            if (!ok) sink_error(s, "qd_format_records failed (malformed record text)");
            if (ok && !gzip_member(text.data(), (size_t)w, s->level, member)) {
                sink_error(s, "gzip compression failed");
                ok = false;
            }
            if (!ok) member.clear();  // keep the file's sequence moving
            const int64_t out_bytes = (int64_t)member.size();
            deliver(s, p.f, p.seq, std::move(member));
            std::lock_guard<std::mutex> g(s->m);
            s->pending_bytes -= p.text_bytes;
            --s->pending_jobs;
            ++s->members;
            s->bytes_in += w > 0 ? w : 0;
            s->bytes_out += out_bytes;
            s->cv.notify_all();
        });
    }
    formatted.wait();
    return QD_OK;
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
