// Device-resident chunk pipeline of libquade_hip.so (include/quade_hip.h, qd_pipe_*).
//
// Replaces the reference's per-pair loop body and everything it calls -- src/Quade.py:203-221 (four FastqReader.next(),
// slice + fuse, Sample.FINDER) and src/FastqWriter.py:61-90 (name tag, record format, gzip append) -- for whole chunks, with
// the fastq TEXT staying in HBM from the inflater to the coder:
//
//   files --read()--> pinned --H2D--> [BGZF blocks]  inflate_bgzf_blocks2 + CRC-32 check          (quade_inflate.hip, quade_text.hip)
//                                     [other input]  text as the host's readers inflate it --H2D-->
//     -> window of text per stream -> line / record scan (skip-malformed inside the stream)
//     -> lock step: pair j = kept record j of every stream, the chunk ends with its first exhausted stream (Quade.py:210-224)
//     -> index rows -> demux_fast (codes, counters)                                                (quade_kernels.hip)
//     -> stable sort by destination -> output offsets -> format "@name:IDX[:MOL]\nseq\n+\nqual\n"  (quade_text.hip)
//     -> CRC-32 -> lz_subblocks / huff_pieces -> members packed                                    (quade_deflate.hip)
//     --D2H--> pinned --write()--> <dest>_R1/_R2.fastq.gz, in input order
//
// Only compressed bytes cross PCIe (text does for inputs the device cannot inflate: ordinary gzip members, plain files).
// Threads: one feeder per input stream (file -> pinned -> device ring, running ahead into the next chunks), the driver (the
// caller's thread: launches, three small read-backs per batch), a collector (members -> files on the library's pool).
// One compute stream: the device runs the stages of consecutive batches back to back; uploads and downloads have their own.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/quade_hip.h"
#include "quade_deflate.h"
#include "quade_inflate.h"
#include "quade_inflate3.h"
#include "quade_io_internal.h"
#include "quade_pool.h"
#include "quade_text.h"

uint32_t qd_crc32_combine_host(uint32_t crc1, uint32_t crc2, uint64_t len2);  // quade_io.cpp (zlib's)

namespace {

constexpr size_t SEG_BYTES = 16u << 20;  // bytes per upload (compressed blocks or text)
constexpr size_t SEG_TEXT_MAX = 128u << 20;  // text one BGZF upload inflates to at most (a buffer of blocks that hold more goes out in several)
constexpr int RING_SLOTS = 64;           // device ring per stream: how far a feeder runs ahead of the kernels (a 2 M-pair batch of 2 x 150 bp takes ~24 uploads)
constexpr int PIN_SLOTS = 3;             // page-locked upload buffers per stream
constexpr uint32_t PIECE_BYTES = 1u << 20;       // text per gzip member at most (piece_bytes_for)
constexpr size_t WINDOW_MAX = (size_t)1 << 30;   // text per stream and batch (offsets are 32 bit)
constexpr uint32_t LAUNCH_BLOCKS = 8192;         // BGZF blocks per inflate launch at most (sizes the match scratch)
constexpr size_t GROUP_SEGMENTS = 8;             // uploads gathered per inflate launch
constexpr size_t READ_PART = 4u << 20;           // an upload buffer is filled by parallel reads of this much
#ifndef QD_PIPE_INFLATE_STREAMS
#define QD_PIPE_INFLATE_STREAMS 1 /* 2: the inflate launches alternate between two streams of their own (measured: slower, see qd_pipe::is) */
#endif

hipError_t wait_event_napping(hipEvent_t ev) {  // hipEventSynchronize spins on this runtime: poll, then nap
    for (int i = 0; i < 64; ++i) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
    }
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        usleep(50);
    }
}

struct Tick {  // adds the wall time of its scope to a counter
    double& acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Tick(double& a) : acc(a) {}
    ~Tick() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
thread_local double g_alloc_seconds = 0;  // (device allocations are made by the driving thread)

struct DevBuf {  // grow-only device allocation
    uint8_t* p = nullptr;
    size_t cap = 0;
    // keep: the first `keep` bytes survive the growth (copied on `st`, which is then drained)
    hipError_t need(size_t n, size_t keep = 0, hipStream_t st = nullptr) {
        if (n <= cap) return hipSuccess;
        Tick tick(g_alloc_seconds);
        const size_t want = n + n / 4 + (1u << 16);
        uint8_t* q = nullptr;
        size_t got = want;
        hipError_t e = qd_pool_get(want, (void**)&q, &got);  // (quade_pool.h: from what an earlier pipeline of this process released)
        if (e != hipSuccess) return e;
        // nothing queued anywhere may still use the old allocation (the inflate launches run on streams of their own)
        if (p) (void)hipDeviceSynchronize();
        if (p && keep) {
            e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
        }
        if (p) qd_pool_put(p, cap);
        p = q;
        cap = got;
        return hipSuccess;
    }
    void release() {
        if (p) qd_pool_put(p, cap);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinBuf {  // grow-only page-locked host allocation (quade_pool.h: from what an earlier pipeline of this process released)
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) qd_pool_put_pinned(p, cap);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 4096;
        return qd_pool_get_pinned(want, (void**)&p, &cap);
    }
    void release() {
        if (p) qd_pool_put_pinned(p, cap);
        p = nullptr;
        cap = 0;
    }
};

// Small tables on their way to the device: page-locked chunks that are reused once the copy that read them has run.
class StagePool {
  public:
    ~StagePool() {
        for (Chunk& c : chunks_) {
            if (c.ev) (void)hipEventDestroy(c.ev);
            if (c.p) (void)hipHostFree(c.p);
        }
    }
    // copies n bytes of src to dst (device) on st through a page-locked chunk
    hipError_t upload(void* dst, const void* src, size_t n, hipStream_t st) {
        if (!n) return hipSuccess;
        Chunk* c = nullptr;
        for (Chunk& k : chunks_)
            if (k.cap >= n && (!k.used || hipEventQuery(k.ev) == hipSuccess)) {
                c = &k;
                break;
            }
        if (!c) {
            Chunk k;
            k.cap = std::max<size_t>(n + n / 2, 256u << 10);
            hipError_t e = hipHostMalloc((void**)&k.p, k.cap, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&k.ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
            chunks_.push_back(k);
            c = &chunks_.back();
        }
        memcpy(c->p, src, n);
        hipError_t e = hipMemcpyAsync(dst, c->p, n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        c->used = true;
        return hipEventRecord(c->ev, st);
    }

  private:
    struct Chunk {
        uint8_t* p = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool used = false;
    };
    std::deque<Chunk> chunks_;
};

// ---- feeders: file -> page-locked buffer -> device ring ----------------------------------------------------------------------
enum { SEG_BGZF, SEG_TEXT, SEG_END, SEG_ERROR, SEG_GZIP };
struct Segment {
    int kind = SEG_END;
    int chunk = 0;
    int slot = -1;          // ring slot that holds `bytes` bytes
    size_t bytes = 0;
    size_t text_bytes = 0;  // BGZF: what the blocks inflate to
    std::vector<qd_inflate_block> blocks;  // in_off inside the slot, out_off inside the segment's text
    std::vector<uint32_t> crcs;
    std::vector<uint32_t> starts;  // where every block starts in the slot (its file offset = file_off + that)
    uint32_t longest = 0;
    int64_t file_off = 0;   // BGZF: where the blocks start in the file (a run the device refuses is read again by the host)
    std::string err;
};

struct FileSpec {
    std::string path;
    int64_t start = 0;  // > 0: BGZF blocks from this file offset on (a chunk that several ranks share)
    int64_t end = -1;   // >= 0: up to this offset (a block boundary)
};

class Feeder {
  public:
    Feeder(int device, std::vector<FileSpec> files, bool device_gunzip = false) : device_(device), files_(std::move(files)), device_gunzip_(device_gunzip) {}
    ~Feeder() { stop(); }
    hipError_t start() {
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return e;
        if ((e = hipStreamCreateWithFlags(&up_, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = qd_pool_get((size_t)RING_SLOTS * SEG_BYTES + 4096, (void**)&ring_, &ring_cap_)) != hipSuccess) return e;  // (the third inflater's lanes read up to 512 bytes ahead of a payload)
        for (int i = 0; i < RING_SLOTS; ++i) {
            if ((e = hipEventCreateWithFlags(&ready_[i], hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&consumed_[i], hipEventDisableTiming)) != hipSuccess) return e;
        }
        th_ = std::thread([this] { run(); });
        return hipSuccess;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        if (up_) {
            (void)hipStreamSynchronize(up_);
            (void)hipStreamDestroy(up_);
            up_ = nullptr;
        }
        for (int i = 0; i < RING_SLOTS; ++i) {
            if (ready_[i]) (void)hipEventDestroy(ready_[i]);
            if (consumed_[i]) (void)hipEventDestroy(consumed_[i]);
            ready_[i] = consumed_[i] = nullptr;
        }
        for (int i = 0; i < PIN_SLOTS; ++i) {
            if (pin_[i]) qd_pool_put_pinned(pin_[i], pin_cap_[i]);
            pin_[i] = nullptr;
        }
        if (ring_) qd_pool_put(ring_, ring_cap_);
        ring_ = nullptr;
    }
    // the next segment of the stream (blocks); SEG_END closes a chunk
    Segment pop() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !q_.empty(); });
        Segment s = std::move(q_.front());
        q_.pop_front();
        return s;
    }
    const uint8_t* ring() const { return ring_; }
    hipEvent_t ready(int slot) const { return ready_[slot]; }
    // the driver has queued the last reader of `slot` on st
    hipError_t consumed(int slot, hipStream_t st) {
        const hipError_t e = hipEventRecord(consumed_[slot], st);
        {
            std::lock_guard<std::mutex> g(m_);
            state_[slot] = e == hipSuccess ? 2 : 0;
        }
        cv_.notify_all();
        return e;
    }
    // chunks below `c` need no more input (the chunk ended with another stream)
    void skip_below(int c) {
        skip_.store(c);
        cv_.notify_all();
    }

  private:
    void push(Segment&& s) {
        {
            std::lock_guard<std::mutex> g(m_);
            q_.push_back(std::move(s));
        }
        cv_.notify_all();
    }
    void fail(int chunk, const std::string& msg) {
        Segment s;
        s.kind = SEG_ERROR;
        s.chunk = chunk;
        s.err = msg;
        push(std::move(s));
    }
    bool stopping() {
        std::lock_guard<std::mutex> g(m_);
        return stop_;
    }
    // a page-locked buffer whose last upload has run; nullptr when stopping or out of memory
    uint8_t* take_pin() {
        const int j = pin_next_;
        pin_next_ = (pin_next_ + 1) % PIN_SLOTS;
        if (!pin_[j] && qd_pool_get_pinned(SEG_BYTES + 65536, (void**)&pin_[j], &pin_cap_[j]) != hipSuccess) return nullptr;
        if (pin_slot_[j] >= 0 && wait_event_napping(ready_[pin_slot_[j]]) != hipSuccess) return nullptr;
        pin_slot_[j] = -1;
        pin_cur_ = j;
        return pin_[j];
    }
    // a free ring slot (-1: stopping)
    int take_slot() {
        const int s = slot_next_;
        slot_next_ = (slot_next_ + 1) % RING_SLOTS;
        int state;
        {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return stop_ || state_[s] != 1; });
            if (stop_) return -1;
            state = state_[s];
        }
        if (state == 2 && wait_event_napping(consumed_[s]) != hipSuccess) return -1;
        return s;
    }
    // pin[0 .. n) -> a ring slot; the segment goes to the driver
    bool upload(Segment& seg, const uint8_t* pin, size_t n) {
        const int s = take_slot();
        if (s < 0) {  // stopping -- or the wait for the slot's last reader failed: the driver must not see a clean end of stream
            if (!stopping()) fail(seg.chunk, "upload failed: waiting for a ring slot");
            return false;
        }
        hipError_t e = hipMemcpyAsync(ring_ + (size_t)s * SEG_BYTES, pin, n, hipMemcpyHostToDevice, up_);
        if (e == hipSuccess) e = hipEventRecord(ready_[s], up_);
        if (e != hipSuccess) {
            fail(seg.chunk, std::string("upload failed: ") + hipGetErrorString(e));
            return false;
        }
        pin_slot_[pin_cur_] = s;
        {
            std::lock_guard<std::mutex> g(m_);
            state_[s] = 1;
        }
        seg.slot = s;
        seg.bytes = n;
        push(std::move(seg));
        return true;
    }
    static bool ends_gz(const std::string& p) {
        const size_t n = p.size();
        return n >= 3 && p[n - 3] == '.' && (p[n - 2] == 'g' || p[n - 2] == 'G') && (p[n - 1] == 'z' || p[n - 1] == 'Z');
    }

    void run() {
        (void)hipSetDevice(device_);
        for (int c = 0; c < (int)files_.size(); ++c) {
            if (stopping()) return;
            if (skip_.load() <= c) one_file(c);
            Segment end;
            end.kind = SEG_END;
            end.chunk = c;
            push(std::move(end));
        }
    }

    // text of the file from `offset` on, as the host's readers inflate it
    void text_from(int c, const std::string& path, int64_t offset) {
        std::string err;
        qd_reader* r = qdio::raw_open(path.c_str(), offset, &err);
        if (!r) return fail(c, err);
        for (;;) {
            const uint8_t* p = nullptr;
            size_t len = 0;
            const int rc = qdio::raw_next(r, &p, &len, &err);
            if (rc < 0) fail(c, err);
            if (rc <= 0) break;
            bool ok = true;
            for (size_t at = 0; ok && at < len; at += SEG_BYTES) {
                if (skip_.load() > c || stopping()) {
                    ok = false;
                    break;
                }
                const size_t n = std::min(SEG_BYTES, len - at);
                uint8_t* pin = take_pin();
                if (!pin) {
                    fail(c, "page-locked memory: allocation or upload failed");
                    ok = false;
                    break;
                }
                memcpy(pin, p + at, n);
                Segment seg;
                seg.kind = SEG_TEXT;
                seg.chunk = c;
                ok = upload(seg, pin, n);
            }
            if (!ok) break;
        }
        qdio::raw_close(r);
    }

    void one_file(int c) {
        const std::string& path = files_[c].path;
        const int64_t range_start = files_[c].start, range_end = files_[c].end;
        const bool ranged = range_start > 0 || range_end >= 0;
        if (!ends_gz(path)) {
            if (ranged) return fail(c, path + ": a byte range needs a BGZF file");
            return text_from(c, path, 0);
        }
        const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) return fail(c, path + ": " + strerror(errno));
        std::vector<uint8_t> tail;  // the partial block behind the last whole one of the previous buffer
        int64_t file_pos = range_start;  // file offset of the buffer's first byte
        int64_t read_pos = range_start;  // file offset of the next byte to read
        bool eof = false, first = true;
        int64_t switch_at = -1;     // >= 0: not (or no longer) BGZF from this file offset on
        while (!eof && switch_at < 0) {
            if (skip_.load() > c || stopping()) break;
            uint8_t* pin = take_pin();
            if (!pin) {
                fail(c, "page-locked memory: allocation or upload failed");
                break;
            }
            size_t fill = tail.size();
            if (fill) memcpy(pin, tail.data(), fill);
            tail.clear();
            {
                // one thread copies ~2.5 GB/s out of the page cache, less than the device inflates: the buffer's parts are
                // read side by side on the library's pool (pread at known offsets; the blocks are cut afterwards)
                size_t want = SEG_BYTES - fill;
                if (range_end >= 0) {
                    want = (size_t)std::min<int64_t>((int64_t)want, std::max<int64_t>(range_end - read_pos, 0));
                    if (want == 0) eof = true;
                }
                const int64_t from = read_pos;
                const int parts = (int)((want + READ_PART - 1) / READ_PART);
                std::vector<ssize_t> got((size_t)parts, 0);
                struct {
                    std::mutex m;
                    std::condition_variable cv;
                    int left;
                } latch;
                latch.left = parts;
                for (int q = 0; q < parts; ++q) {
                    uint8_t* dst = pin + fill + (size_t)q * READ_PART;
                    const size_t len = std::min(READ_PART, want - (size_t)q * READ_PART);
                    const int64_t at = from + (int64_t)q * (int64_t)READ_PART;
                    auto job = [fd, dst, len, at, q, &got, &latch] {
                        size_t n = 0;
                        ssize_t r = 0;
                        while (n < len && (r = pread(fd, dst + n, len - n, (off_t)(at + (int64_t)n))) != 0) {
                            if (r < 0) {
                                if (errno == EINTR) continue;
                                break;
                            }
                            n += (size_t)r;
                        }
                        got[(size_t)q] = r < 0 ? -1 : (ssize_t)n;
                        std::lock_guard<std::mutex> g(latch.m);
                        if (--latch.left == 0) latch.cv.notify_all();
                    };
                    if (q + 1 < parts) qdio::pool_submit(job, true);
                    else job();  // (the last part on this thread)
                }
                {
                    std::unique_lock<std::mutex> g(latch.m);
                    latch.cv.wait(g, [&] { return latch.left == 0; });
                }
                for (int q = 0; q < parts; ++q) {
                    if (got[(size_t)q] < 0) {
                        fail(c, path + ": read error");
                        close(fd);
                        return;
                    }
                    fill += (size_t)got[(size_t)q];
                    read_pos += got[(size_t)q];
                    const size_t len = std::min(READ_PART, want - (size_t)q * READ_PART);
                    if ((size_t)got[(size_t)q] < len) {  // the file ends inside this part
                        eof = true;
                        break;
                    }
                }
                if (range_end >= 0 && read_pos >= range_end) eof = true;
            }
            if (first && !qdio::bgzf_block_size(pin, fill)) {  // ordinary gzip
                if (ranged) {
                    fail(c, path + ": no BGZF block at the start of the byte range");
                    close(fd);
                    return;
                }
                if (device_gunzip_ && fill >= 18 && pin[0] == 0x1f && pin[1] == 0x8b && pin[2] == 8) {
                    // the device inflates it (quade_inflate3.hip): the file's bytes go up as they are, upload after upload
                    int64_t at = 0;
                    for (;;) {
                        Segment seg;
                        seg.kind = SEG_GZIP;
                        seg.chunk = c;
                        seg.file_off = at;
                        if (!upload(seg, pin, fill)) break;
                        at += (int64_t)fill;
                        if (eof || skip_.load() > c || stopping()) break;
                        pin = take_pin();
                        if (!pin) {
                            fail(c, "page-locked memory: allocation or upload failed");
                            break;
                        }
                        fill = 0;
                        while (fill < SEG_BYTES) {
                            const ssize_t r = pread(fd, pin + fill, SEG_BYTES - fill, (off_t)(at + (int64_t)fill));
                            if (r < 0 && errno == EINTR) continue;
                            if (r < 0) {
                                fail(c, path + ": read error");
                                close(fd);
                                return;
                            }
                            if (r == 0) {
                                eof = true;
                                break;
                            }
                            fill += (size_t)r;
                        }
                        if (fill == 0) break;
                    }
                    close(fd);
                    return;
                }
                switch_at = 0;  // ... or the host's parallel inflater takes the file
                break;
            }
            first = false;
            Segment seg;
            seg.kind = SEG_BGZF;
            seg.chunk = c;
            seg.file_off = file_pos;
            size_t pos = 0, seg_start = 0;  // seg holds the blocks of pin[seg_start, pos)
            bool upload_failed = false;
            while (pos < fill) {
                const uint8_t* b = pin + pos;
                const size_t avail = fill - pos;
                const size_t bs = qdio::bgzf_block_size(b, avail);
                if (!bs) {
                    if (avail >= 18 || eof) switch_at = file_pos + (int64_t)pos;  // (fewer bytes than a header: the next read completes it)
                    break;
                }
                if (bs > avail) {
                    if (eof) switch_at = file_pos + (int64_t)pos;  // a truncated block: the member loop reports it
                    break;
                }
                const size_t xlen = b[10] | ((size_t)b[11] << 8);
                const uint32_t isize = (uint32_t)b[bs - 4] | ((uint32_t)b[bs - 3] << 8) | ((uint32_t)b[bs - 2] << 16) | ((uint32_t)b[bs - 1] << 24);
                if (bs < 12 + xlen + 8 || (b[3] & ~4) || isize > 65536) {  // not bgzip's layout
                    switch_at = file_pos + (int64_t)pos;
                    break;
                }
                const uint32_t crc = (uint32_t)b[bs - 8] | ((uint32_t)b[bs - 7] << 8) | ((uint32_t)b[bs - 6] << 16) | ((uint32_t)b[bs - 5] << 24);
                const uint32_t in_len = (uint32_t)(bs - 12 - xlen - 8);
                if (seg.text_bytes + isize > SEG_TEXT_MAX && !seg.blocks.empty()) {
                    // text that compresses far better than fastq does: the upload goes out in several segments, so that no window
                    // has to take more than SEG_TEXT_MAX bytes of text in one step (its offsets are 32 bit)
                    if (!upload(seg, pin + seg_start, pos - seg_start)) {
                        upload_failed = true;
                        break;
                    }
                    seg = Segment();
                    seg.kind = SEG_BGZF;
                    seg.chunk = c;
                    seg.file_off = file_pos + (int64_t)pos;
                    seg_start = pos;
                }
                seg.blocks.push_back(qd_inflate_block{(uint32_t)(pos - seg_start + 12 + xlen), in_len, (uint32_t)seg.text_bytes, isize});
                seg.crcs.push_back(crc);
                seg.starts.push_back((uint32_t)(pos - seg_start));
                seg.longest = std::max(seg.longest, in_len);
                seg.text_bytes += isize;
                pos += bs;
            }
            if (upload_failed) break;
            if (switch_at < 0 && pos < fill) tail.assign(pin + pos, pin + fill);
            file_pos += (int64_t)pos;
            if (!seg.blocks.empty() && !upload(seg, pin + seg_start, pos - seg_start)) break;
        }
        close(fd);
        if (switch_at >= 0 && ranged) return fail(c, path + ": the byte range is not BGZF blocks throughout");
        if (switch_at >= 0 && skip_.load() <= c && !stopping()) text_from(c, path, switch_at);
    }

    int device_;
    std::vector<FileSpec> files_;
    bool device_gunzip_ = false;
    hipStream_t up_ = nullptr;
    uint8_t* ring_ = nullptr;
    size_t ring_cap_ = 0;
    hipEvent_t ready_[RING_SLOTS] = {nullptr}, consumed_[RING_SLOTS] = {nullptr};
    int state_[RING_SLOTS] = {0};  // 0 free, 1 with the driver, 2 its last reader is queued (consumed_ tells when it has run)
    uint8_t* pin_[PIN_SLOTS] = {nullptr};
    size_t pin_cap_[PIN_SLOTS] = {0};
    int pin_slot_[PIN_SLOTS] = {-1, -1, -1};  // the ring slot whose upload read this buffer last
    int pin_next_ = 0, pin_cur_ = 0, slot_next_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Segment> q_;
    std::atomic<int> skip_{0};
    bool stop_ = false;
    std::thread th_;
};

// ---- what a batch hands to the collector -------------------------------------------------------------------------------------
struct FileRun {  // the members of one output file made by one batch: pieces [first, first + n)
    uint32_t code;
    int k;
    uint32_t first, n;
    uint64_t text_bytes;
};
struct BatchOut {
    int set = -1;  // resource set; -1: no members, just a message
    qd_sink* sink = nullptr;
    uint32_t n_pieces = 0;
    std::vector<FileRun> files;
    std::vector<qd_deflate_piece> pieces;  // host copy (text_off / text_len: the host codes a piece the device gave up)
    int level = 1;
    hipEvent_t done = nullptr;  // recorded on the compute stream behind the batch's last kernel
    std::string message;        // printed when the batch (and everything before it) is in its files
    bool last = false;
};

struct OutSet {  // per-batch output resources, two of them: the collector drains one while the device fills the other
    DevBuf text, pieces, members, member_len, member_off, packed;
    PinBuf h_len;
    hipEvent_t done = nullptr;
    bool busy = false;
};

}  // namespace

struct qd_pipe_stats_impl {
    int64_t pairs = 0, batches = 0, bgzf_blocks = 0, host_inflated_runs = 0, text_segments = 0, pieces = 0, host_coded_pieces = 0;
    int64_t text_in_bytes = 0, text_out_bytes = 0, gzip_bytes = 0, rescans = 0;
    // where the driver's and the collector's wall time goes (seconds)
    double wait_input = 0, wait_sync = 0, wait_out_set = 0, alloc = 0, collector_wait = 0, download = 0, append = 0, run = 0;
    int64_t gzip_steps = 0, gzip_units = 0, gzip_members = 0, gzip_fallbacks = 0;
};

struct qd_pipe {
    qd_ctx* ctx = nullptr;
    int device = 0;
    std::string err;
    int64_t batch_pairs = 2000000;
    int64_t test_fail_inflate_batch = -1;  // option: the device "refuses" the BGZF blocks of this batch (the host inflates them)
    int64_t test_host_code_every = 0;      // option: every k-th member is coded by the host as if the device had given it up
    int64_t member_slots_bytes = (int64_t)12 << 30;  // option: device memory one batch's member slots may take (sizes the pieces, piece_bytes_for)
    qd_plan plan{};
    qd_layout lay{};
    hipStream_t cs = nullptr, ds = nullptr;
    hipEvent_t sync_ev = nullptr;
    // Option "inflate_streams" = 2: the inflate launches alternate between two streams of their own, so that the next launch
    // (another stream's window, the short index reads' few hundred blocks) fills the CUs a launch's last blocks leave idle; the
    // compute stream joins a window's launches when it first reads the window (join_inflate).  Measured: the launches overlap
    // (their summed time 441 -> 700 ms for the same blocks) and the job gets SLOWER, 0.75 -> 0.82 s per 16 M pairs
    // (profiles/r04_ab_inflate_streams.txt) -- two 150 KB workgroups do not fit a CU, and blocks of two launches competing for
    // the CUs run longer than they save at the tails.  So the default is 1: down the compute stream, one after the other.
    hipStream_t is[2] = {nullptr, nullptr};
    // Option "coder_stream" (default 0; r05, measured and not kept as the default): the tail of a batch -- the coder's launches, the
    // pieces' CRC-32s, the members and their packing -- on a stream of its own behind the batch's format kernel, so that the scans,
    // the row packer, the demultiplexer and the sort of batch k + 1 do not queue behind the coder of batch k.  The tail's tables
    // (sub-blocks, ranges, CRCs, tokens) exist once: the compute stream waits for the tail of batch k before it uploads those of
    // batch k + 1.  It overlaps as designed (profiles/r05_e2e_bgzf_timeline_coder_stream.txt: the coder on a queue of its own beside
    // the whole token launch) and the job is no faster (profiles/r05_coder_stream_ab.txt: 34.5 / 29.8 against 31.2 / 30.8 M pairs/s;
    // single members 24.5 / 24.8 against 24.4 / 25.3): the token launch stretches from 23 to 30-34 ms beside the coder -- the
    // device is busy either way, the batch's time is the sum of its kernels' work.
    hipStream_t es = nullptr;
    hipEvent_t formatted = nullptr, coded = nullptr;
    bool coded_pending = false;
    int coder_stream = 0;
    int peek_records = 1;  // option "peek_records": size a run's buffers and first top-up from the heads of its first files (peek_record_bytes)
    hipEvent_t tables_up = nullptr;  // the launch's block tables are on the device (recorded on cs)
    int n_is = QD_PIPE_INFLATE_STREAMS, next_is = 0;
    DevBuf matches_b;                // the second stream's match lists
    // Option "inflate_form" (default 3): which kernels inflate the BGZF blocks.  3 = quade_inflate3.hip: every symbol decoded once, one
    // lane per block, the blocks of ALL streams' pending uploads in ONE launch (the token kernel's throughput is the number of
    // blocks in flight), then a workgroup per block resolves the tokens; 2 = the speculative spans of quade_inflate.hip, a launch per
    // window and eight uploads.
    int inflate_form = 3;
    struct Queued3 {  // blocks of one window waiting for the next launch
        int stream;
        Feeder* feeder;
        std::vector<int> slots;
        uint32_t first, n, block_base;
    };
    // Option "device_gunzip" (default 1): ordinary gzip files are inflated on the device as well (qd_gz, quade_inflate3.hip)
    int device_gunzip = 1;
    int inflate_overlap = 1;  // the third inflater's launches (and the gzip steps) on a stream of their own
    qd_gz* gz = nullptr;
    int64_t gz_units0 = 0;
    std::vector<qd_inflate3_job> q3_jobs;   // (out: the offset inside the window's text until the launch -- the buffer may still grow)
    std::vector<Queued3> q3_parts;
    DevBuf jobs3, status3, scratch3;
    StagePool stage;
    qd_pipe_stats_impl st;

    struct Window {
        DevBuf buf[2];
        int cur = 0;
        uint32_t len = 0;       // bytes of text in buf[cur]
        bool eof = false;       // the feeder has delivered the stream's last byte
        bool dirty = true;      // text arrived since the last scan
        double avg = 0;         // bytes per record, learned
        uint32_t carry_kept = 0;
        DevBuf tile_counts, tile_base, lines, rec_tile, recs, status, crc, blk, expect;
        uint32_t line_cap = 0;
        std::vector<Segment> pending;                                  // BGZF uploads waiting for their inflate launch
        uint32_t pending_text = 0;
        struct Run {          // a stretch of this batch's BGZF text: what the host inflates again when the device refuses a block
            int64_t file_off;  // where its blocks start in the file
            size_t bytes;      // compressed bytes
            int64_t at;        // window offset of its text (negative: the front of it was dropped by a carry since)
            uint32_t text_bytes;
        };
        std::vector<Run> runs;
        uint32_t n_blocks = 0;                                         // blocks inflated into this window since the last verification
        hipEvent_t inflated[2] = {nullptr, nullptr};                   // the window's last launch on each inflate stream
        bool in_flight[2] = {false, false};
        std::string path;
        qd_scan_result res{};
        // an ordinary gzip file on its way through the device's gzip kernels: the file's bytes [comp_off, comp_off + comp_len) lie in
        // comp[ccur]; the next block header (or, need_header: the next member's header) is known exactly
        struct Gz {
            bool active = false;
            DevBuf comp[2];
            int ccur = 0;
            uint64_t comp_off = 0, comp_len = 0;
            bool file_done = false;   // the feeder has delivered the file's last byte
            bool need_header = true;
            uint64_t hdr_off = 0;     // file offset of the next member's header
            uint64_t bit = 0;         // file position (bits) of the next block header
            DevBuf carried;           // the 32 KiB of text in front of it
            uint32_t carried_valid = 0;
            uint32_t member_crc = 0;
            uint64_t member_text = 0, member_off = 0;
            double ratio = 3.5;       // text per compressed byte, learned
            int fd = -1;
            int64_t file_size = 0;
            uint64_t stepped_end = ~0ull;  // comp_off + comp_len when the last step ran (nothing new since: no step)
            bool stepped_done = false;
            qd_reader* host = nullptr;  // the host's inflater took the stream over (the device gave it up)
            uint64_t host_skip = 0;     // text of the current member that was delivered already
        } gz;
    } win[4];
    int n_streams = 4;  // R1, R2, I1 [, I2]
    DevBuf d_res;       // qd_scan_result[4] + pack's short counter
    PinBuf h_res;
    DevBuf matches;
    // index rows, codes, routing scratch (sized for batch_pairs)
    DevBuf rows_seq[2], rows_qual[2], rows_len[2], codes, mol, short_idx, dest, len1, len2, hist, tmp, perm, sdest, g1, g2, scan_tiles, first, g1_first,
        g2_first;
    PinBuf h_first;
    // tables and scratch of the format / CRC / coder launches: read by kernels on the compute stream only, so one set serves every batch
    DevBuf subs, first_sub, ranges, crc, tokens, sub_out, sub_bytes, base1, base2;
    OutSet out[2];
    PinBuf slab[3];                                // the collector's download ring
    hipEvent_t slab_ev[3] = {nullptr, nullptr, nullptr};
    size_t window_max = WINDOW_MAX;                // text a window may hold (the index pass of a shared chunk raises it for its one window)
    bool reserved = false;                         // the buffers have been sized for batch_pairs (after the first scan)
    std::thread reserve_thread;                    // allocates the output side meanwhile (reserve_output)
    int reserve_rc = QD_OK;
    int64_t max_r1_bytes = 0;                      // size of the run's largest seq_R1 file
    bool r1_compressed = true;
    std::mutex om;
    std::condition_variable ocv;
    // collector
    std::thread collector;
    std::mutex cm;
    std::condition_variable ccv;
    std::deque<BatchOut> cq;
    bool collector_failed = false;
    std::string collector_err;
};

namespace {

thread_local std::string g_pipe_error;

int pfail(qd_pipe* p, int code, const std::string& msg) {
    if (p) {
        if (p->err.empty() || code != QD_OK) p->err = msg;
    } else {
        g_pipe_error = msg;
    }
    return code;
}
#define PCHK(p, call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return pfail((p), QD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// waits until everything queued on the compute stream so far has run (the driver's read-backs)
int sync_compute(qd_pipe* p) {
    Tick tick(p->st.wait_sync);
    PCHK(p, hipEventRecord(p->sync_ev, p->cs));
    PCHK(p, wait_event_napping(p->sync_ev));
    return QD_OK;
}

// ---- collector: members of finished batches -> files ------------------------------------------------------------------------------
void collector_fail(qd_pipe* p, const std::string& msg) {
    std::lock_guard<std::mutex> g(p->cm);
    if (!p->collector_failed) {
        p->collector_failed = true;
        p->collector_err = msg;
    }
}

void collect_one(qd_pipe* p, BatchOut& b) {
    if (b.set >= 0 && b.n_pieces) {
        OutSet& o = p->out[b.set];
        hipError_t e;
        {
            Tick tick(p->st.collector_wait);
            e = wait_event_napping(b.done);
        }
        std::unique_ptr<Tick> dl(new Tick(p->st.download));
        uint64_t* off = nullptr;
        uint32_t* len = nullptr;
        if (e == hipSuccess) e = o.h_len.need((size_t)(b.n_pieces + 1) * 8 + (size_t)b.n_pieces * 4);
        if (e == hipSuccess) {
            off = reinterpret_cast<uint64_t*>(o.h_len.p);
            len = reinterpret_cast<uint32_t*>(o.h_len.p + (size_t)(b.n_pieces + 1) * 8);
            e = hipMemcpyAsync(off, o.member_off.p, (size_t)(b.n_pieces + 1) * 8, hipMemcpyDeviceToHost, p->ds);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(len, o.member_len.p, (size_t)b.n_pieces * 4, hipMemcpyDeviceToHost, p->ds);
        if (e == hipSuccess) e = hipEventRecord(o.done, p->ds);
        if (e == hipSuccess) e = wait_event_napping(o.done);
        const uint64_t total = e == hipSuccess ? off[b.n_pieces] : 0;
        if (e == hipSuccess && p->test_host_code_every > 0)  // test option: every k-th member is treated as one the device gave up
            for (uint32_t i = 0; i < b.n_pieces; i += (uint32_t)p->test_host_code_every) len[i] = 0;
        // members the device gave up (they did not fit their slots: text that does not compress): their text comes back instead
        std::vector<std::vector<uint8_t>> rescue(b.n_pieces);
        std::vector<uint8_t> text;
        for (uint32_t i = 0; e == hipSuccess && i < b.n_pieces; ++i) {
            if (len[i] != 0) continue;
            text.resize(b.pieces[i].text_len);
            if (!text.empty()) e = hipMemcpyAsync(text.data(), o.text.p + b.pieces[i].text_off, text.size(), hipMemcpyDeviceToHost, p->ds);
            if (e == hipSuccess) e = hipStreamSynchronize(p->ds);
            if (e == hipSuccess && !qdio::host_gzip_member(text.data(), text.size(), b.level, &rescue[i])) {
                collector_fail(p, "gzip compression failed");
                break;
            }
            ++p->st.host_coded_pieces;
        }
        dl.reset();
        // The packed members come back through a ring of page-locked slabs (a whole batch's worth of page-locked memory would cost
        // ~1 ms per MB to make); a slab's bytes are appended to their files by jobs on the library's pool, one job per file, while
        // the next slab is on its way.  Slabs are worked off strictly one after the other, so every file sees its bytes in order.
        struct Item {  // the members of the batch in piece order: a run of device members = packed[a, b), or a piece the host coded
            uint32_t file;
            bool host;
            uint64_t a, b;
            uint32_t piece;
        };
        std::vector<Item> items;
        std::vector<void*> files(b.files.size(), nullptr);
        for (size_t fi = 0; e == hipSuccess && fi < b.files.size(); ++fi) {
            const FileRun& f = b.files[fi];
            files[fi] = qdio::sink_file(b.sink, f.code, f.k);  // (created in the order the reference would have: batch, destination)
            for (uint32_t i = f.first; i < f.first + f.n; ++i) {
                if (len[i] == 0) items.push_back(Item{(uint32_t)fi, true, 0, 0, i});
                else if (!items.empty() && !items.back().host && items.back().file == fi && items.back().b == off[i]) items.back().b = off[i + 1];
                else items.push_back(Item{(uint32_t)fi, false, off[i], off[i + 1], i});
            }
        }
        struct Frag {
            uint32_t file;
            const uint8_t* p;
            size_t n;
        };
        struct Latch {
            std::mutex m;
            std::condition_variable cv;
            size_t n = 0;
            void wait() {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [this] { return n == 0; });
            }
        };
        constexpr uint64_t SLAB = 32u << 20;
        constexpr int NSLAB = 3;
        const uint64_t n_slabs = std::max<uint64_t>(1, (total + SLAB - 1) / SLAB);
        for (int k = 0; e == hipSuccess && k < NSLAB && (uint64_t)k < n_slabs; ++k) {
            if (!p->slab[k].p) e = p->slab[k].need(SLAB);
            if (e == hipSuccess && !p->slab_ev[k]) e = hipEventCreateWithFlags(&p->slab_ev[k], hipEventDisableTiming);
        }
        auto fetch = [&](uint64_t i) -> hipError_t {
            const uint64_t A = i * SLAB, B = std::min(total, A + SLAB);
            hipError_t r = hipSuccess;
            if (B > A) r = hipMemcpyAsync(p->slab[i % NSLAB].p, o.packed.p + A, (size_t)(B - A), hipMemcpyDeviceToHost, p->ds);
            if (r == hipSuccess) r = hipEventRecord(p->slab_ev[i % NSLAB], p->ds);
            return r;
        };
        std::unique_ptr<Latch> prev;
        size_t it = 0;
        uint64_t gz_bytes = 0;
        if (e == hipSuccess) e = fetch(0);
        for (uint64_t i = 0; e == hipSuccess && i < n_slabs; ++i) {
            if (i + 1 < n_slabs) e = fetch(i + 1);  // (its slab's jobs -- slab i - 2 -- were waited for below)
            {
                Tick tick(p->st.download);
                if (e == hipSuccess) e = wait_event_napping(p->slab_ev[i % NSLAB]);
            }
            if (e != hipSuccess) break;
            const uint64_t A = i * SLAB, B = std::min(total, A + SLAB);
            const uint8_t* base = p->slab[i % NSLAB].p;
            std::vector<Frag> frags;
            while (it < items.size()) {
                Item& x = items[it];
                if (x.host) {
                    frags.push_back(Frag{x.file, rescue[x.piece].data(), rescue[x.piece].size()});
                    ++it;
                    continue;
                }
                if (x.a >= B && i + 1 < n_slabs) break;  // a later slab's
                const uint64_t lo = std::max(x.a, A), hi = std::min(x.b, B);
                if (hi > lo) frags.push_back(Frag{x.file, base + (lo - A), (size_t)(hi - lo)});
                if (x.b <= B) {
                    ++it;
                } else {
                    x.a = B;
                    break;
                }
            }
            Tick tick(p->st.append);
            if (prev) prev->wait();
            std::unique_ptr<Latch> cur(new Latch());
            std::vector<std::vector<Frag>> jobs;
            for (const Frag& f : frags) {
                if (jobs.empty() || jobs.back().back().file != f.file) jobs.emplace_back();
                jobs.back().push_back(f);
                gz_bytes += f.n;
            }
            cur->n = jobs.size();
            Latch* L = cur.get();
            qd_sink* sink = b.sink;
            for (std::vector<Frag>& j : jobs) {
                void* file = files[j[0].file];
                qdio::pool_submit([file, sink, L, j = std::move(j)] {
                    if (file)
                        for (const Frag& f : j) qdio::sink_append(sink, file, f.p, f.n);
                    std::lock_guard<std::mutex> g(L->m);
                    if (--L->n == 0) L->cv.notify_all();
                });
            }
            prev = std::move(cur);
        }
        if (prev) {
            Tick tick(p->st.append);
            prev->wait();
        }
        if (e != hipSuccess) {
            collector_fail(p, std::string("download of the members: ") + hipGetErrorString(e));
        } else {
            for (const FileRun& f : b.files) qdio::sink_account(b.sink, f.n, f.n, (int64_t)f.text_bytes, 0);
            qdio::sink_account(b.sink, 0, 0, 0, (int64_t)gz_bytes);
            p->st.gzip_bytes += (int64_t)gz_bytes;
        }
        {
            std::lock_guard<std::mutex> g(p->om);
            o.busy = false;
        }
        p->ocv.notify_all();
    }
    if (!b.message.empty()) {
        fputs(b.message.c_str(), stdout);
        fflush(stdout);
    }
}

void collector_thread(qd_pipe* p) {
    (void)hipSetDevice(p->device);
    for (;;) {
        BatchOut b;
        {
            std::unique_lock<std::mutex> g(p->cm);
            p->ccv.wait(g, [p] { return !p->cq.empty(); });
            b = std::move(p->cq.front());
            p->cq.pop_front();
        }
        collect_one(p, b);
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.last) {
            std::lock_guard<std::mutex> g(p->cm);
            p->ccv.notify_all();
            return;
        }
    }
}

void to_collector(qd_pipe* p, BatchOut&& b) {
    {
        std::lock_guard<std::mutex> g(p->cm);
        p->cq.push_back(std::move(b));
    }
    p->ccv.notify_all();
}

// ---- driver ------------------------------------------------------------------------------------------------------------------------
using Window = qd_pipe::Window;

// the compute stream goes on behind the window's inflate launches
int join_inflate(qd_pipe* p, Window& w) {
    for (int k = 0; k < 2; ++k)
        if (w.in_flight[k]) {
            PCHK(p, hipStreamWaitEvent(p->cs, w.inflated[k], 0));
            w.in_flight[k] = false;
        }
    return QD_OK;
}

// room for `extra` more bytes of text in the window (the line kernels read whole tiles: padding behind the text)
int window_room(qd_pipe* p, Window& w, size_t extra) {
    const size_t need = (size_t)w.len + extra + 2 * QD_TEXT_TILE;
    if ((size_t)w.len + extra > p->window_max) return pfail(p, QD_ERR_UNSUPPORTED, w.path + ": more text in one window than its 32-bit offsets reach (lower batch_pairs)");
    PCHK(p, w.buf[w.cur].need(need, w.len, p->cs));
    return QD_OK;
}

constexpr uint32_t LAUNCH_BLOCKS3 = 32768;  // blocks per launch of the third inflater at most (sizes its token scratch: 131 KB a block)
constexpr size_t FLUSH_SEGMENTS3 = RING_SLOTS * 3 / 4;  // uploads one window may queue before a launch frees their ring slots

// the third inflater: everything queued goes down in one launch (more when it exceeds LAUNCH_BLOCKS3), every window's blocks are checked
int flush_inflate3(qd_pipe* p) {
    if (p->q3_jobs.empty()) return QD_OK;
    const size_t n = p->q3_jobs.size();
    // Option "inflate_overlap" (default 1): the launches go down a stream of their own, so that the token kernel of batch k + 1 -- a
    // latency, a wave per SIMD at most -- shares the device with what batch k still has queued on the compute stream (format, coder); the
    // windows' scans wait for the event (join_inflate).  It reads the upload ring and writes the windows' text behind what the carry
    // copies (compute stream) move to their front: no overlap with anything queued there.
    const hipStream_t xs = p->inflate_overlap ? p->is[0] : p->cs;
    for (const qd_pipe::Queued3& q : p->q3_parts) {
        Window& w = p->win[q.stream];
        for (uint32_t i = q.first; i < q.first + q.n; ++i) p->q3_jobs[i].out = w.buf[w.cur].p + reinterpret_cast<uintptr_t>(p->q3_jobs[i].out);
        for (int slot : q.slots) PCHK(p, hipStreamWaitEvent(xs, q.feeder->ready(slot), 0));
    }
    PCHK(p, p->jobs3.need(n * sizeof(qd_inflate3_job), 0, p->cs));
    PCHK(p, p->status3.need(n * 4, 0, p->cs));
    PCHK(p, p->scratch3.need(qd_inflate3_scratch_bytes((uint32_t)std::min<size_t>(n, LAUNCH_BLOCKS3)), 0, p->cs));
    PCHK(p, p->stage.upload(p->jobs3.p, p->q3_jobs.data(), n * sizeof(qd_inflate3_job), xs));
    for (size_t at = 0; at < n; at += LAUNCH_BLOCKS3) {
        const uint32_t m = (uint32_t)std::min<size_t>(LAUNCH_BLOCKS3, n - at);
        PCHK(p, qd_launch_inflate3_jobs(p->jobs3.as<qd_inflate3_job>() + at, m, p->status3.as<int32_t>() + at, p->scratch3.p, xs));
    }
    for (const qd_pipe::Queued3& q : p->q3_parts) {
        for (int slot : q.slots) PCHK(p, q.feeder->consumed(slot, xs));
        const uint32_t* st = p->status3.as<uint32_t>() + q.first;  // (a block's CRC-32 is checked by its resolve kernel: the statuses say it all)
        PCHK(p, qd_text_check_blocks(p->status3.as<int32_t>() + q.first, st, st, q.n, q.block_base, &p->d_res.as<qd_scan_result>()[q.stream].first_bad, xs));
    }
    if (xs != p->cs) {
        for (const qd_pipe::Queued3& q : p->q3_parts) {
            Window& w = p->win[q.stream];
            if (!w.inflated[0]) PCHK(p, hipEventCreateWithFlags(&w.inflated[0], hipEventDisableTiming));
            PCHK(p, hipEventRecord(w.inflated[0], xs));
            w.in_flight[0] = true;
        }
    }
    p->q3_jobs.clear();
    p->q3_parts.clear();
    return QD_OK;
}

// the pending BGZF uploads of a window -> inflate launches (+ CRC-32 check of every block), text appended to the window
int launch_inflate(qd_pipe* p, Feeder& f, Window& w, int stream_index) {
    if (w.pending.empty()) return QD_OK;
    int rc = window_room(p, w, w.pending_text);
    if (rc != QD_OK) return rc;
    if (p->inflate_form == 3) {  // queued: flush_inflate3 launches every window's blocks together
        qd_pipe::Queued3 q;
        q.stream = stream_index;
        q.feeder = &f;
        q.first = (uint32_t)p->q3_jobs.size();
        q.block_base = w.n_blocks;
        for (Segment& s : w.pending) {
            const uint8_t* base = f.ring() + (size_t)s.slot * SEG_BYTES;
            for (size_t i = 0; i < s.blocks.size(); ++i) {
                const qd_inflate_block& b = s.blocks[i];
                qd_inflate3_job j;
                j.payload = base + b.in_off;
                j.out = reinterpret_cast<uint8_t*>((uintptr_t)w.len + b.out_off);
                j.in_len = b.in_len;
                j.out_len = b.out_len;
                j.expect_crc = s.crcs[i];
                j.check_crc = 1;
                p->q3_jobs.push_back(j);
            }
            q.slots.push_back(s.slot);
            w.runs.push_back(Window::Run{s.file_off, s.bytes, (int64_t)w.len, (uint32_t)s.text_bytes});
            w.len += (uint32_t)s.text_bytes;
            p->st.text_in_bytes += (int64_t)s.text_bytes;
        }
        q.n = (uint32_t)p->q3_jobs.size() - q.first;
        w.n_blocks += q.n;
        p->st.bgzf_blocks += (int64_t)q.n;
        w.pending.clear();
        w.pending_text = 0;
        w.dirty = true;
        p->q3_parts.push_back(std::move(q));
        size_t queued = 0;
        for (const qd_pipe::Queued3& x : p->q3_parts)
            if (x.stream == stream_index) queued += x.slots.size();
        return queued >= FLUSH_SEGMENTS3 ? flush_inflate3(p) : QD_OK;
    }
    std::vector<qd_inflate_block> blk;
    std::vector<uint32_t> expect;
    uint32_t longest = 0;
    const int xi = p->n_is > 1 ? p->next_is : -1;
    if (xi >= 0) p->next_is ^= 1;
    const hipStream_t xs = xi >= 0 ? p->is[xi] : p->cs;
    for (Segment& s : w.pending) {
        PCHK(p, hipStreamWaitEvent(xs, f.ready(s.slot), 0));
        for (size_t i = 0; i < s.blocks.size(); ++i) {
            qd_inflate_block b = s.blocks[i];
            b.in_off += (uint32_t)((size_t)s.slot * SEG_BYTES);
            b.out_off += w.len;
            blk.push_back(b);
            expect.push_back(s.crcs[i]);
        }
        longest = std::max(longest, s.longest);
        w.runs.push_back(Window::Run{s.file_off, s.bytes, (int64_t)w.len, (uint32_t)s.text_bytes});
        w.len += (uint32_t)s.text_bytes;
        p->st.text_in_bytes += (int64_t)s.text_bytes;
    }
    const size_t nb = blk.size();
    PCHK(p, w.blk.need((size_t)(w.n_blocks + nb) * sizeof(qd_inflate_block), 0, p->cs));
    PCHK(p, w.expect.need((size_t)(w.n_blocks + nb) * 4, 0, p->cs));
    PCHK(p, w.status.need((size_t)(w.n_blocks + nb) * 4, 0, p->cs));
    PCHK(p, w.crc.need((size_t)(w.n_blocks + nb) * 4, 0, p->cs));
    // (tables of this launch group go behind those of the batch's earlier groups: a growth above drains the stream first)
    qd_inflate_block* d_blk = w.blk.as<qd_inflate_block>() + w.n_blocks;
    uint32_t* d_expect = w.expect.as<uint32_t>() + w.n_blocks;
    int32_t* d_status = w.status.as<int32_t>() + w.n_blocks;
    uint32_t* d_crc = w.crc.as<uint32_t>() + w.n_blocks;
    PCHK(p, p->stage.upload(d_blk, blk.data(), nb * sizeof(qd_inflate_block), p->cs));
    PCHK(p, p->stage.upload(d_expect, expect.data(), nb * 4, p->cs));
    uint32_t* first_bad = &p->d_res.as<qd_scan_result>()[stream_index].first_bad;
    const bool form2 = qd_inflate2_lds(longest) <= 160 * 1024;
    DevBuf& matches = xi == 1 ? p->matches_b : p->matches;
    if (form2) PCHK(p, matches.need((size_t)std::min<size_t>(nb, LAUNCH_BLOCKS) * QD_INFLATE_MATCHES_PER_BLOCK * 8, 0, p->cs));
    if (xi >= 0) {  // behind everything the compute stream has queued so far: the tables above, the last readers of the window's buffer
        PCHK(p, hipEventRecord(p->tables_up, p->cs));
        PCHK(p, hipStreamWaitEvent(xs, p->tables_up, 0));
    }
    for (size_t at = 0; at < nb; at += LAUNCH_BLOCKS) {
        const uint32_t n = (uint32_t)std::min<size_t>(LAUNCH_BLOCKS, nb - at);
        if (form2) {
            // (every block's CRC-32 against its trailer is checked by the kernel, while the text is in LDS)
            PCHK(p, qd_launch_inflate2(f.ring(), d_blk + at, n, w.buf[w.cur].p, d_status + at, matches.as<unsigned long long>(), QD_INFLATE_MATCHES_PER_BLOCK,
                                       longest, xs, nullptr, d_expect + at));
        } else {
            PCHK(p, qd_launch_inflate(f.ring(), d_blk + at, n, w.buf[w.cur].p, d_status + at, xs));
        }
    }
    for (Segment& s : w.pending) PCHK(p, f.consumed(s.slot, xs));
    if (form2) {
        PCHK(p, qd_text_check_blocks(d_status, d_expect, d_expect, (uint32_t)nb, w.n_blocks, first_bad, xs));  // (the statuses say it all)
    } else {  // the one-wave form does not check: a CRC-32 pass over the text, one range per block
        PCHK(p, qd_text_crc32_blocks(w.buf[w.cur].p, d_blk, (uint32_t)nb, d_crc, xs));
        PCHK(p, qd_text_check_blocks(d_status, d_crc, d_expect, (uint32_t)nb, w.n_blocks, first_bad, xs));
    }
    if (xi >= 0) {
        if (!w.inflated[xi]) PCHK(p, hipEventCreateWithFlags(&w.inflated[xi], hipEventDisableTiming));
        PCHK(p, hipEventRecord(w.inflated[xi], xs));
        w.in_flight[xi] = true;
    }
    w.n_blocks += (uint32_t)nb;
    p->st.bgzf_blocks += (int64_t)nb;
    w.pending.clear();
    w.pending_text = 0;
    w.dirty = true;
    return QD_OK;
}

// ---- ordinary gzip files through the device's gzip kernels (qd_gz) --------------------------------------------------------------------
void gz_close(Window& w) {
    Window::Gz& g = w.gz;
    if (g.fd >= 0) close(g.fd);
    if (g.host) qdio::raw_close(g.host);
    g.fd = -1;
    g.host = nullptr;
    g.active = false;
    g.comp_off = g.comp_len = 0;
    g.file_done = false;
    g.need_header = true;
    g.hdr_off = g.bit = 0;
    g.carried_valid = 0;
    g.member_crc = 0;
    g.member_text = g.member_off = g.host_skip = 0;
    g.stepped_end = ~0ull;
    g.stepped_done = false;
}

// bytes of a gzip member's header at p[0 .. n) (RFC 1952); 0: none, or it does not end inside n
size_t gz_header_bytes(const uint8_t* h, size_t n) {
    if (n < 10 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xe0)) return 0;
    const int flg = h[3];
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return 0;
        at += 2 + (h[at] | ((size_t)h[at + 1] << 8));
    }
    for (int bit : {8, 16})
        if (flg & bit) {
            while (at < n && h[at]) ++at;
            ++at;
        }
    if (flg & 2) at += 2;
    return at <= n ? at : 0;
}

// the window's file is an ordinary gzip file that the device inflates: its state (idempotent)
int gz_activate(qd_pipe* p, Window& w) {
    Window::Gz& g = w.gz;
    if (g.active) return QD_OK;
    const double ratio = g.ratio;  // (a look at the file's head may have told: gz_close keeps nothing)
    gz_close(w);
    g.ratio = ratio;
    g.active = true;
    g.fd = open(w.path.c_str(), O_RDONLY | O_CLOEXEC);
    struct stat sb;
    if (g.fd < 0 || fstat(g.fd, &sb) != 0) return pfail(p, QD_ERR_FORMAT, w.path + ": " + strerror(errno));
    g.file_size = (int64_t)sb.st_size;
    PCHK(p, g.carried.need(32768, 0, p->cs));
    if (!p->gz) {
        p->gz = new qd_gz();
        p->gz_units0 = 0;
    }
    return QD_OK;
}

// an upload of the file's bytes joins the stream's compressed buffer
int gz_append(qd_pipe* p, Feeder& f, Window& w, Segment& s) {
    Window::Gz& g = w.gz;
    {
        const int rc = gz_activate(p, w);
        if (rc != QD_OK) return rc;
    }
    if ((uint64_t)s.file_off != g.comp_off + g.comp_len) return pfail(p, QD_ERR_STATE, w.path + ": uploads out of order");
    const hipStream_t gs = p->inflate_overlap ? p->is[0] : p->cs;  // (the gzip steps' stream: gz_steps)
    PCHK(p, g.comp[g.ccur].need((size_t)g.comp_len + s.bytes + 8192, (size_t)g.comp_len, gs));
    PCHK(p, hipStreamWaitEvent(gs, f.ready(s.slot), 0));
    PCHK(p, hipMemcpyAsync(g.comp[g.ccur].p + g.comp_len, f.ring() + (size_t)s.slot * SEG_BYTES, s.bytes, hipMemcpyDeviceToDevice, gs));
    PCHK(p, f.consumed(s.slot, gs));
    g.comp_len += s.bytes;
    return QD_OK;
}

// The device gave the stream up (damaged, or beyond what it decodes): the host's inflater reads the current member again from its
// start, the text that was delivered already is skipped, and what follows reaches the window as text.  Counted (gzip_fallbacks).
int gz_fallback(qd_pipe* p, Feeder& f, Window& w, int chunk) {
    Window::Gz& g = w.gz;
    ++p->st.gzip_fallbacks;
    while (!g.file_done) {  // what the feeder still uploads of this file is dropped
        Segment s = f.pop();
        if (s.kind == SEG_ERROR) return pfail(p, QD_ERR_FORMAT, s.err);
        if (s.kind == SEG_END) {
            if (s.chunk == chunk) g.file_done = true;
            continue;
        }
        PCHK(p, f.consumed(s.slot, p->cs));
    }
    std::string err;
    g.host = qdio::raw_open(w.path.c_str(), (int64_t)g.member_off, &err);
    if (!g.host) return pfail(p, QD_ERR_FORMAT, err);
    g.host_skip = g.member_text;
    g.comp_len = 0;
    return QD_OK;
}

// text from the host's inflater until the window holds `want` bytes or the file ends
int gz_host_fill(qd_pipe* p, Window& w, size_t want) {
    Window::Gz& g = w.gz;
    while (!w.eof && (size_t)w.len < want) {
        const uint8_t* ptr = nullptr;
        size_t len = 0;
        std::string err;
        const int rc = qdio::raw_next(g.host, &ptr, &len, &err);
        if (rc < 0) return pfail(p, QD_ERR_FORMAT, err);
        if (rc == 0) {
            w.eof = true;
            w.dirty = true;
            break;
        }
        if (g.host_skip) {
            const size_t drop = (size_t)std::min<uint64_t>(g.host_skip, len);
            g.host_skip -= drop;
            ptr += drop;
            len -= drop;
        }
        for (size_t at = 0; at < len;) {
            const size_t n = std::min<size_t>(len - at, (size_t)256 << 20);
            const int r = window_room(p, w, n);
            if (r != QD_OK) return r;
            PCHK(p, hipStreamSynchronize(p->cs));
            PCHK(p, hipMemcpy(w.buf[w.cur].p + w.len, ptr + at, n, hipMemcpyHostToDevice));
            w.len += (uint32_t)n;
            at += n;
            w.dirty = true;
            ++p->st.text_segments;
            p->st.text_in_bytes += (int64_t)n;
        }
    }
    return QD_OK;
}

// One inflate step of every gzip stream that has compressed bytes waiting: probe + tokens for all of them in one launch each, then
// the text appended to the windows, the members' trailers checked.  `feeders`: for a stream that falls back to the host.
int gz_steps(qd_pipe* p, std::vector<std::unique_ptr<Feeder>>& feeders, int chunk) {
    std::vector<qd_gz_step> steps;
    std::vector<int> who;
    std::vector<uint64_t> byte0s;
    for (int s = 0; s < p->n_streams; ++s) {
        Window& w = p->win[s];
        Window::Gz& g = w.gz;
        if (!g.active || g.host || w.eof) continue;
        if (g.need_header) {  // the next member's header (from the file: the compressed bytes live on the device)
            if ((int64_t)g.hdr_off >= g.file_size) {
                if (g.file_done) {
                    w.eof = true;
                    w.dirty = true;
                }
                continue;
            }
            uint8_t h[65536];
            const ssize_t got = pread(g.fd, h, sizeof h, (off_t)g.hdr_off);
            const size_t hb = got > 0 ? gz_header_bytes(h, (size_t)got) : 0;
            if (!hb) {
                bool zeros = got > 0;
                for (ssize_t k = 0; k < got && zeros; ++k) zeros = h[k] == 0;
                if (zeros && (int64_t)g.hdr_off + got >= g.file_size) {  // (padding behind the last member)
                    g.hdr_off = (uint64_t)g.file_size;
                    w.eof = g.file_done;
                    w.dirty = true;
                    continue;
                }
                g.member_off = g.hdr_off;
                g.member_text = 0;
                const int rc = gz_fallback(p, *feeders[(size_t)s], w, chunk);  // (the host's reader says what is wrong with it)
                if (rc != QD_OK) return rc;
                continue;
            }
            g.member_off = g.hdr_off;
            g.member_text = 0;
            g.member_crc = 0;
            g.carried_valid = 0;
            g.bit = 8 * (g.hdr_off + hb);
            g.need_header = false;
        }
        if (g.bit >= 8 * (g.comp_off + g.comp_len)) continue;  // nothing of it on the device yet
        // (a step that made no progress -- the bytes on the device end inside a block -- is not tried again until more have arrived)
        if (g.stepped_end == g.comp_off + g.comp_len && g.stepped_done == g.file_done) continue;
        qd_gz_step st{};
        const uint64_t byte0 = ((g.bit >> 3) - g.comp_off) & ~(uint64_t)15;
        st.comp = g.comp[g.ccur].p + byte0;
        st.comp_bytes = g.comp_len - byte0;
        st.bit_start = g.bit - 8 * (g.comp_off + byte0);
        st.at_end = g.file_done ? 1 : 0;
        st.carried = g.carried.p;
        st.carried_valid = g.carried_valid;
        steps.push_back(st);
        who.push_back(s);
        byte0s.push_back(byte0);
    }
    if (steps.empty()) return QD_OK;
    const int n = (int)steps.size();
    // A stream of their own ("inflate_overlap"): a step ends with the host waiting for its results, and on the compute stream that wait
    // would include everything batch k still has queued there (format, coder).  What a step writes -- the windows' text behind what
    // the carry copies move to their front -- nothing queued on the compute stream touches.
    // (is[0], the BGZF launches' stream too: the runtime spreads streams over four hardware queues, and is[1] landed on the compute
    //  stream's -- its kernels ran strictly one after the other with the coder's, profiles/r05_e2e_gz_timeline_before.txt)
    const hipStream_t gs = p->inflate_overlap ? p->is[0] : p->cs;
    static double t_decode = 0, t_room = 0, t_resolve = 0, t_sync = 0, t_post = 0;  // (QUADE_PIPE_TRACE: where a gzip step's wall time goes)
    static const bool trace = getenv("QUADE_PIPE_TRACE") != nullptr;
    struct Report {
        ~Report() {
            if (trace) fprintf(stderr, "[pipe] gzip steps so far: decode %.3f s, window room %.3f s, resolve (host side) %.3f s, sync + finish %.3f s, bookkeeping %.3f s\n",
                               t_decode, t_room, t_resolve, t_sync, t_post);
        }
    } report;
    {
        Tick tick(p->st.wait_sync);
        Tick t2(t_decode);
        PCHK(p, p->gz->decode(steps.data(), n, gs));
    }
    p->st.gzip_steps += n;
    std::vector<uint8_t*> out((size_t)n, nullptr);
    std::unique_ptr<Tick> tk(new Tick(t_room));
    for (int i = 0; i < n; ++i) {
        Window& w = p->win[who[(size_t)i]];
        if (steps[(size_t)i].failed) continue;
        if (steps[(size_t)i].text_len > 0xF0000000ull) return pfail(p, QD_ERR_UNSUPPORTED, w.path + ": more text in one step than a window's 32-bit offsets reach");
        const int rc = window_room(p, w, (size_t)steps[(size_t)i].text_len);
        if (rc != QD_OK) return rc;
        out[(size_t)i] = w.buf[w.cur].p + w.len;
    }
    tk.reset(new Tick(t_resolve));
    PCHK(p, p->gz->resolve(steps.data(), n, out.data(), gs));
    tk.reset(new Tick(t_sync));
    if (gs == p->cs) {
        const int rc = sync_compute(p);
        if (rc != QD_OK) return rc;
    } else {
        Tick tick(p->st.wait_sync);
        PCHK(p, hipStreamSynchronize(gs));
    }
    PCHK(p, p->gz->finish(steps.data(), n));
    tk.reset(new Tick(t_post));
    for (int i = 0; i < n; ++i) {
        const int s = who[(size_t)i];
        Window& w = p->win[s];
        Window::Gz& g = w.gz;
        qd_gz_step& st = steps[(size_t)i];
        if (st.failed) {
            const int rc = gz_fallback(p, *feeders[(size_t)s], w, chunk);
            if (rc != QD_OK) return rc;
            continue;
        }
        if (st.text_len) {
            g.member_crc = g.member_text ? qd_crc32_combine_host(g.member_crc, st.crc32, st.text_len) : st.crc32;
            g.member_text += st.text_len;
            w.len += (uint32_t)st.text_len;
            w.dirty = true;
            p->st.text_in_bytes += (int64_t)st.text_len;
            const uint64_t used = (st.bit_next - st.bit_start + 7) / 8;
            if (used) g.ratio = 0.5 * g.ratio + 0.5 * std::min(64.0, std::max(1.0, (double)st.text_len / (double)used));
        }
        g.carried_valid = st.carried_valid;
        const bool progress = st.bit_next != st.bit_start || st.member_end;
        if (!progress && !st.starved) {
            g.stepped_end = g.comp_off + g.comp_len;
            g.stepped_done = g.file_done;
        }
        g.bit = 8 * (g.comp_off + byte0s[(size_t)i]) + st.bit_next;
        if (st.member_end) {  // the trailer: CRC-32 and ISIZE of the member's text
            const uint64_t tr = (g.bit + 7) >> 3;
            uint8_t t[8];
            if (pread(g.fd, t, 8, (off_t)tr) != 8) {
                const int rc = gz_fallback(p, *feeders[(size_t)s], w, chunk);  // (truncated: the host's reader reports it)
                if (rc != QD_OK) return rc;
                continue;
            }
            const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t want_len = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (want_crc != (g.member_text ? g.member_crc : 0u) || want_len != (uint32_t)g.member_text)
                return pfail(p, QD_ERR_FORMAT, w.path + ": CRC-32 or length of a gzip member does not match its text");
            ++p->st.gzip_members;
            g.need_header = true;
            g.hdr_off = tr + 8;
            g.bit = 8 * g.hdr_off;
        } else if (!progress && g.file_done && !st.starved) {
            const int rc = gz_fallback(p, *feeders[(size_t)s], w, chunk);  // the stream ends inside a block: the host's reader reports it
            if (rc != QD_OK) return rc;
            continue;
        }
        // what lies in front of the next block header has been used: the rest moves to the front of the other buffer
        const uint64_t keep_from = std::min<uint64_t>(((g.bit >> 3) - g.comp_off) & ~(uint64_t)15, g.comp_len);
        if (keep_from) {
            const uint64_t left = g.comp_len - keep_from;
            const int nx = g.ccur ^ 1;
            PCHK(p, g.comp[nx].need((size_t)left + 8192, 0, gs));
            if (left) PCHK(p, hipMemcpyAsync(g.comp[nx].p, g.comp[g.ccur].p + keep_from, (size_t)left, hipMemcpyDeviceToDevice, gs));
            g.ccur = nx;
            g.comp_off += keep_from;
            g.comp_len = left;
        }
        if (g.file_done && g.need_header && (int64_t)g.hdr_off >= g.file_size) {
            w.eof = true;
            w.dirty = true;
        }
    }
    return QD_OK;
}

// more input for one window until it holds `want` bytes of text or its stream ends
int top_up(qd_pipe* p, Feeder& f, Window& w, int stream_index, int chunk, size_t want) {
    if (w.gz.active && w.gz.host) return gz_host_fill(p, w, want);
    while (!w.eof && (size_t)w.len + w.pending_text < want) {
        if (w.gz.active) {
            // (an estimate of the text the compressed bytes on the device stand for: gz_steps will tell)
            const uint64_t held = 8 * (w.gz.comp_off + w.gz.comp_len) > w.gz.bit ? w.gz.comp_off + w.gz.comp_len - (w.gz.bit >> 3) : 0;
            const bool stuck = w.gz.stepped_end == w.gz.comp_off + w.gz.comp_len;  // (the last step could use none of what is held: more must come)
            if (w.gz.file_done || (!stuck && (double)w.len + (double)held * w.gz.ratio >= (double)want)) break;
        }
        Segment s;
        {
            Tick tick(p->st.wait_input);
            s = f.pop();
        }
        if (s.kind == SEG_ERROR) return pfail(p, QD_ERR_FORMAT, s.err);
        if (s.kind == SEG_END) {
            if (s.chunk != chunk) return pfail(p, QD_ERR_STATE, "feeder out of step with the driver");
            if (w.gz.active) {  // (its text is still to be made: gz_steps says when the stream has ended)
                w.gz.file_done = true;
                break;
            }
            w.eof = true;
            w.dirty = true;
            break;
        }
        if (s.chunk != chunk) {  // left over from a chunk that ended early
            PCHK(p, f.consumed(s.slot, p->cs));
            continue;
        }
        if (s.kind == SEG_GZIP) {
            const int rc = gz_append(p, f, w, s);
            if (rc != QD_OK) return rc;
            continue;
        }
        if (s.kind == SEG_BGZF) {
            w.pending_text += (uint32_t)s.text_bytes;
            w.pending.push_back(std::move(s));
            if (w.pending.size() >= GROUP_SEGMENTS) {
                const int rc = launch_inflate(p, f, w, stream_index);
                if (rc != QD_OK) return rc;
            }
        } else {  // text: straight into the window
            int rc = launch_inflate(p, f, w, stream_index);  // (order: BGZF text that came first lands first)
            if (rc == QD_OK) rc = window_room(p, w, s.bytes);
            if (rc != QD_OK) return rc;
            PCHK(p, hipStreamWaitEvent(p->cs, f.ready(s.slot), 0));
            PCHK(p, hipMemcpyAsync(w.buf[w.cur].p + w.len, f.ring() + (size_t)s.slot * SEG_BYTES, s.bytes, hipMemcpyDeviceToDevice, p->cs));
            PCHK(p, f.consumed(s.slot, p->cs));
            w.len += (uint32_t)s.bytes;
            w.dirty = true;
            ++p->st.text_segments;
            p->st.text_in_bytes += (int64_t)s.bytes;
        }
    }
    return launch_inflate(p, f, w, stream_index);
}

// drops what is left of `chunk` in a stream (the chunk ended with another stream)
int drain_chunk(qd_pipe* p, Feeder& f, Window& w, int chunk) {
    for (Segment& s : w.pending) PCHK(p, f.consumed(s.slot, p->cs));
    w.pending.clear();
    w.pending_text = 0;
    if (w.gz.active && w.gz.file_done) w.eof = true;  // (the feeder has closed this chunk's file already)
    while (!w.eof) {
        Segment s = f.pop();
        if (s.kind == SEG_ERROR) continue;  // (of a stream nobody reads any more)
        if (s.kind == SEG_END) {
            if (s.chunk == chunk) w.eof = true;
            continue;
        }
        PCHK(p, f.consumed(s.slot, p->cs));
    }
    return QD_OK;
}

int scan_window(qd_pipe* p, Window& w, int stream_index, qd_scan_job* job = nullptr) {
    if (join_inflate(p, w) != QD_OK) return QD_ERR_HIP;
    const uint32_t n_tiles = w.len / QD_TEXT_TILE + 1;
    PCHK(p, w.tile_counts.need((size_t)(n_tiles + 2) * 4, 0, p->cs));
    PCHK(p, w.tile_base.need((size_t)(n_tiles + 2) * 4, 0, p->cs));
    if (w.line_cap == 0 || (w.avg > 0 && (double)w.len / w.avg * 4.4 + 4096 > (double)w.line_cap)) {
        // lines expected: 4 per record of the learned size (first scan: one line per 16 bytes); a scan that finds more says so
        const double guess = w.avg > 0 ? (double)w.len / w.avg * 4.4 : (double)w.len / 16.0;
        w.line_cap = (uint32_t)std::min<double>(4.0e9, guess * 1.25 + 65536);
        w.line_cap = (w.line_cap + 3) & ~3u;
    }
    {
        PCHK(p, w.lines.need((size_t)w.line_cap * 4 + 64, 0, p->cs));
        PCHK(p, w.rec_tile.need(((size_t)w.line_cap / 4 / 1024 + 4) * 4, 0, p->cs));
        PCHK(p, w.recs.need(((size_t)w.line_cap / 4 + 1) * sizeof(qd_rec), 0, p->cs));
        qd_scan_scratch sc;
        sc.tile_counts = w.tile_counts.as<uint32_t>();
        sc.tile_base = w.tile_base.as<uint32_t>();
        sc.lines = w.lines.as<uint32_t>();
        sc.line_cap = w.line_cap;
        sc.rec_tile = w.rec_tile.as<uint32_t>();
        sc.recs = w.recs.as<qd_rec>();
        const bool insert = stream_index < 2;
        const int k = stream_index - 2;
        const uint32_t need = insert ? 0u : (uint32_t)(p->lay.seq_off[k] + p->lay.seq_width[k]);
        if (job) {  // the caller launches several windows' scans together (scan_windows)
            *job = qd_scan_job{w.buf[w.cur].p, w.len, w.eof ? 1 : 0, insert ? 1 : 0, need, sc, p->d_res.as<qd_scan_result>() + stream_index};
            return QD_OK;
        }
        PCHK(p, qd_text_scan(w.buf[w.cur].p, w.len, w.eof ? 1 : 0, insert ? 1 : 0, need, sc, p->d_res.as<qd_scan_result>() + stream_index, p->cs));
        PCHK(p, hipMemcpyAsync(p->h_res.p + (size_t)stream_index * sizeof(qd_scan_result), p->d_res.p + (size_t)stream_index * sizeof(qd_scan_result),
                               sizeof(qd_scan_result), hipMemcpyDeviceToHost, p->cs));
    }
    return QD_OK;
}

// the scans of the windows that changed, stage by stage for all of them (qd_text_scan_many); false: none was dirty
int scan_windows(qd_pipe* p, bool* scanned) {
    qd_scan_job jobs[4];
    int which[4], n = 0;
    for (int s = 0; s < p->n_streams; ++s) {
        Window& w = p->win[s];
        if (!w.dirty) continue;
        const int rc = scan_window(p, w, s, &jobs[n]);
        if (rc != QD_OK) return rc;
        which[n++] = s;
    }
    *scanned = n > 0;
    if (!n) return QD_OK;
    static const bool one_by_one = [] {  // (A/B: QUADE_PIPE_SCAN_MANY=0 launches window by window, as before r05)
        const char* e = getenv("QUADE_PIPE_SCAN_MANY");
        return e && atoi(e) == 0;
    }();
    if (one_by_one) {
        for (int k = 0; k < n; ++k)
            PCHK(p, qd_text_scan(jobs[k].text, jobs[k].len, jobs[k].at_eof, jobs[k].want_names, jobs[k].need, jobs[k].s, jobs[k].result, p->cs));
    } else {
        PCHK(p, qd_text_scan_many(n, jobs, p->cs));
    }
    for (int k = 0; k < n; ++k)
        PCHK(p, hipMemcpyAsync(p->h_res.p + (size_t)which[k] * sizeof(qd_scan_result), p->d_res.p + (size_t)which[k] * sizeof(qd_scan_result), sizeof(qd_scan_result),
                               hipMemcpyDeviceToHost, p->cs));
    return QD_OK;
}

// the host inflates this batch's BGZF text of one window (the device refused a block): same bytes, or the file is damaged
int host_inflate_window(qd_pipe* p, Window& w) {
    if (join_inflate(p, w) != QD_OK) return QD_ERR_HIP;
    const int fd = open(w.path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return pfail(p, QD_ERR_FORMAT, w.path + ": " + strerror(errno));
    std::vector<uint8_t> comp, text;
    int rc = QD_OK;
    for (const Window::Run& r : w.runs) {
        const int64_t file_off = r.file_off;
        const size_t bytes = r.bytes;
        const uint32_t tlen = r.text_bytes;
        if (r.at + (int64_t)tlen <= 0) continue;  // (all of it was dropped by a carry: a shared chunk's lead-in)
        comp.resize(bytes);
        size_t got = 0;
        while (got < bytes) {
            const ssize_t g = pread(fd, comp.data() + got, bytes - got, (off_t)(file_off + (int64_t)got));
            if (g <= 0) break;
            got += (size_t)g;
        }
        text.resize(tlen);
        if (got != bytes || !qdio::host_inflate_members(comp.data(), bytes, text.data(), tlen)) {
            rc = pfail(p, QD_ERR_FORMAT, w.path + ": damaged BGZF block");
            break;
        }
        // (a run whose front was carried away lands with its rest at the window's start; nothing is written behind the window's text)
        const size_t cut = r.at < 0 ? (size_t)(-r.at) : 0, dst = r.at < 0 ? 0 : (size_t)r.at;
        const size_t n = dst < w.len ? std::min<size_t>(tlen - cut, (size_t)w.len - dst) : 0;
        if (n && hipMemcpy(w.buf[w.cur].p + dst, text.data() + cut, n, hipMemcpyHostToDevice) != hipSuccess) {
            rc = pfail(p, QD_ERR_HIP, "hipMemcpy of host-inflated text failed");
            break;
        }
        ++p->st.host_inflated_runs;
    }
    close(fd);
    return rc;
}

// Text per gzip member: 1 MiB -- less when so many destinations take members in one batch that their slots (every member gets one of
// the longest member's bound) would not fit member_slots_bytes: thousands of samples make thousands of small members per batch.
uint32_t piece_bytes_for(const qd_pipe* p, uint64_t text_bytes, uint64_t n_dest) {
    uint32_t pb = PIECE_BYTES;
    while (pb > QD_LZ_SUB && (text_bytes / pb + 2 * n_dest + 2) * (uint64_t)qd_huffman_member_bound(pb) > (uint64_t)p->member_slots_bytes) pb >>= 1;
    return pb;
}

// The coder's scratch of candidates / tokens is 4 bytes per byte of text: a batch's sub-blocks go through it a quarter at a time (a
// launch of some thousand workgroups still fills the device several times over), which keeps 2.5 GB of a 2 M-pair batch's 3.3 unallocated.
uint32_t lz_slice_subs(uint32_t n_subs) { return std::max<uint32_t>(2048, (n_subs + 3) / 4); }

// The output side's buffers (two thirds of the bytes: text, member slots, packed members, the coder's scratch): allocated by a thread of
// their own while the driver fills and scans the first batch's windows -- a device allocation costs ~10 ms per GB, 0.1-0.25 s for a
// pipeline's 12 GB, and the first batch's inflate launches need none of these.  Joined before the first batch is processed.
int reserve_output(qd_pipe* p, uint32_t n_dest, double out_text) {
    if (hipSetDevice(p->device) != hipSuccess) return QD_ERR_HIP;
    const size_t T = (size_t)out_text + (size_t)n_dest * 64;
    const uint32_t piece_bytes = piece_bytes_for(p, T, n_dest);
    const size_t n_pieces = T / piece_bytes + 2 * (size_t)n_dest + 2, n_subs = T / QD_LZ_SUB + n_pieces;
    const size_t out_stride = (size_t)qd_huffman_member_bound(piece_bytes), sub_stride = (size_t)qd_huffman_member_bound(QD_LZ_SUB);
    hipError_t e = hipSuccess;
    auto need = [&](DevBuf& b, size_t n) {  // (first allocations: nothing to keep, no stream involved)
        if (e == hipSuccess) e = b.need(n);
    };
    for (OutSet& o : p->out) {
        need(o.text, T + 64);
        need(o.pieces, n_pieces * sizeof(qd_deflate_piece));
        need(o.members, n_pieces * out_stride);
        need(o.member_len, n_pieces * 4);
        need(o.member_off, (n_pieces + 1) * 8);
        need(o.packed, n_pieces * out_stride);
    }
    need(p->subs, (n_subs + 1) * sizeof(qd_lz_sub));
    need(p->ranges, (n_subs + 1) * sizeof(qd_crc_range));
    need(p->crc, (n_subs + 1) * 4);
    need(p->first_sub, (n_pieces + 1) * 4);
    need(p->tokens, (size_t)lz_slice_subs((uint32_t)std::min<size_t>(n_subs, 0xFFFFFFFFu)) * QD_LZ_SUB * 4);
    need(p->sub_out, n_subs * sub_stride);
    need(p->sub_bytes, n_subs * 4);
    return e == hipSuccess ? QD_OK : QD_ERR_HIP;
}

int join_reserve(qd_pipe* p) {
    if (!p->reserve_thread.joinable()) return QD_OK;
    {
        Tick tick(g_alloc_seconds);
        p->reserve_thread.join();
    }
    return p->reserve_rc == QD_OK ? QD_OK : pfail(p, QD_ERR_HIP, "device allocation of the output buffers failed");
}

// Every buffer at the size a full batch needs, in one go, once the first scan has told what a record of every stream weighs:
// growing them one by one as the first batches arrive drains the compute stream each time (a quarter of a 16 M-pair run).
int reserve_buffers(qd_pipe* p, uint32_t B, uint32_t n_dest) {
    const qd_layout& L = p->lay;
    double out_text = 0;
    for (int s = 0; s < 2; ++s) out_text += (double)B * (p->win[s].avg > 0 ? p->win[s].avg : 400.0) * 1.06;
    if (join_reserve(p) != QD_OK) return QD_ERR_HIP;
    p->reserve_rc = QD_OK;
    try {
        p->reserve_thread = std::thread([p, n_dest, out_text] { p->reserve_rc = reserve_output(p, n_dest, out_text); });
    } catch (const std::system_error&) {  // (no thread to be had: on this one)
        p->reserve_rc = reserve_output(p, n_dest, out_text);
        if (p->reserve_rc != QD_OK) return pfail(p, QD_ERR_HIP, "device allocation of the output buffers failed");
    }
    for (int s = 0; s < p->n_streams; ++s) {
        Window& w = p->win[s];
        const double avg = w.avg > 0 ? w.avg : (s < 2 ? 400.0 : 64.0);
        const size_t text = std::min<size_t>((size_t)((double)B * avg * 1.08) + (96u << 20), WINDOW_MAX + (64u << 20));
        for (int k = 0; k < 2; ++k) PCHK(p, w.buf[k].need(text + 2 * QD_TEXT_TILE, k == w.cur ? w.len : 0, p->cs));
        const size_t lines = (size_t)((double)text / avg * 4.4 * 1.25) + 65536;
        PCHK(p, w.tile_counts.need((text / QD_TEXT_TILE + 3) * 4, 0, p->cs));
        PCHK(p, w.tile_base.need((text / QD_TEXT_TILE + 3) * 4, 0, p->cs));
        PCHK(p, w.lines.need(lines * 4 + 64, 0, p->cs));
        PCHK(p, w.rec_tile.need((lines / 4 / 1024 + 4) * 4, 0, p->cs));
        PCHK(p, w.recs.need((lines / 4 + 1) * sizeof(qd_rec), 0, p->cs));
        const size_t blocks = text / 16384 + 1024;  // (bgzip fills its blocks: ~64 KiB of text each)
        PCHK(p, w.blk.need(blocks * sizeof(qd_inflate_block), 0, p->cs));
        PCHK(p, w.expect.need(blocks * 4, 0, p->cs));
        PCHK(p, w.status.need(blocks * 4, 0, p->cs));
        PCHK(p, w.crc.need(blocks * 4, 0, p->cs));
    }
    {  // ordinary gzip streams: the gzip kernels' buffers for a batch's steps, and the streams' compressed bytes
        uint64_t comp = 0, text = 0;
        for (int s = 0; s < p->n_streams; ++s) {
            Window& w = p->win[s];
            if (!w.gz.active) continue;
            const double avg = w.avg > 0 ? w.avg : (s < 2 ? 400.0 : 64.0);
            const uint64_t t = (uint64_t)((double)B * avg * 1.15) + (64u << 20), c = (uint64_t)((double)t / std::max(1.5, w.gz.ratio)) + 2 * SEG_BYTES;
            comp += c;
            text += t;
            for (int k = 0; k < 2; ++k) PCHK(p, w.gz.comp[k].need((size_t)c + SEG_BYTES, k == w.gz.ccur ? (size_t)w.gz.comp_len : 0, p->cs));
        }
        if (comp && p->gz) {
            Tick tick(g_alloc_seconds);  // (the gzip kernels' buffers are the largest of a batch of gzip streams)
            PCHK(p, p->gz->reserve(comp, text));
        }
    }
    if (p->inflate_form == 3) {  // a batch's blocks of all streams go down together: ~a block per 64 KiB of text
        size_t blocks3 = 0;
        for (int s = 0; s < p->n_streams; ++s) {
            const double avg = p->win[s].avg > 0 ? p->win[s].avg : (s < 2 ? 400.0 : 64.0);
            blocks3 += (size_t)((double)B * avg * 1.08 / 60000.0) + 64 + 3 * (SEG_BYTES * 5 / 60000);  // (+ the uploads a top-up overshoots by)
        }
        PCHK(p, p->jobs3.need(blocks3 * sizeof(qd_inflate3_job), 0, p->cs));
        PCHK(p, p->status3.need(blocks3 * 4, 0, p->cs));
        PCHK(p, p->scratch3.need(qd_inflate3_scratch_bytes((uint32_t)std::min<size_t>(blocks3, LAUNCH_BLOCKS3)), 0, p->cs));
    } else {
        PCHK(p, p->matches.need((size_t)LAUNCH_BLOCKS * QD_INFLATE_MATCHES_PER_BLOCK * 8, 0, p->cs));
        if (p->n_is > 1) PCHK(p, p->matches_b.need((size_t)LAUNCH_BLOCKS * QD_INFLATE_MATCHES_PER_BLOCK * 8, 0, p->cs));
    }
    const size_t n = B;
    for (int k = 0; k < L.n_streams; ++k) {
        PCHK(p, p->rows_seq[k].need(n * L.seq_stride[k] + 64, 0, p->cs));
        PCHK(p, p->rows_qual[k].need(n * L.qual_stride[k] + 64, 0, p->cs));
        PCHK(p, p->rows_len[k].need(n + 64, 0, p->cs));
    }
    PCHK(p, p->codes.need(n * 2 + 64, 0, p->cs));
    if (L.mol_width) PCHK(p, p->mol.need(n * L.mol_width + 64, 0, p->cs));
    PCHK(p, p->short_idx.need((n / 2 + 64) * 4, 0, p->cs));
    const size_t H = 256 * ((n + 1023) / 1024);
    PCHK(p, p->hist.need((H + H / 4096 + 8) * 4, 0, p->cs));
    for (DevBuf* b : {&p->len1, &p->len2, &p->tmp, &p->perm, &p->g1, &p->g2}) PCHK(p, b->need((n + 1) * 4 + 64, 0, p->cs));
    PCHK(p, p->dest.need(n * 2 + 64, 0, p->cs));
    PCHK(p, p->sdest.need(n * 2 + 64, 0, p->cs));
    PCHK(p, p->scan_tiles.need((n / 4096 + 4) * 4, 0, p->cs));
    return QD_OK;
}

// an output set the collector is done with
int take_out_set(qd_pipe* p, int b) {
    OutSet& o = p->out[b];
    Tick tick(p->st.wait_out_set);
    std::unique_lock<std::mutex> g(p->om);
    p->ocv.wait(g, [&] { return !o.busy; });
    o.busy = true;
    return b;
}

// pairs [0, n) of the four windows: rows -> codes -> sorted by destination -> formatted -> coded; members to the collector
int process_batch(qd_pipe* p, uint32_t n, qd_sink* sink, int64_t batch_index) {
    if (join_reserve(p) != QD_OK) return QD_ERR_HIP;
    const qd_layout& L = p->lay;
    const qdio::SinkInfo si = qdio::sink_info(sink);
    const uint32_t S = si.n_samples, nd = 2 * S + 1;
    const int ni = L.n_streams;
    Window* iw[2] = {&p->win[2], &p->win[3]};
    // 1. index rows
    for (int k = 0; k < ni; ++k) {
        PCHK(p, p->rows_seq[k].need((size_t)n * L.seq_stride[k] + 64, 0, p->cs));
        PCHK(p, p->rows_qual[k].need((size_t)n * L.qual_stride[k] + 64, 0, p->cs));
        PCHK(p, p->rows_len[k].need((size_t)n + 64, 0, p->cs));
    }
    PCHK(p, p->codes.need((size_t)n * 2 + 64, 0, p->cs));
    if (L.mol_width) PCHK(p, p->mol.need((size_t)n * L.mol_width + 64, 0, p->cs));
    const uint32_t short_cap = n / 2 + 64;
    PCHK(p, p->short_idx.need((size_t)short_cap * 4, 0, p->cs));
    uint32_t* d_nshort = reinterpret_cast<uint32_t*>(p->d_res.p + 4 * sizeof(qd_scan_result));
    PCHK(p, hipMemsetAsync(d_nshort, 0, 4, p->cs));
    qd_pack_args pa{};
    for (int k = 0; k < ni; ++k) {
        pa.text[k] = iw[k]->buf[iw[k]->cur].p;
        pa.recs[k] = iw[k]->recs.as<qd_rec>();
        pa.seq[k] = p->rows_seq[k].p;
        pa.qual[k] = p->rows_qual[k].p;
        pa.len[k] = p->rows_len[k].p;
    }
    pa.short_idx = p->short_idx.as<uint32_t>();
    pa.n_short = d_nshort;
    pa.short_cap = short_cap;
    PCHK(p, qd_text_pack_rows(L, n, pa, p->cs));
    // 2. codes (src/Sample.py:56-91); the counters move in the context
    qd_rows rows{};
    bool ragged = false;
    for (int k = 0; k < ni; ++k) {
        rows.seq[k] = p->rows_seq[k].p;
        rows.qual[k] = p->rows_qual[k].p;
        ragged = ragged || iw[k]->res.n_short > 0;  // (short reads somewhere in the window: maybe among these pairs)
    }
    uint8_t* d_mol = L.mol_width ? p->mol.p : nullptr;
    if (ragged) {
        uint32_t* h_nshort = reinterpret_cast<uint32_t*>(p->h_res.p + 4 * sizeof(qd_scan_result));
        PCHK(p, hipMemcpyAsync(h_nshort, d_nshort, 4, hipMemcpyDeviceToHost, p->cs));
        int rc = sync_compute(p);
        if (rc != QD_OK) return rc;
        ragged = *h_nshort > 0;
        if (ragged) {
            for (int k = 0; k < ni; ++k) rows.len[k] = p->rows_len[k].p;
            rc = qd_demux_device_ragged(p->ctx, n, &rows, p->codes.as<uint16_t>(), d_mol, *h_nshort, p->short_idx.as<uint32_t>(), p->cs);
            if (rc != QD_OK) return pfail(p, rc, std::string("demux: ") + qd_last_error(p->ctx));
        }
    }
    if (!ragged) {
        const int rc = qd_demux_device(p->ctx, n, &rows, p->codes.as<uint16_t>(), d_mol, p->cs);
        if (rc != QD_OK) return pfail(p, rc, std::string("demux: ") + qd_last_error(p->ctx));
    }
    // 3. destinations, output lengths, stable sort by destination, output offsets
    PCHK(p, p->dest.need((size_t)n * 2 + 64, 0, p->cs));
    PCHK(p, p->len1.need((size_t)n * 4 + 64, 0, p->cs));
    PCHK(p, p->len2.need((size_t)n * 4 + 64, 0, p->cs));
    const size_t H = 256 * (((size_t)n + 1023) / 1024);
    PCHK(p, p->hist.need((H + H / 4096 + 8) * 4, 0, p->cs));
    PCHK(p, p->tmp.need((size_t)n * 4 + 64, 0, p->cs));
    PCHK(p, p->perm.need((size_t)n * 4 + 64, 0, p->cs));
    PCHK(p, p->sdest.need((size_t)n * 2 + 64, 0, p->cs));
    PCHK(p, p->g1.need(((size_t)n + 1) * 4 + 64, 0, p->cs));
    PCHK(p, p->g2.need(((size_t)n + 1) * 4 + 64, 0, p->cs));
    PCHK(p, p->scan_tiles.need(((size_t)n / 4096 + 4) * 4, 0, p->cs));
    PCHK(p, p->first.need((size_t)nd * 4, 0, p->cs));
    PCHK(p, p->g1_first.need((size_t)nd * 4, 0, p->cs));
    PCHK(p, p->g2_first.need((size_t)nd * 4, 0, p->cs));
    qd_route_args ra{};
    ra.codes = p->codes.as<uint16_t>();
    ra.r1 = p->win[0].recs.as<qd_rec>();
    ra.r2 = p->win[1].recs.as<qd_rec>();
    for (int k = 0; k < ni; ++k) ra.idx[k] = iw[k]->recs.as<qd_rec>();
    ra.dest = p->dest.as<uint16_t>();
    ra.len1 = p->len1.as<uint32_t>();
    ra.len2 = p->len2.as<uint32_t>();
    PCHK(p, qd_text_dest_lens(p->plan, S, si.write_pass, si.write_fail, si.write_undet, n, ra, p->cs));
    PCHK(p, qd_text_sort_by_dest(ra.dest, n, nd, p->hist.as<uint32_t>(), p->tmp.as<uint32_t>(), p->perm.as<uint32_t>(), p->cs));
    PCHK(p, qd_text_scan_gathered(ra.len1, p->perm.as<uint32_t>(), n, p->scan_tiles.as<uint32_t>(), p->g1.as<uint32_t>(), ra.dest, p->sdest.as<uint16_t>(), p->cs));
    PCHK(p, qd_text_scan_gathered(ra.len2, p->perm.as<uint32_t>(), n, p->scan_tiles.as<uint32_t>(), p->g2.as<uint32_t>(), nullptr, nullptr, p->cs));
    PCHK(p, qd_text_dest_bounds(p->sdest.as<uint16_t>(), p->g1.as<uint32_t>(), p->g2.as<uint32_t>(), n, nd, p->first.as<uint32_t>(), p->g1_first.as<uint32_t>(),
                                p->g2_first.as<uint32_t>(), p->cs));
    // what the next batch keeps of every window
    {
        const qd_rec* recs[4] = {nullptr, nullptr, nullptr, nullptr};
        qd_scan_result* res[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int s = 0; s < p->n_streams; ++s) {
            recs[s] = p->win[s].recs.as<qd_rec>();
            res[s] = p->d_res.as<qd_scan_result>() + s;
        }
        const uint32_t taken[4] = {n, n, n, n};
        PCHK(p, qd_text_carry_info(recs, res, p->n_streams, taken, p->cs));
    }
    // 4. read back: per-destination bounds, totals, carry starts
    PCHK(p, p->h_first.need((size_t)nd * 12 + 16));
    uint32_t* h_first = reinterpret_cast<uint32_t*>(p->h_first.p);
    uint32_t* h_g1f = h_first + nd;
    uint32_t* h_g2f = h_g1f + nd;
    uint32_t* h_tot = h_g2f + nd;  // G1[n], G2[n]
    PCHK(p, hipMemcpyAsync(h_first, p->first.p, (size_t)nd * 4, hipMemcpyDeviceToHost, p->cs));
    PCHK(p, hipMemcpyAsync(h_g1f, p->g1_first.p, (size_t)nd * 4, hipMemcpyDeviceToHost, p->cs));
    PCHK(p, hipMemcpyAsync(h_g2f, p->g2_first.p, (size_t)nd * 4, hipMemcpyDeviceToHost, p->cs));
    PCHK(p, hipMemcpyAsync(h_tot, p->g1.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, p->cs));
    PCHK(p, hipMemcpyAsync(h_tot + 1, p->g2.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, p->cs));
    PCHK(p, hipMemcpyAsync(p->h_res.p, p->d_res.p, (size_t)p->n_streams * sizeof(qd_scan_result), hipMemcpyDeviceToHost, p->cs));
    int rc = sync_compute(p);
    if (rc != QD_OK) return rc;
    for (int s = 0; s < p->n_streams; ++s) p->win[s].res.carry_start = reinterpret_cast<qd_scan_result*>(p->h_res.p)[s].carry_start;
    // 5. layout of the output text: every destination's R1 region, then every R2 region, 16-byte aligned; pieces of 1 MiB
    std::vector<int64_t> base1(nd, 0), base2(nd, 0);
    const uint32_t piece_bytes = piece_bytes_for(p, (uint64_t)h_tot[0] + h_tot[1], nd);
    BatchOut bo;
    bo.sink = sink;
    bo.level = si.level;
    std::vector<qd_lz_sub> subs;
    std::vector<uint32_t> first_sub;
    std::vector<qd_crc_range> ranges;
    uint64_t at = 0;
    {
        // a destination without pairs starts where the next one does
        std::vector<uint32_t> g1s(nd + 1), g2s(nd + 1);
        g1s[nd] = h_tot[0];
        g2s[nd] = h_tot[1];
        for (int64_t d = (int64_t)nd - 1; d >= 0; --d) {
            const bool none = h_first[d] == 0xFFFFFFFFu;
            g1s[d] = none ? g1s[d + 1] : h_g1f[d];
            g2s[d] = none ? g2s[d + 1] : h_g2f[d];
        }
        for (int k = 0; k < 2; ++k)
            for (uint32_t d = 0; d < nd; ++d) {
                const std::vector<uint32_t>& gs = k ? g2s : g1s;
                const uint64_t bytes = gs[d + 1] - gs[d];
                (k ? base2 : base1)[d] = (int64_t)at - (int64_t)gs[d];
                if (!bytes) continue;
                FileRun fr;
                fr.code = d == 2 * S ? QD_CODE_UNDETERMINED : d;
                fr.k = k;
                fr.first = (uint32_t)bo.pieces.size();
                fr.text_bytes = bytes;
                for (uint64_t a = 0; a < bytes; a += piece_bytes) {
                    const uint32_t plen = (uint32_t)std::min<uint64_t>(piece_bytes, bytes - a);
                    first_sub.push_back((uint32_t)subs.size());
                    for (uint32_t q = 0; q < plen; q += QD_LZ_SUB) {
                        const uint32_t slen = std::min<uint32_t>(QD_LZ_SUB, plen - q);
                        subs.push_back(qd_lz_sub{at + a + q, slen, (uint32_t)bo.pieces.size()});
                        ranges.push_back(qd_crc_range{at + a + q, slen, 0});
                    }
                    bo.pieces.push_back(qd_deflate_piece{at + a, plen, 0});
                }
                fr.n = (uint32_t)bo.pieces.size() - fr.first;
                bo.files.push_back(fr);
                at = (at + bytes + 15) & ~(uint64_t)15;
            }
        first_sub.push_back((uint32_t)subs.size());
    }
    const uint32_t n_pieces = (uint32_t)bo.pieces.size(), n_subs = (uint32_t)subs.size();
    p->st.pairs += n;
    ++p->st.batches;
    p->st.text_out_bytes += (int64_t)h_tot[0] + h_tot[1];
    p->st.pieces += n_pieces;
    if (!n_pieces) return QD_OK;  // every destination's write flag is off
    // 6. format, CRC-32, code, pack
    const int set = take_out_set(p, (int)(batch_index & 1));
    OutSet& o = p->out[set];
    bo.set = set;
    bo.n_pieces = n_pieces;
    const int64_t out_stride = qd_huffman_member_bound(piece_bytes);
    const int64_t sub_stride = qd_huffman_member_bound(QD_LZ_SUB);
    PCHK(p, o.text.need((size_t)at + 64, 0, p->cs));
    PCHK(p, p->base1.need((size_t)nd * 8, 0, p->cs));
    PCHK(p, p->base2.need((size_t)nd * 8, 0, p->cs));
    PCHK(p, o.pieces.need((size_t)n_pieces * sizeof(qd_deflate_piece), 0, p->cs));
    PCHK(p, p->subs.need((size_t)(n_subs + 1) * sizeof(qd_lz_sub), 0, p->cs));
    PCHK(p, p->first_sub.need((size_t)(n_pieces + 1) * 4, 0, p->cs));
    PCHK(p, p->ranges.need((size_t)(n_subs + 1) * sizeof(qd_crc_range), 0, p->cs));
    PCHK(p, p->crc.need((size_t)(n_subs + 1) * 4, 0, p->cs));
    PCHK(p, o.members.need((size_t)n_pieces * (size_t)out_stride, 0, p->cs));
    PCHK(p, o.member_len.need((size_t)n_pieces * 4, 0, p->cs));
    PCHK(p, o.member_off.need((size_t)(n_pieces + 1) * 8, 0, p->cs));
    PCHK(p, o.packed.need((size_t)n_pieces * (size_t)out_stride, 0, p->cs));
    // (the tail of the batch before still reads the sub-block tables and the coder's scratch: the uploads below wait for it)
    const hipStream_t ts = p->coder_stream ? p->es : p->cs;  // the tail's stream
    if (p->coded_pending && ts != p->cs) PCHK(p, hipStreamWaitEvent(p->cs, p->coded, 0));
    p->coded_pending = false;
    PCHK(p, p->stage.upload(p->base1.p, base1.data(), (size_t)nd * 8, p->cs));
    PCHK(p, p->stage.upload(p->base2.p, base2.data(), (size_t)nd * 8, p->cs));
    PCHK(p, p->stage.upload(o.pieces.p, bo.pieces.data(), (size_t)n_pieces * sizeof(qd_deflate_piece), p->cs));
    PCHK(p, p->stage.upload(p->subs.p, subs.data(), (size_t)n_subs * sizeof(qd_lz_sub), p->cs));
    PCHK(p, p->stage.upload(p->first_sub.p, first_sub.data(), (size_t)(n_pieces + 1) * 4, p->cs));
    PCHK(p, p->stage.upload(p->ranges.p, ranges.data(), (size_t)n_subs * sizeof(qd_crc_range), p->cs));
    qd_format_args fa{};
    fa.perm = p->perm.as<uint32_t>();
    fa.sdest = p->sdest.as<uint16_t>();
    fa.g1 = p->g1.as<uint32_t>();
    fa.g2 = p->g2.as<uint32_t>();
    fa.base1 = p->base1.as<int64_t>();
    fa.base2 = p->base2.as<int64_t>();
    fa.text1 = p->win[0].buf[p->win[0].cur].p;
    fa.text2 = p->win[1].buf[p->win[1].cur].p;
    fa.r1 = ra.r1;
    fa.r2 = ra.r2;
    for (int k = 0; k < ni; ++k) {
        fa.itext[k] = iw[k]->buf[iw[k]->cur].p;
        fa.idx[k] = ra.idx[k];
    }
    fa.out1 = o.text.p;
    fa.out2 = o.text.p;
    PCHK(p, qd_text_format(p->plan, S, si.write_pass, si.write_fail, si.write_undet, n, fa, p->cs));
    if (ts != p->cs) {  // the tail starts when the text is formatted (and the tables above are up)
        PCHK(p, hipEventRecord(p->formatted, p->cs));
        PCHK(p, hipStreamWaitEvent(ts, p->formatted, 0));
    }
    static_assert(sizeof(qd_deflate_piece) == 16 && offsetof(qd_deflate_piece, crc32) == 12, "the combined CRCs land in the piece table");
    uint32_t* piece_crc = reinterpret_cast<uint32_t*>(o.pieces.p) + 3;
    if (si.level == 1) {
        const uint32_t slice = lz_slice_subs(n_subs);
        PCHK(p, p->tokens.need((size_t)slice * QD_LZ_SUB * 4, 0, p->cs));
        PCHK(p, p->sub_out.need((size_t)n_subs * (size_t)sub_stride, 0, p->cs));
        PCHK(p, p->sub_bytes.need((size_t)n_subs * 4, 0, p->cs));
        // the sub-blocks' CRC-32s come out of the coder (taken while a sub-block's text is staged), the pieces' are combined from them
        for (uint32_t at = 0; at < n_subs; at += slice)
            PCHK(p, qd_launch_lz_subblocks(o.text.p, p->subs.as<qd_lz_sub>() + at, std::min(slice, n_subs - at), p->tokens.as<uint32_t>(), p->sub_out.p + (size_t)at * sub_stride,
                                           sub_stride, p->sub_bytes.as<uint32_t>() + at, p->crc.as<uint32_t>() + at, ts));
        PCHK(p, qd_text_crc32_combine(p->ranges.as<qd_crc_range>(), p->crc.as<uint32_t>(), p->first_sub.as<uint32_t>(), n_pieces, piece_crc, 4, ts));
        PCHK(p, qd_launch_lz_members(o.pieces.as<qd_deflate_piece>(), n_pieces, p->subs.as<qd_lz_sub>(), p->first_sub.as<uint32_t>(), n_subs, p->sub_out.p, sub_stride,
                                     p->sub_bytes.as<uint32_t>(), o.members.p, out_stride, o.member_len.as<uint32_t>(), ts));
    } else {
        PCHK(p, qd_text_crc32(o.text.p, p->ranges.as<qd_crc_range>(), n_subs, p->crc.as<uint32_t>(), ts));
        PCHK(p, qd_text_crc32_combine(p->ranges.as<qd_crc_range>(), p->crc.as<uint32_t>(), p->first_sub.as<uint32_t>(), n_pieces, piece_crc, 4, ts));
        PCHK(p, qd_launch_huffman(o.text.p, o.pieces.as<qd_deflate_piece>(), n_pieces, o.members.p, out_stride, o.member_len.as<uint32_t>(), ts));
    }
    PCHK(p, qd_text_pack_members(o.members.p, out_stride, o.member_len.as<uint32_t>(), n_pieces, o.member_off.as<uint64_t>(), o.packed.p, ts));
    PCHK(p, hipEventCreateWithFlags(&bo.done, hipEventDisableTiming));
    PCHK(p, hipEventRecord(bo.done, ts));
    if (ts != p->cs) {
        PCHK(p, hipEventRecord(p->coded, ts));
        p->coded_pending = true;
    }
    to_collector(p, std::move(bo));
    return QD_OK;
}

// a D2D move of a window's text from `from` on to the front of its other buffer
int carry_window(qd_pipe* p, Window& w, uint32_t from_in) {
    if (join_inflate(p, w) != QD_OK) return QD_ERR_HIP;
    const uint32_t from = std::min(from_in, w.len), left = w.len - from;
    const int nx = w.cur ^ 1;
    PCHK(p, w.buf[nx].need((size_t)left + 2 * QD_TEXT_TILE, 0, p->cs));
    if (left) PCHK(p, hipMemcpyAsync(w.buf[nx].p, w.buf[w.cur].p + from, left, hipMemcpyDeviceToDevice, p->cs));
    w.cur = nx;
    w.len = left;
    w.dirty = true;
    // the text of the runs not verified yet moved with it (a shared chunk drops its lead-in before the first scan: a block the
    // device then refuses must be inflated by the host to where its text lies NOW)
    for (Window::Run& r : w.runs) r.at -= (int64_t)from;
    return QD_OK;
}

// Bytes per fastq record at the head of a file (gzip of any framing, BGZF, or plain text): the first ~256 KB of its text, inflated by
// zlib on the host.  0: could not tell (unreadable, damaged, fewer than two records) -- the run then learns the size from its first
// small window, as it always did.  What the answer is used for: sizing the buffers and the first top-up of a run (r05: a run's first
// batch used to cost two inflate launches -- a small one to learn the record size, then the batch's own -- and a token launch takes
// ~20 ms whatever it holds).  An estimate only: a window that turns out short is topped up, a buffer that turns out small grows.
struct Peek {
    double record_bytes = 0;   // 0: could not tell
    bool plain_gzip = false;   // gzip, and not BGZF at its start: what the feeder uploads as it is when the device inflates such files
    double ratio = 0;          // text made per compressed byte read (plain gzip)
};
Peek peek_record_bytes(const char* path, int64_t start) {
    Peek pk;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return pk;
    std::vector<uint8_t> in(192 << 10), out(384 << 10);
    const ssize_t got = pread(fd, in.data(), in.size(), (off_t)std::max<int64_t>(start, 0));
    close(fd);
    if (got < 32) return pk;
    const uint8_t* text = in.data();
    size_t n_text = (size_t)got;
    if (in[0] == 0x1f && in[1] == 0x8b) {
        pk.plain_gzip = in[2] == 8 && !qdio::bgzf_block_size(in.data(), (size_t)got);  // (the feeder's test: Feeder::one_file)
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 32) != Z_OK) return pk;
        z.next_in = in.data();
        z.avail_in = (uInt)got;
        z.next_out = out.data();
        z.avail_out = (uInt)out.size();
        for (;;) {
            const int rc = inflate(&z, Z_NO_FLUSH);
            if (rc == Z_STREAM_END && z.avail_in > 18 && z.avail_out > 0) {  // the next member (BGZF: one per 64 KiB)
                if (inflateReset(&z) != Z_OK) break;
                continue;
            }
            break;  // (out of input or of room, the end, or an error: what has been made so far is text either way)
        }
        n_text = out.size() - z.avail_out;
        const size_t used = (size_t)got - z.avail_in;
        if (used) pk.ratio = (double)n_text / (double)used;
        inflateEnd(&z);
        text = out.data();
    }
    size_t lines = 0, last4 = 0;
    for (size_t i = 0; i < n_text; ++i)
        if (text[i] == '\n' && (++lines & 3) == 0) last4 = i + 1;
    if (lines >= 8) pk.record_bytes = (double)last4 / (double)(lines / 4);
    return pk;
}

int run_chunk(qd_pipe* p, std::vector<std::unique_ptr<Feeder>>& feeders, int chunk, const qd_pipe_chunk& spec, int64_t* batch_index) {
    qd_sink* sink = spec.sink;
    const int ns = p->n_streams;
    const uint32_t B = (uint32_t)std::min<int64_t>(p->batch_pairs, 0x7FFFFFFF);
    // a chunk that several ranks share: this rank's part starts skip_bytes into the text behind start_offset, skip_kept kept
    // records further on, and is max_pairs pairs long
    uint64_t skip_bytes[4], skip_kept[4];
    for (int s = 0; s < 4; ++s) {
        skip_bytes[s] = spec.skip_bytes[s] > 0 ? (uint64_t)spec.skip_bytes[s] : 0;
        skip_kept[s] = spec.skip_kept[s] > 0 ? (uint64_t)spec.skip_kept[s] : 0;
    }
    uint64_t pairs_left = spec.max_pairs > 0 ? (uint64_t)spec.max_pairs : ~0ull;
    for (int s = 0; s < ns; ++s) {
        Window& w = p->win[s];
        if (join_inflate(p, w) != QD_OK) return QD_ERR_HIP;
        w.len = 0;
        w.eof = false;
        w.dirty = true;
        w.carry_kept = 0;
        w.n_blocks = 0;
        w.runs.clear();
        w.pending.clear();
        w.pending_text = 0;
        gz_close(w);
    }
    if (!p->reserved && p->peek_records && !spec.skip_bytes[0] && !spec.skip_kept[0]) {
        // the run's first chunk: what a record of every stream weighs, from the head of its file -- buffers and the first top-up
        // are sized before anything is inflated on the device
        double avg[4] = {0, 0, 0, 0};
        Peek pk[4];
        bool all = true;
        for (int s = 0; s < ns; ++s) {
            pk[s] = peek_record_bytes(p->win[s].path.c_str(), spec.start_offset[s]);
            avg[s] = pk[s].record_bytes;
            all = all && avg[s] >= 16.0;
        }
        if (all) {
            for (int s = 0; s < ns; ++s) {
                p->win[s].avg = avg[s];
                if (pk[s].plain_gzip && p->device_gunzip) {  // the gzip kernels' buffers are sized with the rest (reserve_buffers)
                    p->win[s].gz.ratio = std::min(64.0, std::max(1.0, pk[s].ratio));
                    const int rc = gz_activate(p, p->win[s]);
                    if (rc != QD_OK) return rc;
                }
            }
            p->reserved = true;
            const uint32_t most = (uint32_t)std::min<double>((double)B, (double)p->max_r1_bytes * (p->r1_compressed ? 8.0 : 1.0) / avg[0] + 1024.0);
            const int rc = reserve_buffers(p, most, 2 * qdio::sink_info(sink).n_samples + 1);
            if (rc != QD_OK) return rc;
        }
    }
    std::vector<size_t> want(ns, 1);  // first round: one upload, to learn the stream's bytes per record ...
    if (p->reserved)                  // ... unless the chunks before (or a look at the files' heads) have told (a small first top-up is an inflate launch of its own)
        for (int s = 0; s < ns; ++s)
            if (p->win[s].avg > 0) want[s] = std::min<size_t>((size_t)((double)B * p->win[s].avg * 1.03) + 4096, WINDOW_MAX * 3 / 4);
    static const bool trace = getenv("QUADE_PIPE_TRACE") != nullptr;  // (debugging: one line per turn of the loop)
    for (uint64_t turn = 0;; ++turn) {
        if (trace && (turn < 60 || turn % 1000000 == 0)) {
            fprintf(stderr, "[pipe] chunk %d turn %llu:", chunk, (unsigned long long)turn);
            for (int s = 0; s < ns; ++s) {
                const Window& w = p->win[s];
                fprintf(stderr, " [%d len %u want %zu kept %u eof %d%s", s, w.len, want[s], w.res.n_kept, (int)w.eof, w.gz.active ? " gz" : "");
                if (w.gz.active)
                    fprintf(stderr, " comp %llu+%llu bit %llu hdr %d done %d host %d", (unsigned long long)w.gz.comp_off, (unsigned long long)w.gz.comp_len,
                            (unsigned long long)w.gz.bit, (int)w.gz.need_header, (int)w.gz.file_done, w.gz.host ? 1 : 0);
                fprintf(stderr, "]");
            }
            fprintf(stderr, "\n");
        }
        // (QUADE_PIPE_TRACE: where the host is, in ms since the first mark)
        auto mark = [&](const char* what) {
            if (!trace) return;
            static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            fprintf(stderr, "[pipe] %9.3f ms  chunk %d turn %llu: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), chunk,
                    (unsigned long long)turn, what);
        };
        // 1. text
        mark("top-up");
        for (int s = 0; s < ns; ++s) {
            const int rc = top_up(p, *feeders[s], p->win[s], s, chunk, want[s]);
            if (rc != QD_OK) return rc;
        }
        mark("inflate launches");
        {
            int rc = flush_inflate3(p);  // (the third inflater: the blocks of every stream's uploads in one launch)
            if (rc == QD_OK) rc = gz_steps(p, feeders, chunk);  // ... and a step of every ordinary gzip stream
            if (rc != QD_OK) return rc;
        }
        mark("inflate launched / gzip step done");
        // (a shared chunk: the text in front of this rank's first record goes before anything is scanned)
        {
            bool more = false;
            for (int s = 0; s < ns; ++s) {
                Window& w = p->win[s];
                if (!skip_bytes[s]) continue;
                if ((uint64_t)w.len < skip_bytes[s] && !w.eof) {
                    want[s] = (size_t)skip_bytes[s] + 1;
                    more = true;
                    continue;
                }
                const int rc = carry_window(p, w, (uint32_t)std::min<uint64_t>(skip_bytes[s], w.len));
                if (rc != QD_OK) return rc;
                skip_bytes[s] = 0;
            }
            if (more) continue;
        }
        // 2. records of the windows that changed
        bool scanned = false;
        {
            const int rc = scan_windows(p, &scanned);
            if (rc != QD_OK) return rc;
        }
        if (scanned) {
            mark("scans queued");
            int rc = sync_compute(p);
            if (rc != QD_OK) return rc;
            mark("scans done");
            bool again = false;
            for (int s = 0; s < ns; ++s) {
                Window& w = p->win[s];
                if (!w.dirty) continue;
                w.res = reinterpret_cast<qd_scan_result*>(p->h_res.p)[s];
                bool redo = false;
                const bool refused = w.n_blocks && (w.res.first_bad != 0xFFFFFFFFu || *batch_index == p->test_fail_inflate_batch);
                if (refused) {  // the device did not inflate (or check) a block: the host does this batch's blocks of the stream
                    rc = host_inflate_window(p, w);
                    if (rc != QD_OK) return rc;
                    redo = true;
                }
                if (w.n_blocks) PCHK(p, hipMemsetAsync(&p->d_res.as<qd_scan_result>()[s].first_bad, 0xFF, 4, p->cs));
                w.n_blocks = 0;
                w.runs.clear();
                if (w.res.overflow) {  // more lines than the table held: it is known how many now
                    w.line_cap = (w.res.n_lines + w.res.n_lines / 8 + 4096 + 3) & ~3u;
                    redo = true;
                }
                if (redo) {
                    ++p->st.rescans;
                    again = true;
                    continue;
                }
                w.dirty = false;
                if (w.res.n_records) w.avg = std::max(16.0, (double)w.res.tail_start / (double)w.res.n_records);
            }
            if (again) continue;
        }
        if (!p->reserved) {  // every stream's record size is known now
            p->reserved = true;
            // (not for more pairs than the run's largest seq_R1 file can hold: gzip at its best makes 1 byte of 8)
            const double avg1 = p->win[0].avg > 0 ? p->win[0].avg : 400.0;
            const uint32_t most = (uint32_t)std::min<double>((double)B, (double)p->max_r1_bytes * (p->r1_compressed ? 8.0 : 1.0) / avg1 + 1024.0);
            const int rc = reserve_buffers(p, most, 2 * qdio::sink_info(sink).n_samples + 1);
            if (rc != QD_OK) return rc;
            for (int s = 0; s < ns; ++s) p->win[s].dirty = true;  // (the scans' tables moved: once more, over these first small windows)
            continue;
        }
        // (a shared chunk: the kept records in front of this rank's first pair are dropped, window by window)
        {
            uint32_t drop[4] = {0, 0, 0, 0};
            bool any = false;
            for (int s = 0; s < ns; ++s) {
                drop[s] = (uint32_t)std::min<uint64_t>(skip_kept[s], p->win[s].res.n_kept);
                any = any || skip_kept[s] > 0;
            }
            if (any) {
                const qd_rec* recs[4] = {nullptr, nullptr, nullptr, nullptr};
                qd_scan_result* res[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int s = 0; s < ns; ++s) {
                    recs[s] = p->win[s].recs.as<qd_rec>();
                    res[s] = p->d_res.as<qd_scan_result>() + s;
                }
                PCHK(p, qd_text_carry_info(recs, res, ns, drop, p->cs));
                PCHK(p, hipMemcpyAsync(p->h_res.p, p->d_res.p, (size_t)ns * sizeof(qd_scan_result), hipMemcpyDeviceToHost, p->cs));
                int rc = sync_compute(p);
                if (rc != QD_OK) return rc;
                for (int s = 0; s < ns; ++s) {
                    Window& w = p->win[s];
                    skip_kept[s] -= drop[s];
                    if (skip_kept[s] && w.eof && drop[s] == w.res.n_kept) skip_kept[s] = 0;  // (the stream ends before this rank's part: nothing to do)
                    rc = carry_window(p, w, reinterpret_cast<qd_scan_result*>(p->h_res.p)[s].carry_start);
                    if (rc != QD_OK) return rc;
                    const double per = w.avg > 0 ? w.avg : 256.0;
                    want[s] = std::min<size_t>((size_t)w.len + (size_t)((double)(skip_kept[s] + B) * per * 1.03) + 4096, p->window_max * 3 / 4);
                }
                continue;
            }
        }
        // 3. a stream short of records that has more input: top it up (its window is scanned again)
        bool short_of = false;
        for (int s = 0; s < ns; ++s) {
            Window& w = p->win[s];
            if (w.res.n_kept >= B || w.eof) continue;
            if (w.res.n_kept > 0 && (size_t)w.len > WINDOW_MAX / 2) continue;  // a full window: a smaller batch rather than more text
            // (... and a batch that is most of B rather than another round: a top-up is an inflate launch of its own, which takes as
            //  long for a few hundred blocks as for ten thousand -- a lane decodes its block's symbols one after the other)
            if (p->inflate_form == 3 && (uint64_t)w.res.n_kept * 10 >= (uint64_t)B * 8) continue;
            const double per = w.avg > 0 ? w.avg : 256.0;
            const size_t more = (size_t)((double)(B - w.res.n_kept) * per * 1.03) + 4096;
            want[s] = std::min<size_t>(std::max<size_t>((size_t)w.len + more, (size_t)w.len + 1), WINDOW_MAX * 3 / 4);
            if (want[s] <= (size_t)w.len) want[s] = (size_t)w.len + 1;
            short_of = true;
        }
        if (short_of) continue;
        // 4. lock step: pair j = kept record j of every stream (src/Quade.py:210-221)
        uint32_t n = (uint32_t)std::min<uint64_t>(B, pairs_left);
        for (int s = 0; s < ns; ++s) n = std::min(n, p->win[s].res.n_kept);
        if (n) {
            mark("batch");
            const int rc = process_batch(p, n, sink, *batch_index);
            if (rc != QD_OK) return rc;
            mark("batch queued");
            ++*batch_index;
            pairs_left -= n;
        }
        // 5. the chunk ends with its first exhausted stream (src/Quade.py:223-224) -- or with this rank's part of it
        bool done = pairs_left == 0;
        for (int s = 0; s < ns; ++s) done = done || (p->win[s].eof && p->win[s].res.n_kept == n);
        if (done || n == 0) break;
        // 6. what is left of every window moves to the front of its other buffer
        for (int s = 0; s < ns; ++s) {
            Window& w = p->win[s];
            const int rc = carry_window(p, w, w.res.carry_start);
            if (rc != QD_OK) return rc;
            w.carry_kept = w.res.n_kept - n;
            const uint32_t left = w.len;
            const double per = w.avg > 0 ? w.avg : 256.0;
            want[s] = std::min<size_t>((size_t)left + (size_t)((double)(B > w.carry_kept ? B - w.carry_kept : 0) * per * 1.03) + 4096, WINDOW_MAX * 3 / 4);
        }
    }
    // the streams the chunk did not exhaust: their feeders stop reading this chunk, what they had read is dropped
    for (int s = 0; s < ns; ++s) feeders[s]->skip_below(chunk + 1);
    for (int s = 0; s < ns; ++s) {
        const int rc = drain_chunk(p, *feeders[s], p->win[s], chunk);
        if (rc != QD_OK) return rc;
    }
    return QD_OK;
}

// ---- a chunk that several ranks share: the index pass -----------------------------------------------------------------------------------
// file offsets of every BGZF block of a file (and its size behind the last): headers only, 18 bytes per block
int walk_bgzf(qd_pipe* p, const char* path, std::vector<int64_t>* off) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return pfail(p, QD_ERR_FORMAT, std::string(path) + ": " + strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        close(fd);
        return pfail(p, QD_ERR_FORMAT, std::string(path) + ": " + strerror(errno));
    }
    int64_t at = 0;
    uint8_t h[64];
    int rc = QD_OK;
    while (at < (int64_t)sb.st_size) {
        const ssize_t g = pread(fd, h, sizeof h, (off_t)at);
        const size_t bs = g > 0 ? qdio::bgzf_block_size(h, (size_t)g) : 0;
        if (bs < 26 || at + (int64_t)bs > (int64_t)sb.st_size) {
            rc = pfail(p, QD_ERR_UNSUPPORTED, std::string(path) + ": not BGZF blocks throughout (a shared chunk needs bgzip files)");
            break;
        }
        off->push_back(at);
        at += (int64_t)bs;
    }
    close(fd);
    if (rc == QD_OK) off->push_back(at);
    return rc;
}

int index_stream(qd_pipe* p, const char* path, int32_t world, int32_t rank, int32_t grains_per_rank, int32_t want_overlap_blocks, qd_grain_info* out,
                 int32_t cap, int32_t* n_out) {
    std::vector<int64_t> boff;
    int rc = walk_bgzf(p, path, &boff);
    if (rc != QD_OK) return rc;
    const int64_t nb = (int64_t)boff.size() - 1;
    const int64_t G = std::max<int64_t>(1, std::min<int64_t>((int64_t)world * grains_per_rank, nb));
    auto grain_block = [&](int64_t g) { return g * nb / G; };  // first block of grain g (g == G: nb)
    const int64_t g_lo = std::min<int64_t>(G, (int64_t)rank * grains_per_rank), g_hi = std::min<int64_t>(G, g_lo + grains_per_rank);
    *n_out = (int32_t)(g_hi - g_lo);
    if (g_hi <= g_lo) return QD_OK;
    if (g_hi - g_lo > cap) return pfail(p, QD_ERR_INVALID, "grain table too small");
    // one block of lead-in (is the byte in front of the first grain a newline?) and a few of overlap (the record that starts in the last grain)
    const int64_t b_first = grain_block(g_lo), b_end = grain_block(g_hi);
    const int64_t b_from = std::max<int64_t>(0, b_first - 1), b_to = std::min<int64_t>(nb, b_end + want_overlap_blocks);
    Window& w = p->win[0];
    w.path = path;
    w.len = 0;
    w.eof = false;
    w.dirty = true;
    w.n_blocks = 0;
    w.runs.clear();
    w.avg = 0;
    w.line_cap = 0;
    struct WindowMax {  // the index pass holds a rank's whole share of a stream in one window
        qd_pipe* p;
        size_t saved;
        ~WindowMax() { p->window_max = saved; }
    } restore{p, p->window_max};
    p->window_max = (size_t)3500 << 20;
    std::vector<FileSpec> files(1);
    files[0].path = path;
    files[0].start = boff[b_from];
    files[0].end = boff[b_to];
    // (start == 0 would read "the whole file" semantics with end set: fine, the range form only needs end)
    Feeder f(p->device, files);
    hipError_t he = f.start();
    if (he != hipSuccess) return pfail(p, QD_ERR_HIP, std::string("feeder: ") + hipGetErrorString(he));
    // the text offset of every block of the range, as its segments arrive
    std::vector<uint32_t> text_at;  // per block b_from + k
    uint64_t text_total = 0;
    for (;;) {
        Segment sg = f.pop();
        if (sg.kind == SEG_ERROR) {
            rc = pfail(p, QD_ERR_FORMAT, sg.err);
            break;
        }
        if (sg.kind == SEG_END) break;
        if (sg.kind != SEG_BGZF) {
            rc = pfail(p, QD_ERR_UNSUPPORTED, std::string(path) + ": not BGZF");
            break;
        }
        for (const qd_inflate_block& b : sg.blocks) {
            text_at.push_back((uint32_t)text_total);
            text_total += b.out_len;
        }
        if (text_total > p->window_max) {
            rc = pfail(p, QD_ERR_UNSUPPORTED, std::string(path) + ": this rank's share of the chunk exceeds one window (more ranks, or smaller chunks)");
            PCHK(p, f.consumed(sg.slot, p->cs));
            break;
        }
        w.pending_text += (uint32_t)sg.text_bytes;
        w.pending.push_back(std::move(sg));
        if (w.pending.size() >= GROUP_SEGMENTS) {
            rc = launch_inflate(p, f, w, 0);
            if (rc != QD_OK) break;
        }
    }
    if (rc == QD_OK) rc = launch_inflate(p, f, w, 0);
    if (rc == QD_OK) rc = flush_inflate3(p);
    if (rc != QD_OK) {  // (what was queued refers to this feeder's ring)
        p->q3_jobs.clear();
        p->q3_parts.clear();
    }
    if (rc == QD_OK && (int64_t)text_at.size() != b_to - b_from) rc = pfail(p, QD_ERR_FORMAT, std::string(path) + ": block walk and reader disagree");
    const bool at_eof = b_to == nb;
    w.eof = at_eof;
    // lines of the window (the line table may have to grow once), then the grains
    for (int attempt = 0; rc == QD_OK && attempt < 3; ++attempt) {
        rc = scan_window(p, w, 0);
        if (rc == QD_OK) rc = sync_compute(p);
        if (rc != QD_OK) break;
        w.res = reinterpret_cast<qd_scan_result*>(p->h_res.p)[0];
        if (w.n_blocks && w.res.first_bad != 0xFFFFFFFFu) {
            rc = host_inflate_window(p, w);
            if (rc != QD_OK) break;
            PCHK(p, hipMemsetAsync(&p->d_res.as<qd_scan_result>()[0].first_bad, 0xFF, 4, p->cs));
            w.n_blocks = 0;
            continue;
        }
        if (!w.res.overflow) break;
        w.line_cap = (w.res.n_lines + w.res.n_lines / 8 + 4096 + 3) & ~3u;
    }
    if (rc == QD_OK && w.res.overflow) rc = pfail(p, QD_ERR_HIP, "line table overflow");
    if (w.n_blocks) PCHK(p, hipMemsetAsync(&p->d_res.as<qd_scan_result>()[0].first_bad, 0xFF, 4, p->cs));
    w.n_blocks = 0;
    w.runs.clear();
    const int ng = (int)(g_hi - g_lo);
    std::vector<uint32_t> gstart((size_t)ng + 1);
    std::vector<qd_grain_index> gi((size_t)ng);
    if (rc == QD_OK) {
        for (int k = 0; k <= ng; ++k) {
            const int64_t b = grain_block(g_lo + k);
            gstart[(size_t)k] = b - b_from < (int64_t)text_at.size() ? text_at[(size_t)(b - b_from)] : (uint32_t)text_total;
        }
        if (g_hi == G && at_eof) gstart[(size_t)ng] = (uint32_t)text_total + 1;  // (the unterminated last line of the file ends "at" text_total)
        DevBuf d_g, d_out;
        PCHK(p, d_g.need(((size_t)ng + 1) * 4, 0, p->cs));
        PCHK(p, d_out.need((size_t)ng * sizeof(qd_grain_index), 0, p->cs));
        PCHK(p, p->stage.upload(d_g.p, gstart.data(), ((size_t)ng + 1) * 4, p->cs));
        PCHK(p, qd_text_grain_index(w.buf[w.cur].p, w.lines.as<uint32_t>(), w.line_cap, p->d_res.as<qd_scan_result>(), d_g.as<uint32_t>(), (uint32_t)ng,
                                    at_eof ? 1 : 0, d_out.as<qd_grain_index>(), p->cs));
        PCHK(p, hipStreamSynchronize(p->cs));
        PCHK(p, hipMemcpy(gi.data(), d_out.p, (size_t)ng * sizeof(qd_grain_index), hipMemcpyDeviceToHost));
        d_g.release();
        d_out.release();
        for (int k = 0; k < ng; ++k) {
            qd_grain_info& o = out[k];
            o.file_offset = boff[(size_t)grain_block(g_lo + k)];
            o.n_lines = gi[(size_t)k].n_lines;
            for (int q = 0; q < 4; ++q) {
                o.kept[q] = gi[(size_t)k].kept[q];
                o.skip_bytes[q] = gi[(size_t)k].first_head[q] == 0xFFFFFFFFu ? 0xFFFFFFFFu : gi[(size_t)k].first_head[q] - gstart[(size_t)k];
                o.incomplete[q] = gi[(size_t)k].incomplete[q];
            }
        }
    }
    f.stop();
    w.pending.clear();  // (a failed pass may leave uploads of this feeder's ring queued: the ring is gone with the feeder)
    w.pending_text = 0;
    w.len = 0;
    w.eof = false;
    w.line_cap = 0;
    w.avg = 0;
    return rc;
}

}  // namespace

extern "C" {

int qd_pipe_index(qd_pipe* p, const char* path, int32_t world, int32_t rank, int32_t grains_per_rank, qd_grain_info* out, int32_t cap, int32_t* n_out) {
    if (!p || !path || world < 1 || rank < 0 || rank >= world || grains_per_rank < 1 || !out || cap < 1 || !n_out) return pfail(p, QD_ERR_INVALID, "bad arguments");
    p->err.clear();
    PCHK(p, hipSetDevice(p->device));
    return index_stream(p, path, world, rank, grains_per_rank, 4, out, cap, n_out);
}

const char* qd_pipe_last_error(const qd_pipe* p) { return p ? p->err.c_str() : g_pipe_error.c_str(); }

int qd_pipe_create(qd_ctx* ctx, qd_pipe** out) {
    if (!ctx || !out) return pfail(nullptr, QD_ERR_INVALID, "bad arguments");
    *out = nullptr;
    qd_layout L;
    qd_plan P;
    int32_t dev = -1;
    if (qd_get_layout(ctx, &L) != QD_OK || qd_get_plan(ctx, &P) != QD_OK || qd_context_device(ctx, &dev) != QD_OK)
        return pfail(nullptr, QD_ERR_STATE, "the context needs a plan before a pipeline is made on it");
    std::unique_ptr<qd_pipe> p(new qd_pipe());
    p->ctx = ctx;
    p->device = dev;
    p->lay = L;
    p->plan = P;
    p->n_streams = 2 + L.n_streams;
    if (const char* e = getenv("QUADE_PIPE_INFLATE_FORM")) p->inflate_form = atoi(e) == 2 ? 2 : 3;  // (measurement: A/B of the inflaters inside the pipeline)
    if (const char* e = getenv("QUADE_PIPE_DEVICE_GUNZIP")) p->device_gunzip = atoi(e) ? 1 : 0;
    if (const char* e = getenv("QUADE_PIPE_INFLATE_OVERLAP")) p->inflate_overlap = atoi(e) ? 1 : 0;
    if (const char* e = getenv("QUADE_PIPE_CODER_STREAM")) p->coder_stream = atoi(e) ? 1 : 0;
    if (const char* e = getenv("QUADE_PIPE_PEEK")) p->peek_records = atoi(e) ? 1 : 0;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&p->cs, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p->ds, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p->sync_ev, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&p->is[0], hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&p->is[1], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p->tables_up, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&p->es, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p->formatted, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->coded, hipEventDisableTiming) != hipSuccess)
        return pfail(nullptr, QD_ERR_HIP, "stream creation failed");
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&p->out[i].done, hipEventDisableTiming) != hipSuccess) return pfail(nullptr, QD_ERR_HIP, "event creation failed");
    if (p->d_res.need(4 * sizeof(qd_scan_result) + 64) != hipSuccess || p->h_res.need(4 * sizeof(qd_scan_result) + 64) != hipSuccess)
        return pfail(nullptr, QD_ERR_HIP, "allocation failed");
    if (hipMemset(p->d_res.p, 0xFF, 4 * sizeof(qd_scan_result) + 64) != hipSuccess) return pfail(nullptr, QD_ERR_HIP, "hipMemset failed");
    *out = p.release();
    return QD_OK;
}

int qd_pipe_set_option(qd_pipe* p, const char* name, int64_t value) {
    if (!p || !name) return QD_ERR_INVALID;
    const std::string n(name);
    if (n == "batch_pairs" && value >= 1) p->batch_pairs = value;
    else if (n == "test_fail_inflate_batch") p->test_fail_inflate_batch = value;
    else if (n == "test_host_code_every" && value >= 0) p->test_host_code_every = value;
    else if (n == "member_slots_bytes" && value >= (1 << 20)) p->member_slots_bytes = value;
    else if (n == "inflate_streams" && (value == 1 || value == 2)) p->n_is = (int)value;
    else if (n == "inflate_form" && (value == 2 || value == 3)) p->inflate_form = (int)value;
    else if (n == "device_gunzip" && (value == 0 || value == 1)) p->device_gunzip = (int)value;
    else if (n == "inflate_overlap" && (value == 0 || value == 1)) p->inflate_overlap = (int)value;
    else if (n == "coder_stream" && (value == 0 || value == 1)) p->coder_stream = (int)value;
    else if (n == "peek_records" && (value == 0 || value == 1)) p->peek_records = (int)value;
    else return pfail(p, QD_ERR_INVALID, "unknown option " + n);
    return QD_OK;
}

int qd_pipe_run(qd_pipe* p, const qd_pipe_chunk* chunks, int32_t n_chunks, qd_pipe_stats* stats) {
    if (!p || n_chunks < 0 || (n_chunks && !chunks)) return pfail(p, QD_ERR_INVALID, "bad arguments");
    p->err.clear();
    PCHK(p, hipSetDevice(p->device));
    const auto run_t0 = std::chrono::steady_clock::now();
    g_alloc_seconds = 0;
    p->st = qd_pipe_stats_impl();  // the statistics are per call (a pipe that runs chunk after chunk: the caller adds them up)
    p->q3_jobs.clear();  // (a run that failed half way may have left blocks queued: their ring is gone)
    p->q3_parts.clear();
    p->gz_units0 = p->gz ? p->gz->stats().units : 0;
    const int ns = p->n_streams;
    for (int c = 0; c < n_chunks; ++c) {
        if (!chunks[c].r1 || !chunks[c].r2 || !chunks[c].i1 || (ns == 4 && !chunks[c].i2) || !chunks[c].sink) return pfail(p, QD_ERR_INVALID, "chunk without files or sink");
        const int level = qdio::sink_info(chunks[c].sink).level;
        if (level != 1 && level != -1) return pfail(p, QD_ERR_UNSUPPORTED, "the device codes gzip_level 1 and -1; other levels run on the host's pool");
    }
    p->max_r1_bytes = 0;
    p->r1_compressed = false;
    for (int c = 0; c < n_chunks; ++c) {
        struct stat sb;
        if (stat(chunks[c].r1, &sb) == 0) p->max_r1_bytes = std::max<int64_t>(p->max_r1_bytes, (int64_t)sb.st_size);
        const size_t n = strlen(chunks[c].r1);
        p->r1_compressed = p->r1_compressed || (n >= 3 && strcmp(chunks[c].r1 + n - 3, ".gz") == 0) || (n >= 3 && strcmp(chunks[c].r1 + n - 3, ".GZ") == 0);
    }
    p->reserved = false;
    {
        std::lock_guard<std::mutex> g(p->om);
        for (OutSet& o : p->out) o.busy = false;  // (a run that failed half way may have left one taken)
    }
    std::vector<std::unique_ptr<Feeder>> feeders;
    for (int s = 0; s < ns; ++s) {
        std::vector<FileSpec> files;
        for (int c = 0; c < n_chunks; ++c) {
            FileSpec f;
            f.path = s == 0 ? chunks[c].r1 : s == 1 ? chunks[c].r2 : s == 2 ? chunks[c].i1 : chunks[c].i2;
            f.start = chunks[c].start_offset[s] > 0 ? chunks[c].start_offset[s] : 0;
            files.push_back(std::move(f));
        }
        feeders.emplace_back(new Feeder(p->device, std::move(files), p->device_gunzip != 0));
    }
    for (auto& f : feeders) PCHK(p, f->start());
    {
        std::lock_guard<std::mutex> g(p->cm);
        p->collector_failed = false;
        p->collector_err.clear();
    }
    p->collector = std::thread(collector_thread, p);
    int rc = QD_OK;
    int64_t batch_index = 0;
    for (int c = 0; c < n_chunks && rc == QD_OK; ++c) {
        for (int s = 0; s < ns; ++s) p->win[s].path = s == 0 ? chunks[c].r1 : s == 1 ? chunks[c].r2 : s == 2 ? chunks[c].i1 : chunks[c].i2;
        if (chunks[c].begin_message) {
            fputs(chunks[c].begin_message, stdout);
            fflush(stdout);
        }
        rc = run_chunk(p, feeders, c, chunks[c], &batch_index);
        if (rc == QD_OK && chunks[c].end_message) {
            BatchOut m;
            m.message = chunks[c].end_message;
            to_collector(p, std::move(m));
        }
        std::lock_guard<std::mutex> g(p->cm);
        if (p->collector_failed) rc = pfail(p, QD_ERR_FORMAT, p->collector_err);
    }
    {
        BatchOut last;
        last.last = true;
        to_collector(p, std::move(last));
    }
    p->collector.join();
    (void)hipStreamSynchronize(p->cs);
    if (p->es) (void)hipStreamSynchronize(p->es);
    p->coded_pending = false;
    for (auto& f : feeders) f->stop();
    {
        std::lock_guard<std::mutex> g(p->cm);
        if (rc == QD_OK && p->collector_failed) rc = pfail(p, QD_ERR_FORMAT, p->collector_err);
    }
    if (stats) {
        stats->pairs = p->st.pairs;
        stats->batches = p->st.batches;
        stats->bgzf_blocks = p->st.bgzf_blocks;
        stats->host_inflated_runs = p->st.host_inflated_runs;
        stats->text_segments = p->st.text_segments;
        stats->pieces = p->st.pieces;
        stats->host_coded_pieces = p->st.host_coded_pieces;
        stats->text_in_bytes = p->st.text_in_bytes;
        stats->text_out_bytes = p->st.text_out_bytes;
        stats->gzip_bytes = p->st.gzip_bytes;
        stats->rescans = p->st.rescans;
        stats->wait_input_s = p->st.wait_input;
        stats->wait_sync_s = p->st.wait_sync;
        stats->wait_out_set_s = p->st.wait_out_set;
        stats->alloc_s = g_alloc_seconds;
        stats->collector_wait_s = p->st.collector_wait;
        stats->download_s = p->st.download;
        stats->append_s = p->st.append;
        stats->run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - run_t0).count();
        stats->gzip_steps = p->st.gzip_steps;
        stats->gzip_units = p->gz ? p->gz->stats().units - p->gz_units0 : 0;
        stats->gzip_members = p->st.gzip_members;
        stats->gzip_fallbacks = p->st.gzip_fallbacks;
    }
    return rc;
}

int qd_pipe_destroy(qd_pipe* p) {
    if (!p) return QD_OK;
    if (p->reserve_thread.joinable()) p->reserve_thread.join();
    (void)hipSetDevice(p->device);
    if (p->cs) (void)hipStreamSynchronize(p->cs);
    if (p->ds) (void)hipStreamSynchronize(p->ds);
    for (hipStream_t st : p->is)
        if (st) (void)hipStreamSynchronize(st);
    for (qd_pipe::Window& w : p->win) {
        for (hipEvent_t ev : w.inflated)
            if (ev) (void)hipEventDestroy(ev);
        for (DevBuf* b : {&w.buf[0], &w.buf[1], &w.tile_counts, &w.tile_base, &w.lines, &w.rec_tile, &w.recs, &w.status, &w.crc, &w.blk, &w.expect}) b->release();
    }
    for (DevBuf* b : {&p->jobs3, &p->status3, &p->scratch3}) b->release();
    for (qd_pipe::Window& w : p->win) {
        gz_close(w);
        for (DevBuf* b : {&w.gz.comp[0], &w.gz.comp[1], &w.gz.carried}) b->release();
    }
    delete p->gz;
    p->gz = nullptr;
    for (DevBuf* b : {&p->d_res, &p->matches, &p->matches_b, &p->rows_seq[0], &p->rows_seq[1], &p->rows_qual[0], &p->rows_qual[1], &p->rows_len[0], &p->rows_len[1], &p->codes, &p->mol,
                      &p->short_idx, &p->dest, &p->len1, &p->len2, &p->hist, &p->tmp, &p->perm, &p->sdest, &p->g1, &p->g2, &p->scan_tiles, &p->first, &p->g1_first,
                      &p->g2_first, &p->subs, &p->first_sub, &p->ranges, &p->crc, &p->tokens, &p->sub_out, &p->sub_bytes, &p->base1, &p->base2})
        b->release();
    for (OutSet& o : p->out) {
        for (DevBuf* b : {&o.text, &o.pieces, &o.members, &o.member_len, &o.member_off, &o.packed}) b->release();
        o.h_len.release();
        if (o.done) (void)hipEventDestroy(o.done);
    }
    p->h_res.release();
    p->h_first.release();
    for (int k = 0; k < 3; ++k) {
        p->slab[k].release();
        if (p->slab_ev[k]) (void)hipEventDestroy(p->slab_ev[k]);
    }
    if (p->sync_ev) (void)hipEventDestroy(p->sync_ev);
    if (p->tables_up) (void)hipEventDestroy(p->tables_up);
    if (p->formatted) (void)hipEventDestroy(p->formatted);
    if (p->coded) (void)hipEventDestroy(p->coded);
    if (p->es) (void)hipStreamDestroy(p->es);
    for (hipStream_t st : p->is)
        if (st) (void)hipStreamDestroy(st);
    if (p->cs) (void)hipStreamDestroy(p->cs);
    if (p->ds) (void)hipStreamDestroy(p->ds);
    delete p;
    return QD_OK;
}

}  // extern "C"
