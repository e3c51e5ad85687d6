// Host-buffer entry points over the device text stages (quade_text.hip): what qd_pipe_run chains on the device, one stage
// at a time, for bindings that hold text in host memory and for the tests (each stage on its own).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/quade_hip.h"
#include "quade_text.h"

namespace {
struct Dev {  // a scratch allocation freed at scope exit
    void* p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};
#define TCHK(call)                            \
    do {                                      \
        if ((call) != hipSuccess) return QD_ERR_HIP; \
    } while (0)
}  // namespace

extern "C" {

int64_t qd_dev_fastq_scan(int device_id, const uint8_t* text, int64_t text_len, int32_t at_eof, int32_t want_names, int32_t need,
                          int64_t line_cap, uint32_t* recs_out, int64_t recs_cap, uint32_t* result_out) {
    if (!text || text_len < 0 || text_len > (int64_t)1 << 30 || line_cap < 4 || !result_out) return QD_ERR_INVALID;
    TCHK(hipSetDevice(device_id));
    const uint32_t len = (uint32_t)text_len, n_tiles = len / QD_TEXT_TILE + 1;
    line_cap &= ~(int64_t)3;
    Dev d_text, tc, tb, lines, rt, recs, res;
    TCHK(d_text.alloc((size_t)len + 2 * QD_TEXT_TILE));
    TCHK(tc.alloc((size_t)(n_tiles + 2) * 4));
    TCHK(tb.alloc((size_t)(n_tiles + 2) * 4));
    TCHK(lines.alloc((size_t)line_cap * 4 + 64));
    TCHK(rt.alloc(((size_t)line_cap / 4 / 1024 + 4) * 4));
    TCHK(recs.alloc(((size_t)line_cap / 4 + 1) * sizeof(qd_rec)));
    TCHK(res.alloc(sizeof(qd_scan_result)));
    TCHK(hipMemset(res.p, 0xFF, sizeof(qd_scan_result)));
    if (len) TCHK(hipMemcpy(d_text.p, text, len, hipMemcpyHostToDevice));
    qd_scan_scratch sc;
    sc.tile_counts = tc.as<uint32_t>();
    sc.tile_base = tb.as<uint32_t>();
    sc.lines = lines.as<uint32_t>();
    sc.line_cap = (uint32_t)line_cap;
    sc.rec_tile = rt.as<uint32_t>();
    sc.recs = recs.as<qd_rec>();
    TCHK(qd_text_scan(d_text.as<uint8_t>(), len, at_eof, want_names, (uint32_t)need, sc, res.as<qd_scan_result>(), nullptr));
    TCHK(hipDeviceSynchronize());
    qd_scan_result r;
    TCHK(hipMemcpy(&r, res.p, sizeof r, hipMemcpyDeviceToHost));
    memcpy(result_out, &r, sizeof r);
    if (r.overflow) return 0;
    const int64_t n = std::min<int64_t>(r.n_kept, recs_cap);
    if (n && recs_out) TCHK(hipMemcpy(recs_out, recs.p, (size_t)n * sizeof(qd_rec), hipMemcpyDeviceToHost));
    return r.n_kept;
}

int qd_dev_crc32(int device_id, const uint8_t* data, int64_t n, int64_t range_bytes, uint32_t* crc_out) {
    if ((!data && n) || n < 0 || range_bytes < 1 || range_bytes > 65536 || !crc_out) return QD_ERR_INVALID;
    TCHK(hipSetDevice(device_id));
    std::vector<qd_crc_range> ranges;
    for (int64_t a = 0; a < n; a += range_bytes) ranges.push_back(qd_crc_range{(uint64_t)a + 3, (uint32_t)std::min<int64_t>(range_bytes, n - a), 0});
    const uint32_t first[2] = {0, (uint32_t)ranges.size()};
    Dev d, dr, dc, df, out;
    TCHK(d.alloc((size_t)n + 16));
    TCHK(dr.alloc(ranges.size() * sizeof(qd_crc_range)));
    TCHK(dc.alloc(ranges.size() * 4));
    TCHK(df.alloc(8));
    TCHK(out.alloc(4));
    if (n) TCHK(hipMemcpy(d.as<uint8_t>() + 3, data, (size_t)n, hipMemcpyHostToDevice));  // (a misaligned start on purpose)
    if (!ranges.empty()) TCHK(hipMemcpy(dr.p, ranges.data(), ranges.size() * sizeof(qd_crc_range), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(df.p, first, 8, hipMemcpyHostToDevice));
    TCHK(qd_text_crc32(d.as<uint8_t>(), dr.as<qd_crc_range>(), (uint32_t)ranges.size(), dc.as<uint32_t>(), nullptr));
    TCHK(qd_text_crc32_combine(dr.as<qd_crc_range>(), dc.as<uint32_t>(), df.as<uint32_t>(), 1, out.as<uint32_t>(), 1, nullptr));
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(crc_out, out.p, 4, hipMemcpyDeviceToHost));
    return QD_OK;
}

int qd_dev_sort_by_dest(int device_id, const uint16_t* dest, int64_t n, int32_t n_dest, const uint32_t* len, uint32_t* perm_out, uint32_t* offsets_out) {
    if (!dest || n < 1 || n > 0x7FFFFFFF || n_dest < 1 || n_dest > 65536 || !perm_out) return QD_ERR_INVALID;
    TCHK(hipSetDevice(device_id));
    const size_t H = 256 * (((size_t)n + 1023) / 1024);
    Dev d, hist, tmp, perm, dl, tiles, g;
    TCHK(d.alloc((size_t)n * 2));
    TCHK(hist.alloc((H + H / 4096 + 8) * 4));
    TCHK(tmp.alloc((size_t)n * 4));
    TCHK(perm.alloc((size_t)n * 4));
    TCHK(dl.alloc((size_t)n * 4));
    TCHK(tiles.alloc(((size_t)n / 4096 + 4) * 4));
    TCHK(g.alloc(((size_t)n + 1) * 4));
    TCHK(hipMemcpy(d.p, dest, (size_t)n * 2, hipMemcpyHostToDevice));
    TCHK(qd_text_sort_by_dest(d.as<uint16_t>(), (uint32_t)n, (uint32_t)n_dest, hist.as<uint32_t>(), tmp.as<uint32_t>(), perm.as<uint32_t>(), nullptr));
    if (len && offsets_out) {
        TCHK(hipMemcpy(dl.p, len, (size_t)n * 4, hipMemcpyHostToDevice));
        TCHK(qd_text_scan_gathered(dl.as<uint32_t>(), perm.as<uint32_t>(), (uint32_t)n, tiles.as<uint32_t>(), g.as<uint32_t>(), nullptr, nullptr, nullptr));
    }
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(perm_out, perm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (len && offsets_out) TCHK(hipMemcpy(offsets_out, g.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    return QD_OK;
}

}  // extern "C"
