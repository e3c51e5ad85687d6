// C ABI of libquade_hip.so: context, plan, barcode table, launches, counters, pinned slots.
// Host C++ over the HIP runtime; kernels live in quade_kernels.hip.  See include/quade_hip.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is bound at run time (dlopen), single-GPU users never load it

#include <algorithm>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/quade_hip.h"
#include "quade_common.h"
#include "quade_kernels.h"
#include "quade_inflate.h"
#include "quade_inflate3.h"
#include "quade_deflate.h"

typedef uint64_t u64;

namespace {

thread_local std::string g_create_error;

struct Slot {
    hipStream_t stream = nullptr;   // kernel + downloads of this slot
    hipEvent_t uploaded = nullptr;  // recorded on the context's upload stream behind this slot's H2D copies
    uint8_t* h_base = nullptr;  // pinned
    uint8_t* d_base = nullptr;
    qd_slot_buffers h{};        // host views
    qd_slot_buffers d{};        // device views (same struct, device pointers)
    // pinned + device: the two streams' lists merged (unique, ascending; 2 * short_cap indices), then the
    // listed reads' lengths per stream (2 * short_cap bytes each): all the fixup kernel needs of the len rows
    uint32_t* h_short = nullptr;
    uint32_t* d_short = nullptr;
    bool busy = false;
};

enum { K_FAST = 1, K_GENERIC = 2 };

}  // namespace

struct qd_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    mutable std::string err;
    char dev_name[256] = {0};
    int cu = 0;
    int64_t total_mem = 0;

    bool have_plan = false, have_table = false;
    qd_plan plan{};
    qd_layout lay{};

    // host copy of the barcodes
    int32_t S = 0;
    std::vector<uint8_t> bc;
    std::vector<int32_t> bc_off;

    // device table
    uint32_t* d_slots_fast = nullptr;
    uint32_t* d_slots_gen = nullptr;
    u64* d_bk16 = nullptr;
    u64* d_bk32 = nullptr;
    u64* d_bkv = nullptr;   // wide plans: per barcode its two slices as the kernel extracts them (confirmation compare)
    uint8_t* d_blen = nullptr;
    uint32_t mask_fast = 0, mask_gen = 0, seed_fast = 0, seed_gen = 0;
    bool fast_ok = false;
    bool wide = false;      // the fast table holds nibble-packed keys (16 < K <= 32)
    uint32_t lds_bk_off = 0, lds_hist_off = 0, lds_strip_off = 0;
    size_t lds_bytes = 0;       // table image + histogram
    size_t lds_strip_bytes = 0; // per wave: 128*M bytes of molecular staging (M % 4 == 0), else 0
    int opt_wg_per_cu = 0;    // 0 = occupancy query
    int opt_force_generic = 0;
    int opt_block = 0;        // 0 = automatic
    int opt_slot_factor = 0;  // 0 = automatic: QD_SLOT_FACTOR slots per barcode, half that for sample sheets whose image then fits a CU three times
    int opt_mol_strips = 1;   // LDS-staged molecular stores in the fast kernel
    int opt_kernel = 0;       // 0 = automatic, K_FAST / K_GENERIC
    QdKernelCache kcache;     // per-context launch memo (attribute set, occupancy)
    // streams this context has work on (its own, its slots', the caller's) with an event recorded after
    // the last operation issued on each: waits are scoped to the context, never the whole device
    std::vector<std::pair<hipStream_t, hipEvent_t>> tracked;

    // counters
    qd_row_t* d_partial = nullptr;  // [partial_rows][cnt_stride] 32-bit rows the kernels flush into
    u64* d_acc = nullptr;      // [cnt_stride] 64-bit totals: rows folded so far + demux_fixup's signed moves
    u64* d_counts = nullptr;   // [cnt_stride + 4]: 2S+1 counters, then TOTAL at [cnt_stride] (reduce scratch)
    u64* d_total = nullptr;    // all-reduce result, same shape
    uint32_t partial_rows = 0, cnt_stride = 0;
    uint64_t total_pairs = 0;
    uint64_t pairs_in_rows = 0;             // pairs launched since the rows were last folded (bounds every row counter)
    uint64_t fold_limit = 0xFFFFFFFFull;    // option "fold_pairs": fold before pairs_in_rows would pass this
    uint64_t folds = 0;

    std::vector<Slot> slots;
    int64_t slot_pairs = 0;
    // One upload stream for all slots: their H2D copies run one after the other at the full link rate and a
    // slot's kernel starts as soon as ITS rows have arrived.  With the uploads on the slots' own streams the
    // copies of all submitted slots shared the link, finished together, and the link then idled while the
    // kernels, the downloads and the host's next submits went by (48.7 of the link's 57 GB/s, tools/h2d_probe.py).
    hipStream_t up_stream = nullptr;
};

namespace {

int fail(const qd_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), QD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

hipStream_t resolve_stream(const qd_ctx* c, void* stream) {
    return stream == QD_STREAM_CONTEXT ? c->stream : (hipStream_t)stream;  // NULL = HIP's null stream
}

// remember that `st` carries work of this context up to now
hipError_t track(qd_ctx* c, hipStream_t st) {
    for (auto& t : c->tracked)
        if (t.first == st) return hipEventRecord(t.second, st);
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    c->tracked.emplace_back(st, ev);
    return hipEventRecord(ev, st);
}

// host waits for everything this context issued
hipError_t wait_all(qd_ctx* c) {
    for (auto& t : c->tracked) {
        hipError_t e = hipEventSynchronize(t.second);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(c->stream);
}

// the context's own stream waits (on the device) for everything this context issued elsewhere
hipError_t join_into_own_stream(qd_ctx* c) {
    for (auto& t : c->tracked) {
        if (t.first == c->stream) continue;
        hipError_t e = hipStreamWaitEvent(c->stream, t.second, 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void forget_stream(qd_ctx* c, hipStream_t st) {
    for (size_t i = 0; i < c->tracked.size(); ++i)
        if (c->tracked[i].first == st) {
            (void)hipEventDestroy(c->tracked[i].second);
            c->tracked.erase(c->tracked.begin() + (long)i);
            return;
        }
}

// canonical key of a byte string: little-endian packed, zero padded
void canon(const uint8_t* b, int len, u64 w[QD_KEY_WORDS]) {
    for (int i = 0; i < QD_KEY_WORDS; ++i) w[i] = 0;
    for (int i = 0; i < len && i < QD_MAX_KEY; ++i) w[i >> 3] |= (u64)b[i] << (8 * (i & 7));
}

// open-addressing table over the given barcode ordinals; picks the seed with the shortest
// worst-case probe sequence.  Returns slots (size mask+1).
std::vector<uint32_t> build_slots(const std::vector<int>& ids, const std::vector<u64>& keys32,
                                  const std::vector<uint8_t>& blen, uint32_t& mask, uint32_t& seed,
                                  const std::vector<u64>* packed16 = nullptr, uint32_t K = 0, uint32_t slot_factor = QD_SLOT_FACTOR) {
    // >= 4 slots per barcode (half that many cost +8 % on S = 1536 and +13 % on S = 96, twice as many nothing:
    // profiles/r02_slot_table_load.txt), and never fewer than 256: a dozen barcodes in 64 slots have every lane of
    // a wave probing the same few LDS words (S = 12: -7 % with 256 slots)
    uint32_t m = 256;
    while (m < slot_factor * (uint32_t)ids.size()) m <<= 1;
    mask = m - 1;
    std::vector<uint32_t> best;
    uint32_t best_worst = ~0u;
    for (uint32_t sd = 0; sd < 8; ++sd) {
        std::vector<uint32_t> t(m, QD_EMPTY_SLOT);
        uint32_t worst = 0;
        for (int id : ids) {
            const uint32_t h = packed16 ? qd_hash_wide((*packed16)[2 * (size_t)id], (*packed16)[2 * (size_t)id + 1], K, sd)
                                        : qd_hash_key(&keys32[(size_t)id * QD_KEY_WORDS], blen[id], sd);
            uint32_t s = h & mask, probes = 1;
            while (t[s] != QD_EMPTY_SLOT) {
                s = (s + 1) & mask;
                ++probes;
            }
            t[s] = qd_slot_entry(h, (uint32_t)id);
            worst = std::max(worst, probes);
        }
        if (worst < best_worst) {
            best_worst = worst;
            best.swap(t);
            seed = sd;
        }
        if (best_worst <= 2) break;
    }
    return best;
}

void free_table(qd_ctx* c) {
    if (c->d_slots_fast) (void)hipFree(c->d_slots_fast);
    if (c->d_slots_gen) (void)hipFree(c->d_slots_gen);
    if (c->d_bk16) (void)hipFree(c->d_bk16);
    if (c->d_bk32) (void)hipFree(c->d_bk32);
    if (c->d_bkv) (void)hipFree(c->d_bkv);
    c->d_bkv = nullptr;
    if (c->d_blen) (void)hipFree(c->d_blen);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->d_acc) (void)hipFree(c->d_acc);
    c->d_acc = nullptr;
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_total) (void)hipFree(c->d_total);
    c->d_total = nullptr;
    c->d_slots_fast = c->d_slots_gen = nullptr;
    c->d_bk16 = c->d_bk32 = nullptr;
    c->d_blen = nullptr;
    c->d_partial = nullptr;
    c->d_counts = nullptr;
    c->have_table = false;
}

// (re)build the device table from the host barcodes and the current plan
int rebuild(qd_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_all(c));  // nothing of this context may still read the old table
    free_table(c);
    const int S = c->S;
    const int K = c->lay.key_width;
    std::vector<u64> k32((size_t)std::max(S, 1) * QD_KEY_WORDS, 0), k16((size_t)std::max(S, 1) * 2, 0);
    std::vector<uint8_t> blen(std::max(S, 1), 0);
    std::vector<int> ids_fast, ids_gen;
    std::map<std::string, int> seen;
    // Wide plan: 16 < K <= 32 with every slice and window inside 16 bytes (so both index reads carry a part of
    // the key) and at most 8 molecular bytes per read; its fast table is built over nibble-packed keys, which
    // is injective only on the alphabet the reference admits (src/Sample.py:40,141): any K-long barcode with
    // another byte sends the plan to the generic kernel.
    const qd_layout& Lw = c->lay;
    const int w1 = c->plan.idx1_end - c->plan.idx1_start, w2 = c->plan.dual ? c->plan.idx2_end - c->plan.idx2_start : 0;
    bool wide = K > 16 && K <= 32 && Lw.n_streams == 2 && w1 <= 16 && w2 <= 16 && Lw.mol_width <= 16;
    for (int k = 0; wide && k < Lw.n_streams; ++k) {
        const int mw = (k == 0 ? c->plan.mol1_end - c->plan.mol1_start : c->plan.mol2_end - c->plan.mol2_start);
        wide = Lw.seq_stride[k] <= 16 && Lw.qual_stride[k] <= 16 && mw <= 8;
    }
    std::vector<u64> kv((size_t)std::max(S, 1) * 4, 0);
    for (int i = 0; i < S; ++i) {
        const uint8_t* b = c->bc.data() + c->bc_off[i];
        const int len = c->bc_off[i + 1] - c->bc_off[i];
        std::string s((const char*)b, (size_t)len);
        if (seen.count(s)) {
            char m[128];
            snprintf(m, sizeof m, "barcode %d duplicates barcode %d (Index is not unique)", i, seen[s]);
            return fail(c, QD_ERR_BARCODE, m);
        }
        seen[s] = i;
        blen[i] = (uint8_t)std::min(len, 255);
        if (len == 0 || len > QD_MAX_KEY) continue;  // can never equal a slice (empty: see DESIGN.md)
        canon(b, len, &k32[(size_t)i * QD_KEY_WORDS]);
        ids_gen.push_back(i);
        if (len == K && K <= 16) {
            k16[2 * (size_t)i] = k32[(size_t)i * QD_KEY_WORDS];
            k16[2 * (size_t)i + 1] = k32[(size_t)i * QD_KEY_WORDS + 1];
            ids_fast.push_back(i);
        } else if (len == K && wide) {
            for (int j = 0; j < len; ++j)
                if (!strchr("ACGTN", b[j]) || b[j] == 0) wide = false;
            u64 s1[QD_KEY_WORDS], s2[QD_KEY_WORDS];
            canon(b, w1, s1);           // the barcode's part in index read 1, then in index read 2
            canon(b + w1, len - w1, s2);
            u64* v = &kv[(size_t)i * 4];
            v[0] = s1[0];
            v[1] = s1[1];
            v[2] = s2[0];
            v[3] = s2[1];
            qd_wide_key(v[0], v[1], v[2], v[3], w1, &k16[2 * (size_t)i], &k16[2 * (size_t)i + 1]);
            ids_fast.push_back(i);
        }
    }
    if (K > 16 && !wide) ids_fast.clear();
    c->wide = wide;
    const uint32_t slot_factor = c->opt_slot_factor > 0 ? (uint32_t)c->opt_slot_factor : (uint32_t)QD_SLOT_FACTOR;
    std::vector<uint32_t> sf = wide ? build_slots(ids_fast, k32, blen, c->mask_fast, c->seed_fast, &k16, (uint32_t)K, slot_factor)
                                    : build_slots(ids_fast, k32, blen, c->mask_fast, c->seed_fast, nullptr, 0, slot_factor);
    std::vector<uint32_t> sg = build_slots(ids_gen, k32, blen, c->mask_gen, c->seed_gen);

    HIPCHK(c, hipMalloc(&c->d_slots_fast, sf.size() * 4));
    HIPCHK(c, hipMalloc(&c->d_slots_gen, sg.size() * 4));
    HIPCHK(c, hipMalloc(&c->d_bk16, k16.size() * 8));
    HIPCHK(c, hipMalloc(&c->d_bk32, k32.size() * 8));
    HIPCHK(c, hipMalloc(&c->d_blen, blen.size()));
    HIPCHK(c, hipMalloc(&c->d_bkv, kv.size() * 8));
    HIPCHK(c, hipMemcpy(c->d_bkv, kv.data(), kv.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_slots_fast, sf.data(), sf.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_slots_gen, sg.data(), sg.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_bk16, k16.data(), k16.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_bk32, k32.data(), k32.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_blen, blen.data(), blen.size(), hipMemcpyHostToDevice));

    // LDS image of the fast kernel: slots | keys (16 B each) | histogram (2S+1)
    c->lds_bk_off = (uint32_t)(((size_t)(c->mask_fast + 1) * 4 + 15) & ~(size_t)15);
    c->lds_hist_off = c->lds_bk_off + (uint32_t)S * 16;
    c->lds_bytes = ((size_t)c->lds_hist_off + (size_t)(2 * S + 1) * 4 + 15) & ~(size_t)15;
    c->lds_strip_off = (uint32_t)c->lds_bytes;
    // "UMI in index read 1" shapes (quade_kernels.hip StaticUmi1): 8-base barcodes at the start of both index reads, a molecular
    // index of 9..12 bases right behind the first one, none in the second -- rows of 18 / 20 bytes, static kernels only
    const bool umi1 = [&] {
        const qd_layout& Lu = c->lay;
        const qd_plan& Pu = c->plan;
        const int mw = Pu.mol1_end - Pu.mol1_start;
        return Lu.n_streams == 2 && K == 16 && Pu.idx1_end - Pu.idx1_start == 8 && Pu.idx2_end - Pu.idx2_start == 8 && mw >= 9 && mw <= 12 &&
               Pu.mol2_end == Pu.mol2_start && Lu.mol_width == mw && Pu.mol1_start == Pu.idx1_end && Lu.seq_off[0] == Pu.idx1_start &&
               Lu.seq_off[1] == Pu.idx2_start && Lu.seq_stride[0] == ((8 + mw + 1) & ~1) && Lu.seq_stride[1] == 8 && Lu.qual_stride[0] == 8 &&
               Lu.qual_stride[1] == 8;
    }();
    // "UMI in both index reads" (StaticUmi2, r05): the same in both reads -- rows of 18 / 20 bytes in both streams, 18 .. 24 bytes of
    // molecular index per pair
    const bool umi2 = [&] {
        const qd_layout& Lu = c->lay;
        const qd_plan& Pu = c->plan;
        const int mw = Pu.mol1_end - Pu.mol1_start;
        return Lu.n_streams == 2 && K == 16 && Pu.idx1_end - Pu.idx1_start == 8 && Pu.idx2_end - Pu.idx2_start == 8 && mw >= 9 && mw <= 12 &&
               Pu.mol2_end - Pu.mol2_start == mw && Lu.mol_width == 2 * mw && Pu.mol1_start == Pu.idx1_end && Pu.mol2_start == Pu.idx2_end &&
               Lu.seq_off[0] == Pu.idx1_start && Lu.seq_off[1] == Pu.idx2_start && Lu.seq_stride[0] == ((8 + mw + 1) & ~1) &&
               Lu.seq_stride[1] == ((8 + mw + 1) & ~1) && Lu.qual_stride[0] == 8 && Lu.qual_stride[1] == 8;
    }();
    c->lds_strip_bytes = (c->lay.mol_width > 0 && (c->lay.mol_width % 4 == 0 || umi1 || umi2) && c->opt_mol_strips) ? (size_t)128 * c->lay.mol_width : 0;
    if (c->lds_bytes + 16 * c->lds_strip_bytes > 150 * 1024) c->lds_strip_bytes = 0;  // 16 waves per workgroup at most

    const qd_layout& L = c->lay;
    bool ok = (K >= 1 && K <= 16 && L.mol_width <= 16 && c->lds_bytes <= 150 * 1024);
    for (int k = 0; k < L.n_streams; ++k) {
        // rows of two reads in at most two 16-byte loads; slices of at most 8 bytes
        ok = ok && L.seq_stride[k] <= 16 && L.qual_stride[k] <= 8;
        const int mw = (k == 0 ? c->plan.mol1_end - c->plan.mol1_start : c->plan.mol2_end - c->plan.mol2_start);
        ok = ok && mw <= 8 && L.qual_width[k] <= 8;
    }
    c->fast_ok = ok || ((c->wide || umi1 || umi2) && c->lds_bytes <= 150 * 1024);

    c->cnt_stride = (uint32_t)((2 * S + 1 + 3) & ~3);
    // one counter row per workgroup (modulo), capped at 64 MiB of rows for very large tables
    c->partial_rows = (uint32_t)std::max<size_t>(8, std::min<size_t>((size_t)c->cu * 8, ((size_t)64 << 20) / ((size_t)c->cnt_stride * sizeof(qd_row_t))));
    HIPCHK(c, hipMalloc(&c->d_partial, (size_t)c->partial_rows * c->cnt_stride * sizeof(qd_row_t)));
    HIPCHK(c, hipMalloc(&c->d_acc, (size_t)c->cnt_stride * 8));
    HIPCHK(c, hipMalloc(&c->d_counts, ((size_t)c->cnt_stride + 4) * 8));
    HIPCHK(c, hipMalloc(&c->d_total, ((size_t)c->cnt_stride + 4) * 8));
    HIPCHK(c, hipMemset(c->d_partial, 0, (size_t)c->partial_rows * c->cnt_stride * sizeof(qd_row_t)));
    HIPCHK(c, hipMemset(c->d_acc, 0, (size_t)c->cnt_stride * 8));
    c->total_pairs = 0;
    c->pairs_in_rows = 0;
    c->have_table = true;
    return QD_OK;
}

void fill_params(const qd_ctx* c, DemuxParams& p, bool fast) {
    memset(&p, 0, sizeof p);
    const qd_layout& L = c->lay;
    const qd_plan& P = c->plan;
    p.partial = c->d_partial;
    p.adjust = c->d_acc;
    p.slots = fast ? c->d_slots_fast : c->d_slots_gen;
    p.slot_mask = fast ? c->mask_fast : c->mask_gen;
    p.seed = fast ? c->seed_fast : c->seed_gen;
    p.gslots = c->d_slots_gen;
    p.gmask = c->mask_gen;
    p.gseed = c->seed_gen;
    p.bk16 = c->d_bk16;
    p.bk32 = c->d_bk32;
    p.bkv = c->d_bkv;
    p.wide = (fast && c->wide) ? 1 : 0;
    p.blen = c->d_blen;
    p.n_samples = (uint32_t)c->S;
    p.cnt_stride = c->cnt_stride;
    p.partial_rows = c->partial_rows;
    p.lds_bk_off = c->lds_bk_off;
    p.lds_hist_off = c->lds_hist_off;
    p.mol_strip_off = (fast && c->lds_strip_bytes) ? c->lds_strip_off : 0;
    p.thr = (uint32_t)(P.min_qual + 33);
    p.n_streams = L.n_streams;
    p.K = L.key_width;
    p.M = L.mol_width;
    const int is[2] = {P.idx1_start, P.idx2_start}, ie[2] = {P.idx1_end, P.idx2_end};
    const int ms[2] = {P.mol1_start, P.mol2_start}, me[2] = {P.mol1_end, P.mol2_end};
    for (int k = 0; k < 2; ++k) {
        p.seq_stride[k] = L.seq_stride[k];
        p.qual_stride[k] = L.qual_stride[k];
        const bool used = k < L.n_streams;
        p.idx_w[k] = used ? ie[k] - is[k] : 0;
        p.mol_w[k] = used ? me[k] - ms[k] : 0;
        p.idx_col[k] = is[k];
        p.mol_col[k] = ms[k];
        p.idx_off[k] = p.idx_w[k] ? is[k] - L.seq_off[k] : 0;
        p.mol_off[k] = p.mol_w[k] ? ms[k] - L.seq_off[k] : 0;
        p.idx_mask[k] = p.idx_w[k] >= 8 ? ~0ull : ((1ull << (8 * p.idx_w[k])) - 1);
        p.idx_mask_hi[k] = p.idx_w[k] >= 16 ? ~0ull : (p.idx_w[k] > 8 ? ((1ull << (8 * (p.idx_w[k] - 8))) - 1) : 0);
        p.mol_mask[k] = p.mol_w[k] >= 8 ? ~0ull : ((1ull << (8 * p.mol_w[k])) - 1);
    }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// which kernel takes a batch: dense = per-read lengths apply to (potentially) every pair
int pick_kernel(const qd_ctx* c, bool dense_len) {
    if (!c->fast_ok || dense_len || c->opt_force_generic || c->opt_kernel == K_GENERIC) return K_GENERIC;
    return K_FAST;
}

// n_short < 0: no exception list (len rows, if any, apply to every pair -> generic kernel)
// short_len: the listed reads' lengths, compact ([k][i] for short_idx[i]); NULL: rows->len holds them per pair
// The kernels count into 32-bit rows.  A row counter never exceeds the number of pairs launched since the rows
// were last emptied, so before that number could pass 2^32 - 1 the rows are added to the 64-bit totals and
// zeroed: once per ~4.3 G pairs, behind everything the context has in flight (a host wait, ~tens of us).
int fold_rows(qd_ctx* c) {
    HIPCHK(c, wait_all(c));
    const uint32_t ncnt = (uint32_t)(2 * c->S + 1);
    hipError_t e = qd_launch_reduce(c->d_partial, c->partial_rows, c->cnt_stride, ncnt, c->d_acc, c->d_acc, c->stream);
    if (e != hipSuccess) return fail(c, QD_ERR_HIP, std::string("fold launch: ") + hipGetErrorString(e));
    HIPCHK(c, hipMemsetAsync(c->d_partial, 0, (size_t)c->partial_rows * c->cnt_stride * sizeof(qd_row_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pairs_in_rows = 0;
    ++c->folds;
    return QD_OK;
}

int launch(qd_ctx* c, int64_t n, const qd_rows* rows, uint16_t* codes, uint8_t* mol, hipStream_t st,
           int64_t n_short = -1, const uint32_t* short_idx = nullptr, const uint8_t* const* short_len = nullptr) {
    const qd_layout& L = c->lay;
    bool has_len = false;
    for (int k = 0; k < L.n_streams; ++k) {
        if (!rows->seq[k] || !rows->qual[k]) return fail(c, QD_ERR_INVALID, "NULL row pointer");
        if (!aligned16(rows->seq[k]) || !aligned16(rows->qual[k]))
            return fail(c, QD_ERR_INVALID, "row buffers must be 16-byte aligned");
        has_len = has_len || rows->len[k] != nullptr;
    }
    if (short_len) has_len = true;
    if (!codes || !aligned16(codes)) return fail(c, QD_ERR_INVALID, "codes buffer NULL or not 16-byte aligned");
    if (L.mol_width > 0 && (!mol || !aligned16(mol)))
        return fail(c, QD_ERR_INVALID, "mol buffer NULL or not 16-byte aligned");
    if (n > (int64_t)0xFFFFFFFF) return fail(c, QD_ERR_INVALID, "more than 2^32-1 pairs in one batch");
    if (c->pairs_in_rows + (uint64_t)n > c->fold_limit) {
        const int r = fold_rows(c);
        if (r != QD_OK) return r;
    }
    // a listed minority of short reads: fast kernel for everybody, the listed pairs redone afterwards
    const bool sparse = has_len && n_short >= 0 && n_short <= n / 2;
    const int kind = pick_kernel(c, has_len && !sparse);
    DemuxParams p;
    fill_params(c, p, kind != K_GENERIC);
    for (int k = 0; k < L.n_streams; ++k) {
        p.seq[k] = rows->seq[k];
        p.qual[k] = rows->qual[k];
        p.len[k] = rows->len[k];
    }
    p.codes = codes;
    p.mol = mol;
    p.n = n;
    hipError_t e;
    bool fast_done = false;
    if (kind == K_FAST) {
        e = qd_launch_fast(p, c->kcache, c->cu, c->opt_wg_per_cu, c->opt_block, c->lds_bytes, c->lds_strip_bytes, st);
        fast_done = e == hipSuccess;
        if (e == hipErrorInvalidValue) {  // no instantiation for this layout in this build (tuning builds leave static shapes out): the generic kernel takes it
            (void)hipGetLastError();
            fill_params(c, p, false);
            for (int k = 0; k < L.n_streams; ++k) {
                p.seq[k] = rows->seq[k];
                p.qual[k] = rows->qual[k];
                p.len[k] = rows->len[k];
            }
            p.codes = codes;
            p.mol = mol;
            p.n = n;
        }
    }
    if (!fast_done && (kind != K_FAST || e == hipErrorInvalidValue)) {
        const int64_t nb = (n + QD_GEN_BLOCK - 1) / QD_GEN_BLOCK;
        const int grid = (int)std::min<int64_t>(nb, (int64_t)c->cu * 8);
        e = qd_launch_generic(p, grid, st);
    }
    if (e != hipSuccess) return fail(c, QD_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    if (fast_done && sparse && n_short > 0) {
        p.exc = short_idx;
        p.n_exc = (uint32_t)n_short;
        for (int k = 0; k < L.n_streams; ++k) p.exc_len[k] = short_len ? short_len[k] : nullptr;
        e = qd_launch_fixup(p, st);
        if (e != hipSuccess) return fail(c, QD_ERR_HIP, std::string("fixup launch: ") + hipGetErrorString(e));
    }
    c->total_pairs += (uint64_t)n;
    c->pairs_in_rows += (uint64_t)n;
    e = track(c, st);
    if (e != hipSuccess) return fail(c, QD_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
    return QD_OK;
}

}  // namespace

extern "C" {

int qd_version(void) { return QD_ABI_VERSION; }

const char* qd_strerror(int code) {
    switch (code) {
        case QD_OK: return "ok";
        case QD_ERR_INVALID: return "invalid argument";
        case QD_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
        case QD_ERR_HIP: return "HIP runtime error";
        case QD_ERR_STATE: return "plan and barcodes must be set first";
        case QD_ERR_UNSUPPORTED: return "plan outside the supported envelope";
        case QD_ERR_BARCODE: return "barcode table rejected";
        case QD_ERR_FORMAT: return "malformed fastq text";
    }
    return "unknown error";
}

const char* qd_last_error(const qd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int qd_device_count(int32_t* n_devices) {
    if (!n_devices) return fail(nullptr, QD_ERR_INVALID, "n_devices is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *n_devices = 0;
        return fail(nullptr, QD_ERR_NO_DEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) + ")");
    }
    *n_devices = n;
    return QD_OK;
}

int qd_create(int device_id, qd_ctx** out) {
    if (!out) return fail(nullptr, QD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, QD_ERR_NO_DEVICE,
                    std::string("no HIP device visible (") + hipGetErrorString(e) + "); this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, QD_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return fail(nullptr, QD_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, QD_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    if ((e = hipSetDevice(device_id)) != hipSuccess)
        return fail(nullptr, QD_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    qd_ctx* c = new qd_ctx();
    c->device = device_id;
    c->cu = prop.multiProcessorCount;
    c->total_mem = (int64_t)prop.totalGlobalMem;
    snprintf(c->dev_name, sizeof c->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return fail(nullptr, QD_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = c;
    return QD_OK;
}

int qd_destroy(qd_ctx* c) {
    if (!c) return QD_OK;
    (void)hipSetDevice(c->device);
    (void)wait_all(c);
    qd_slots_destroy(c);
    free_table(c);
    for (auto& t : c->tracked) (void)hipEventDestroy(t.second);
    c->tracked.clear();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return QD_OK;
}

int qd_device_info(const qd_ctx* c, char* name, int32_t cap, int32_t* cus, int64_t* mem) {
    if (!c) return QD_ERR_INVALID;
    if (name && cap > 0) snprintf(name, (size_t)cap, "%s", c->dev_name);
    if (cus) *cus = c->cu;
    if (mem) *mem = c->total_mem;
    return QD_OK;
}

int qd_set_plan(qd_ctx* c, const qd_plan* plan) {
    if (!c || !plan) return QD_ERR_INVALID;
    qd_layout L;
    const int r = qd_plan_layout(plan, &L);
    if (r != QD_OK) return fail(c, r, "plan rejected: positions must satisfy 0 <= start <= end <= 255, window <= 64, "
                                       "fused barcode <= 32, 0 <= minimal_qual <= 40");
    if (!c->slots.empty()) return fail(c, QD_ERR_STATE, "destroy the slots before changing the plan");
    c->plan = *plan;
    c->lay = L;
    c->have_plan = true;
    if (c->S > 0 || c->have_table) return rebuild(c);
    return QD_OK;
}

int qd_get_layout(const qd_ctx* c, qd_layout* out) {
    if (!c || !out) return QD_ERR_INVALID;
    if (!c->have_plan) return fail(c, QD_ERR_STATE, "qd_set_plan first");
    *out = c->lay;
    return QD_OK;
}

int qd_get_plan(const qd_ctx* c, qd_plan* out) {
    if (!c || !out) return QD_ERR_INVALID;
    if (!c->have_plan) return fail(c, QD_ERR_STATE, "no plan set");
    *out = c->plan;
    return QD_OK;
}

int qd_context_device(const qd_ctx* c, int32_t* device_id) {
    if (!c || !device_id) return QD_ERR_INVALID;
    *device_id = c->device;
    return QD_OK;
}

int qd_set_barcodes(qd_ctx* c, int32_t S, const uint8_t* barcodes, const int32_t* offsets) {
    if (!c || S < 0 || S > QD_MAX_SAMPLES || (S > 0 && (!barcodes || !offsets)))
        return fail(c, QD_ERR_INVALID, "bad barcode arguments (0 <= n_samples <= 32767)");
    if (!c->have_plan) return fail(c, QD_ERR_STATE, "qd_set_plan first");
    for (int i = 0; i < S; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(c, QD_ERR_INVALID, "offsets must be non-decreasing");
    c->S = S;
    c->bc.assign(barcodes, barcodes + (S ? offsets[S] : 0));
    c->bc_off.assign(offsets, offsets + (S ? S + 1 : 0));
    if (S == 0) c->bc_off.assign(1, 0);
    return rebuild(c);
}

int qd_kernel_kind(const qd_ctx* c, int has_len) {
    if (!c || !c->have_table) return QD_ERR_STATE;
    return pick_kernel(c, has_len != 0);
}

int qd_set_option(qd_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return QD_ERR_INVALID;
    if (!strcmp(name, "fast_workgroups_per_cu")) {
        if (value < 0 || value > 4096) return fail(c, QD_ERR_INVALID, "fast_workgroups_per_cu must be 0..4096");
        c->opt_wg_per_cu = (int)value;
        return QD_OK;
    }
    if (!strcmp(name, "fast_block")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024)
            return fail(c, QD_ERR_INVALID, "fast_block must be 0, 256, 512 or 1024");
        c->opt_block = (int)value;
        return QD_OK;
    }
    if (!strcmp(name, "slot_factor")) {  // open-addressing slots per barcode of the fast kernel's table (before rounding up to a power of two)
        if (value < 0 || value > 16) return fail(c, QD_ERR_INVALID, "slot_factor must be 0 (automatic) .. 16");
        c->opt_slot_factor = (int)value;
        return c->have_table ? rebuild(c) : QD_OK;
    }
    if (!strcmp(name, "mol_strips")) {
        c->opt_mol_strips = value != 0;
        return c->have_table ? rebuild(c) : QD_OK;
    }
    if (!strcmp(name, "force_generic")) {
        c->opt_force_generic = value != 0;
        return QD_OK;
    }
    if (!strcmp(name, "kernel")) {
        if (value < 0 || value > 2) return fail(c, QD_ERR_INVALID, "kernel must be 0 (automatic), 1 (fast) or 2 (generic)");
        c->opt_kernel = (int)value;
        return QD_OK;
    }
    if (!strcmp(name, "fold_pairs")) {  // test knob: fold the 32-bit counter rows this often (default 2^32 - 1)
        if (value < 1 || value > (int64_t)0xFFFFFFFF) return fail(c, QD_ERR_INVALID, "fold_pairs must be 1..2^32-1");
        c->fold_limit = (uint64_t)value;
        return QD_OK;
    }
    return fail(c, QD_ERR_INVALID, std::string("unknown option ") + name);
}


int qd_demux_device(qd_ctx* c, int64_t n, const qd_rows* rows, uint16_t* codes, uint8_t* mol, void* stream) {
    if (!c || !rows || n < 0) return fail(c, QD_ERR_INVALID, "bad arguments");
    if (!c->have_plan || !c->have_table) return fail(c, QD_ERR_STATE, "qd_set_plan and qd_set_barcodes first");
    if (n == 0) return QD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return launch(c, n, rows, codes, mol, resolve_stream(c, stream));
}

int qd_demux_device_ragged(qd_ctx* c, int64_t n, const qd_rows* rows, uint16_t* codes, uint8_t* mol, int64_t n_short,
                           const uint32_t* short_idx, void* stream) {
    if (!c || !rows || n < 0 || n_short < 0 || (n_short > 0 && !short_idx)) return fail(c, QD_ERR_INVALID, "bad arguments");
    if (!c->have_plan || !c->have_table) return fail(c, QD_ERR_STATE, "qd_set_plan and qd_set_barcodes first");
    for (int k = 0; k < c->lay.n_streams; ++k)
        if (!rows->len[k]) return fail(c, QD_ERR_INVALID, "qd_demux_device_ragged needs the len rows of every stream");
    if (n == 0) return QD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return launch(c, n, rows, codes, mol, resolve_stream(c, stream), n_short, short_idx);
}

int qd_synchronize(qd_ctx* c) {
    if (!c) return QD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_all(c));
    return QD_OK;
}

// counters of this context summed on its device: d_counts[0..2S] + TOTAL at d_counts[cnt_stride], on c->stream
static int counts_to_device(qd_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, join_into_own_stream(c));  // the reduce runs behind this context's launches, whatever stream they used
    const uint32_t ncnt = (uint32_t)(2 * c->S + 1);
    hipError_t e = qd_launch_reduce(c->d_partial, c->partial_rows, c->cnt_stride, ncnt, c->d_acc, c->d_counts, c->stream);
    if (e != hipSuccess) return fail(c, QD_ERR_HIP, std::string("reduce launch: ") + hipGetErrorString(e));
    HIPCHK(c, hipMemcpyAsync(c->d_counts + c->cnt_stride, &c->total_pairs, 8, hipMemcpyHostToDevice, c->stream));
    return QD_OK;
}

// h = device layout [cnt_stride + 1] -> the ABI's vector [2S + 4]
static void compose_counts(const qd_ctx* c, const u64* h, uint64_t* out) {
    uint64_t pass = 0, failq = 0;
    for (int i = 0; i < c->S; ++i) {
        out[4 + 2 * i] = h[2 * i];
        out[5 + 2 * i] = h[2 * i + 1];
        pass += h[2 * i];
        failq += h[2 * i + 1];
    }
    out[0] = h[c->cnt_stride];  // TOTAL: pairs submitted (Sample.py:62)
    out[1] = pass;
    out[2] = failq;
    out[3] = h[2 * c->S];       // UNDETERMINED: counted on the device, not derived
}

int qd_get_counts(qd_ctx* c, uint64_t* out, int32_t n_values) {
    if (!c || !out) return QD_ERR_INVALID;
    if (!c->have_table) return fail(c, QD_ERR_STATE, "qd_set_barcodes first");
    if (n_values != 2 * c->S + 4) return fail(c, QD_ERR_INVALID, "n_values must be 2*S+4");
    const int r = counts_to_device(c);
    if (r != QD_OK) return r;
    std::vector<u64> h((size_t)c->cnt_stride + 1);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->d_counts, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    compose_counts(c, h.data(), out);
    return QD_OK;
}

// Counters another context of this process gathered (qd_get_counts layout) join this context's 64-bit totals: what
// a process with several contexts on one device does before qd_reduce_counts, whose communicator holds one of them.
int qd_add_counts(qd_ctx* c, const uint64_t* counts, int32_t n_values) {
    if (!c || !counts) return QD_ERR_INVALID;
    if (!c->have_table) return fail(c, QD_ERR_STATE, "qd_set_barcodes first");
    if (n_values != 2 * c->S + 4) return fail(c, QD_ERR_INVALID, "n_values must be 2*S+4");
    uint64_t pass = 0, failq = 0;
    for (int i = 0; i < c->S; ++i) {
        pass += counts[4 + 2 * i];
        failq += counts[5 + 2 * i];
    }
    if (pass != counts[1] || failq != counts[2] || counts[0] != pass + failq + counts[3])
        return fail(c, QD_ERR_INVALID, "counter vector is not self-consistent (TOTAL = PASS + FAIL + UNDETERMINED over the samples)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_all(c));  // demux_fixup adds to d_acc on the device: nothing of this context may be in flight
    std::vector<u64> h((size_t)c->cnt_stride);
    HIPCHK(c, hipMemcpy(h.data(), c->d_acc, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2 * c->S; ++i) h[(size_t)i] += counts[4 + i];
    h[(size_t)(2 * c->S)] += counts[3];
    HIPCHK(c, hipMemcpy(c->d_acc, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    c->total_pairs += counts[0];
    return QD_OK;
}

int qd_reset_counts(qd_ctx* c) {
    if (!c) return QD_ERR_INVALID;
    if (!c->have_table) return QD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, join_into_own_stream(c));
    HIPCHK(c, hipMemsetAsync(c->d_partial, 0, (size_t)c->partial_rows * c->cnt_stride * sizeof(qd_row_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_acc, 0, (size_t)c->cnt_stride * 8, c->stream));
    HIPCHK(c, track(c, c->stream));  // later launches on other streams are not ordered behind this: wait here
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->total_pairs = 0;
    c->pairs_in_rows = 0;
    return QD_OK;
}

// ---- pinned slots ---------------------------------------------------------------------------------
static size_t carve(qd_slot_buffers& v, uint8_t* base, const qd_layout& L, int64_t n) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    for (int k = 0; k < 2; ++k) {
        const bool used = k < L.n_streams;
        v.seq[k] = used ? take((size_t)n * L.seq_stride[k]) : nullptr;
        v.qual[k] = used ? take((size_t)n * L.qual_stride[k]) : nullptr;
        v.len[k] = used ? take((size_t)n) : nullptr;
    }
    v.codes = (uint16_t*)take((size_t)n * 2);
    v.mol = L.mol_width ? take((size_t)n * L.mol_width) : nullptr;
    v.max_pairs = n;
    v.short_cap = n / 8 + 64;
    for (int k = 0; k < 2; ++k) v.short_idx[k] = (k < L.n_streams) ? (uint32_t*)take((size_t)v.short_cap * 4) : nullptr;
    return off;
}

int qd_slots_create(qd_ctx* c, int32_t n_slots, int64_t max_pairs) {
    if (!c || n_slots < 1 || n_slots > 64 || max_pairs < 1) return fail(c, QD_ERR_INVALID, "bad slot arguments");
    if (!c->have_plan) return fail(c, QD_ERR_STATE, "qd_set_plan first");
    if (!c->slots.empty()) return fail(c, QD_ERR_STATE, "slots already exist");
    HIPCHK(c, hipSetDevice(c->device));
    qd_slot_buffers probe{};
    const size_t bytes = carve(probe, nullptr, c->lay, max_pairs);
    c->slots.resize((size_t)n_slots);
    c->slot_pairs = max_pairs;
    HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    for (auto& s : c->slots) {
        HIPCHK(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void**)&s.h_base, bytes, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void**)&s.d_base, bytes));
        carve(s.h, s.h_base, c->lay, max_pairs);
        carve(s.d, s.d_base, c->lay, max_pairs);
        HIPCHK(c, hipHostMalloc((void**)&s.h_short, (size_t)s.h.short_cap * 12, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void**)&s.d_short, (size_t)s.h.short_cap * 12));
    }
    return QD_OK;
}

int qd_slots_destroy(qd_ctx* c) {
    if (!c) return QD_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (auto& s : c->slots) {
        if (s.stream) {
            (void)hipStreamSynchronize(s.stream);
            forget_stream(c, s.stream);
            (void)hipStreamDestroy(s.stream);
        }
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        if (s.h_base) (void)hipHostFree(s.h_base);
        if (s.d_base) (void)hipFree(s.d_base);
        if (s.h_short) (void)hipHostFree(s.h_short);
        if (s.d_short) (void)hipFree(s.d_short);
    }
    c->slots.clear();
    if (c->up_stream) {
        (void)hipStreamSynchronize(c->up_stream);
        (void)hipStreamDestroy(c->up_stream);
        c->up_stream = nullptr;
    }
    return QD_OK;
}

int qd_slot_get(qd_ctx* c, int32_t slot, qd_slot_buffers* out) {
    if (!c || !out || slot < 0 || slot >= (int)c->slots.size()) return fail(c, QD_ERR_INVALID, "bad slot");
    *out = c->slots[(size_t)slot].h;
    return QD_OK;
}

// n_short: NULL = no exception lists (has_len decides between the fast and the generic kernel)
static int submit_impl(qd_ctx* c, int32_t slot, int64_t n, int32_t has_len, const int64_t* n_short) {
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, QD_ERR_INVALID, "bad slot");
    if (n < 0 || n > c->slot_pairs) return fail(c, QD_ERR_INVALID, "n_pairs exceeds the slot capacity");
    if (!c->have_table) return fail(c, QD_ERR_STATE, "qd_set_barcodes first");
    Slot& s = c->slots[(size_t)slot];
    if (s.busy) return fail(c, QD_ERR_STATE, "slot already submitted; qd_wait it first");
    HIPCHK(c, hipSetDevice(c->device));
    s.busy = true;
    if (n == 0) return QD_OK;
    const qd_layout& L = c->lay;
    // exception lists of the streams -> one ascending list without duplicates (a pair short in both
    // index reads is redone once), indices beyond the batch dropped
    int64_t m = -1;
    if (n_short) {
        bool listed = true;
        for (int k = 0; k < L.n_streams; ++k) listed = listed && n_short[k] >= 0 && n_short[k] <= s.h.short_cap;
        if (listed) {
            const uint32_t* a = s.h.short_idx[0];
            const uint32_t* b = L.n_streams > 1 ? s.h.short_idx[1] : nullptr;
            int64_t na = n_short[0], nb = b ? n_short[1] : 0, i = 0, j = 0;
            m = 0;
            while (i < na || j < nb) {
                uint32_t v;
                if (j >= nb || (i < na && a[i] <= b[j])) {
                    v = a[i++];
                    if (j < nb && b[j] == v) ++j;
                } else {
                    v = b[j++];
                }
                if ((int64_t)v < n) s.h_short[m++] = v;
            }
            if (m == 0) has_len = 0;  // every short read lies beyond the batch
        }
    }
    // a listed minority on a fast-eligible plan: only their lengths travel, not the len rows (the generic
    // kernel, which reads a length per pair, gets the rows)
    const bool listed = has_len && m > 0 && m <= n / 2 && pick_kernel(c, false) == K_FAST;
    qd_rows rows{};
    for (int k = 0; k < L.n_streams; ++k) {
        HIPCHK(c, hipMemcpyAsync(s.d.seq[k], s.h.seq[k], (size_t)n * L.seq_stride[k], hipMemcpyHostToDevice, c->up_stream));
        HIPCHK(c, hipMemcpyAsync(s.d.qual[k], s.h.qual[k], (size_t)n * L.qual_stride[k], hipMemcpyHostToDevice, c->up_stream));
        rows.seq[k] = s.d.seq[k];
        rows.qual[k] = s.d.qual[k];
        if (has_len && !listed) {
            HIPCHK(c, hipMemcpyAsync(s.d.len[k], s.h.len[k], (size_t)n, hipMemcpyHostToDevice, c->up_stream));
            rows.len[k] = s.d.len[k];
        }
    }
    const uint8_t* dlen[2] = {nullptr, nullptr};
    if (listed) {
        const size_t cap2 = (size_t)s.h.short_cap * 2;
        uint8_t* hl = reinterpret_cast<uint8_t*>(s.h_short + cap2);
        uint8_t* dl = reinterpret_cast<uint8_t*>(s.d_short + cap2);
        for (int k = 0; k < L.n_streams; ++k) {
            for (int64_t i = 0; i < m; ++i) hl[k * cap2 + (size_t)i] = s.h.len[k][s.h_short[i]];
            dlen[k] = dl + k * cap2;
        }
        HIPCHK(c, hipMemcpyAsync(s.d_short, s.h_short, (size_t)m * 4, hipMemcpyHostToDevice, c->up_stream));
        for (int k = 0; k < L.n_streams; ++k)
            HIPCHK(c, hipMemcpyAsync(dl + k * cap2, hl + k * cap2, (size_t)m, hipMemcpyHostToDevice, c->up_stream));
    }
    HIPCHK(c, hipEventRecord(s.uploaded, c->up_stream));
    HIPCHK(c, hipStreamWaitEvent(s.stream, s.uploaded, 0));
    const int r = launch(c, n, &rows, s.d.codes, s.d.mol, s.stream, listed ? m : -1, s.d_short, listed ? dlen : nullptr);
    if (r != QD_OK) return r;
    HIPCHK(c, hipMemcpyAsync(s.h.codes, s.d.codes, (size_t)n * 2, hipMemcpyDeviceToHost, s.stream));
    if (L.mol_width)
        HIPCHK(c, hipMemcpyAsync(s.h.mol, s.d.mol, (size_t)n * L.mol_width, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(c, track(c, s.stream));
    return QD_OK;
}

int qd_submit(qd_ctx* c, int32_t slot, int64_t n, int32_t has_len) { return submit_impl(c, slot, n, has_len, nullptr); }

int qd_submit_ragged(qd_ctx* c, int32_t slot, int64_t n, const int64_t n_short[2]) {
    if (!n_short) return fail(c, QD_ERR_INVALID, "n_short is NULL");
    return submit_impl(c, slot, n, 1, n_short);
}

int qd_wait(qd_ctx* c, int32_t slot) {
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, QD_ERR_INVALID, "bad slot");
    Slot& s = c->slots[(size_t)slot];
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(s.stream));
    s.busy = false;
    return QD_OK;
}

}  // extern "C"

// ---- multi-GPU: the one exchange of the path --------------------------------------------------------
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool ok = false;
    Rccl() {
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) {
            why = std::string("librccl.so.1: ") + dlerror();
            return;
        }
#define QD_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(handle, name))
        QD_SYM(GetUniqueId, "ncclGetUniqueId");
        QD_SYM(CommInitRank, "ncclCommInitRank");
        QD_SYM(CommInitAll, "ncclCommInitAll");
        QD_SYM(CommDestroy, "ncclCommDestroy");
        QD_SYM(AllReduce, "ncclAllReduce");
        QD_SYM(GroupStart, "ncclGroupStart");
        QD_SYM(GroupEnd, "ncclGroupEnd");
        QD_SYM(GetErrorString, "ncclGetErrorString");
#undef QD_SYM
        ok = GetUniqueId && CommInitRank && CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd && GetErrorString;
        if (!ok) why = "librccl.so.1 lacks an expected symbol";
    }
};
Rccl& rccl() {
    static Rccl R;
    return R;
}
thread_local std::string g_comm_error;
int comm_fail(int code, const std::string& msg) {
    g_comm_error = msg;
    return code;
}
}  // namespace

struct qd_comm {
    std::vector<qd_ctx*> ctxs;       // this process's member contexts (one per local device)
    std::vector<ncclComm_t> comms;   // their communicators
    int world = 0, rank0 = 0;        // ranks of the whole communicator; rank of ctxs[0]
};

extern "C" {

const char* qd_comm_last_error(void) { return g_comm_error.c_str(); }

int qd_comm_unique_id(uint8_t id[QD_UNIQUE_ID_BYTES]) {
    if (!id) return comm_fail(QD_ERR_INVALID, "id is NULL");
    Rccl& R = rccl();
    if (!R.ok) return comm_fail(QD_ERR_HIP, R.why);
    static_assert(QD_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");
    ncclUniqueId u;
    const ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) return comm_fail(QD_ERR_HIP, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
    memcpy(id, &u, sizeof u);
    return QD_OK;
}

int qd_comm_create_local(qd_ctx* const* ctxs, int32_t n, qd_comm** out) {
    if (!ctxs || n < 1 || !out) return comm_fail(QD_ERR_INVALID, "bad arguments");
    *out = nullptr;
    Rccl& R = rccl();
    if (!R.ok) return comm_fail(QD_ERR_HIP, R.why);
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return comm_fail(QD_ERR_INVALID, "NULL context");
        for (int d : devs)
            if (d == ctxs[i]->device) return comm_fail(QD_ERR_INVALID, "two member contexts share a device (RCCL wants one rank per device)");
        devs.push_back(ctxs[i]->device);
    }
    std::unique_ptr<qd_comm> cm(new qd_comm());
    cm->ctxs.assign(ctxs, ctxs + n);
    cm->comms.resize((size_t)n);
    cm->world = n;
    const ncclResult_t r = R.CommInitAll(cm->comms.data(), n, devs.data());  // one process, the n local devices
    if (r != ncclSuccess) return comm_fail(QD_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r));
    *out = cm.release();
    return QD_OK;
}

int qd_comm_create_rank(qd_ctx* ctx, int32_t world, int32_t rank, const uint8_t id[QD_UNIQUE_ID_BYTES], qd_comm** out) {
    if (!ctx || world < 1 || rank < 0 || rank >= world || !id || !out) return comm_fail(QD_ERR_INVALID, "bad arguments");
    *out = nullptr;
    Rccl& R = rccl();
    if (!R.ok) return comm_fail(QD_ERR_HIP, R.why);
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(QD_ERR_HIP, "hipSetDevice failed");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    std::unique_ptr<qd_comm> cm(new qd_comm());
    cm->ctxs.push_back(ctx);
    cm->comms.resize(1);
    cm->world = world;
    cm->rank0 = rank;
    const ncclResult_t r = R.CommInitRank(&cm->comms[0], world, u, rank);  // one process per GPU
    if (r != ncclSuccess) return comm_fail(QD_ERR_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(r));
    *out = cm.release();
    return QD_OK;
}

int qd_comm_world(const qd_comm* cm) { return cm ? cm->world : QD_ERR_INVALID; }

int qd_reduce_counts(qd_comm* cm, uint64_t* out, int32_t n_values) {
    if (!cm || !out) return comm_fail(QD_ERR_INVALID, "bad arguments");
    Rccl& R = rccl();
    qd_ctx* c0 = cm->ctxs[0];
    if (!c0->have_table) return comm_fail(QD_ERR_STATE, "qd_set_barcodes first");
    if (n_values != 2 * c0->S + 4) return comm_fail(QD_ERR_INVALID, "n_values must be 2*S+4");
    for (qd_ctx* c : cm->ctxs) {
        if (!c->have_table || c->S != c0->S) return comm_fail(QD_ERR_STATE, "member contexts hold different sample tables");
        const int r = counts_to_device(c);  // every member's counters + TOTAL, summed on its own device
        if (r != QD_OK) return comm_fail(r, c->err);
    }
    // one all-reduce (sum, uint64) of 2S+1 counters + TOTAL over xGMI; latency-bound (<= 24.6 KB at S = 1536)
    const size_t count = (size_t)c0->cnt_stride + 1;
    ncclResult_t r = R.GroupStart();
    for (size_t i = 0; r == ncclSuccess && i < cm->ctxs.size(); ++i) {
        qd_ctx* c = cm->ctxs[i];
        if (hipSetDevice(c->device) != hipSuccess) return comm_fail(QD_ERR_HIP, "hipSetDevice failed");
        r = R.AllReduce(c->d_counts, c->d_total, count, ncclUint64, ncclSum, cm->comms[i], c->stream);
    }
    const ncclResult_t r2 = R.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess)
        return comm_fail(QD_ERR_HIP, std::string("ncclAllReduce: ") + R.GetErrorString(r != ncclSuccess ? r : r2));
    std::vector<u64> h(count);
    if (hipSetDevice(c0->device) != hipSuccess) return comm_fail(QD_ERR_HIP, "hipSetDevice failed");
    hipError_t e = hipMemcpyAsync(h.data(), c0->d_total, count * 8, hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
    for (size_t i = 1; e == hipSuccess && i < cm->ctxs.size(); ++i) {  // every member's collective has completed on return
        if (hipSetDevice(cm->ctxs[i]->device) != hipSuccess) return comm_fail(QD_ERR_HIP, "hipSetDevice failed");
        e = hipStreamSynchronize(cm->ctxs[i]->stream);
    }
    if (e != hipSuccess) return comm_fail(QD_ERR_HIP, std::string("count reduce: ") + hipGetErrorString(e));
    compose_counts(c0, h.data(), out);
    return QD_OK;
}

int qd_comm_destroy(qd_comm* cm) {
    if (!cm) return QD_OK;
    Rccl& R = rccl();
    for (size_t i = 0; i < cm->comms.size(); ++i) {
        (void)hipSetDevice(cm->ctxs[i]->device);
        if (R.ok && cm->comms[i]) (void)R.CommDestroy(cm->comms[i]);
    }
    delete cm;
    return QD_OK;
}

}  // extern "C"

// ---- BGZF inflate on the device (include/quade_hip.h; kernel: quade_inflate.hip) ------------------------------
uint32_t qd_io_crc32(const uint8_t* p, size_t n);  // quade_io.cpp: libdeflate's when loaded, else zlib's
uint32_t qd_crc32_combine_host(uint32_t crc1, uint32_t crc2, uint64_t len2);  // quade_io.cpp (zlib's)

struct qd_inflater {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // blocking-sync event: the calling thread sleeps while the device works
    std::string err;
    // grow-only staging: pinned host + device, for the compressed run, its text, the block table and the states
    uint8_t *h_comp = nullptr, *d_comp = nullptr, *h_out = nullptr, *d_out = nullptr;
    size_t cap_comp = 0, cap_out = 0;
    qd_inflate_block *h_blk = nullptr, *d_blk = nullptr;
    int32_t *h_st = nullptr, *d_st = nullptr;
    size_t cap_blk = 0, cap_st = 0;
    std::vector<uint32_t> crc;
    // which kernel: 2 (the default) = 512 lanes per block, 1 = one wave per block (QUADE_INFLATE_FORM / qd_inflater_set_form); the second form's match lists
    int form = 3;
    unsigned long long* d_matches = nullptr;
    size_t cap_matches = 0;  // blocks the scratch holds
    // the third form (3: one lane decodes a block's symbols once, a workgroup resolves its tokens -- quade_inflate3.hip): its scratch
    void* d_scratch3 = nullptr;
    size_t cap_scratch3 = 0;  // bytes
};

namespace {
int inf_fail(qd_inflater* f, int code, const std::string& msg) {
    if (f) f->err = msg;
    return code;
}
#define INFCHK(f, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return inf_fail((f), QD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t grow_pair(T*& h, T*& d, size_t& cap, size_t need) {
    if (need <= cap) return hipSuccess;
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    h = nullptr;
    d = nullptr;
    cap = 0;
    const size_t n = need + need / 4 + 4096;
    hipError_t e = hipHostMalloc((void**)&h, n * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    e = hipMalloc((void**)&d, n * sizeof(T));
    if (e != hipSuccess) return e;
    cap = n;
    return hipSuccess;
}
thread_local std::string g_inflater_error;

// Waits for an event without holding a core: hipEventSynchronize spins on this runtime even for events made with
// hipEventBlockingSync (measured: the reader's device lanes burnt 0.95 core-seconds per M pairs waiting for their kernels,
// profiles/r03_e2e_16m_level1_stages_device_inflate.txt).  A few immediate polls (short kernels), then naps of 100 us.
hipError_t wait_event_napping(hipEvent_t ev) {
    static const int nap_us = [] {
        const char* e = getenv("QUADE_NAP_US");  // measurement knob
        const int v = e && *e ? atoi(e) : 100;
        return v < 10 ? 10 : (v > 5000 ? 5000 : v);
    }();
    for (int i = 0; i < 64; ++i) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
    }
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        usleep(nap_us);
    }
}
}  // namespace

extern "C" {

int qd_inflater_create(int device_id, qd_inflater** out) {
    if (!out) return QD_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) {
        g_inflater_error = "no such HIP device";
        return QD_ERR_NO_DEVICE;
    }
    qd_inflater* f = new qd_inflater();
    f->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&f->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
        g_inflater_error = "hipStreamCreate failed";
        delete f;
        return QD_ERR_HIP;
    }
    if (const char* e = getenv("QUADE_INFLATE_FORM")) f->form = atoi(e) == 1 ? 1 : (atoi(e) == 2 ? 2 : 3);
    *out = f;
    return QD_OK;
}

const char* qd_inflater_last_error(const qd_inflater* f) { return f ? f->err.c_str() : g_inflater_error.c_str(); }

int qd_inflater_destroy(qd_inflater* f) {
    if (!f) return QD_OK;
    (void)hipSetDevice(f->device);
    if (f->stream) {
        (void)hipStreamSynchronize(f->stream);
        (void)hipStreamDestroy(f->stream);
    }
    if (f->done) (void)hipEventDestroy(f->done);
    if (f->h_comp) (void)hipHostFree(f->h_comp);
    if (f->d_comp) (void)hipFree(f->d_comp);
    if (f->h_out) (void)hipHostFree(f->h_out);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->h_blk) (void)hipHostFree(f->h_blk);
    if (f->d_blk) (void)hipFree(f->d_blk);
    if (f->h_st) (void)hipHostFree(f->h_st);
    if (f->d_st) (void)hipFree(f->d_st);
    if (f->d_matches) (void)hipFree(f->d_matches);
    if (f->d_scratch3) (void)hipFree(f->d_scratch3);
    delete f;
    return QD_OK;
}

int qd_inflater_set_form(qd_inflater* f, int32_t form) {
    if (!f || form < 1 || form > 3) return QD_ERR_INVALID;
    f->form = form;
    return QD_OK;
}

// page-locked host memory for callers that want the text to land in their buffer without a staging copy
void* qd_pinned_alloc(int64_t bytes) {
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void qd_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

static int inflater_run(qd_inflater* f, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len, int32_t* bad_block,
                        bool out_pinned);
int qd_inflater_run(qd_inflater* f, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len, int32_t* bad_block) {
    return inflater_run(f, comp, comp_len, out, out_len, bad_block, false);
}
int qd_inflater_run_pinned(qd_inflater* f, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len, int32_t* bad_block) {
    return inflater_run(f, comp, comp_len, out, out_len, bad_block, true);
}

static int inflater_run(qd_inflater* f, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len, int32_t* bad_block,
                        bool out_pinned) {
    if (bad_block) *bad_block = -1;
    if (!f || !comp || comp_len < 0 || out_len < 0 || (out_len > 0 && !out)) return inf_fail(f, QD_ERR_INVALID, "bad arguments");
    if (comp_len > (int64_t)0xFFFF0000ll || out_len > (int64_t)0xFFFF0000ll) return inf_fail(f, QD_ERR_INVALID, "run larger than 4 GiB");
    // 1. walk the block headers: gzip member with the 'BC' extra subfield = total block size - 1
    std::vector<qd_inflate_block> blk;
    f->crc.clear();
    size_t pos = 0, opos = 0;
    while (pos < (size_t)comp_len) {
        const uint8_t* p = comp + pos;
        const size_t avail = (size_t)comp_len - pos;
        size_t bsize = 0, xlen = 0;
        if (avail >= 18 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 4)) {
            xlen = p[10] | ((size_t)p[11] << 8);
            if (avail >= 12 + xlen)
                for (size_t o = 12; o + 4 <= 12 + xlen;) {
                    const size_t slen = p[o + 2] | ((size_t)p[o + 3] << 8);
                    if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
                        bsize = (size_t)(p[o + 4] | (p[o + 5] << 8)) + 1;
                        break;
                    }
                    o += 4 + slen;
                }
        }
        if (!bsize || bsize > avail || bsize < 12 + xlen + 8 || (p[3] & ~4))  // other gzip flags (name, comment, hcrc): not bgzip's
            return inf_fail(f, QD_ERR_FORMAT, "not a run of whole BGZF blocks");
        const uint8_t* tr = p + bsize - 8;
        const uint32_t crc = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        const uint32_t isize = tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize > (64u << 10) || opos + isize > (size_t)out_len) return inf_fail(f, QD_ERR_FORMAT, "BGZF block sizes do not add up");
        blk.push_back(qd_inflate_block{(uint32_t)(pos + 12 + xlen), (uint32_t)(bsize - 12 - xlen - 8), (uint32_t)opos, isize});
        f->crc.push_back(crc);
        pos += bsize;
        opos += isize;
    }
    if (opos != (size_t)out_len) return inf_fail(f, QD_ERR_FORMAT, "BGZF block sizes do not add up");
    if (blk.empty()) return QD_OK;
    // 2. stage, inflate one block per lane, fetch
    INFCHK(f, hipSetDevice(f->device));
    INFCHK(f, grow_pair(f->h_comp, f->d_comp, f->cap_comp, (size_t)comp_len));
    INFCHK(f, grow_pair(f->h_out, f->d_out, f->cap_out, (size_t)out_len + 16));
    uint8_t* const text = out_pinned ? out : f->h_out;  // where the D2H copy lands
    INFCHK(f, grow_pair(f->h_blk, f->d_blk, f->cap_blk, blk.size()));
    INFCHK(f, grow_pair(f->h_st, f->d_st, f->cap_st, blk.size()));
    memcpy(f->h_comp, comp, (size_t)comp_len);
    memcpy(f->h_blk, blk.data(), blk.size() * sizeof(qd_inflate_block));
    INFCHK(f, hipMemcpyAsync(f->d_comp, f->h_comp, (size_t)comp_len, hipMemcpyHostToDevice, f->stream));
    INFCHK(f, hipMemcpyAsync(f->d_blk, f->h_blk, blk.size() * sizeof(qd_inflate_block), hipMemcpyHostToDevice, f->stream));
    uint32_t longest = 0;
    for (const qd_inflate_block& bk : blk) longest = std::max(longest, bk.in_len);
    bool form3_ran = false;
    if (f->form == 3) {
        const size_t need = qd_inflate3_scratch_bytes((uint32_t)blk.size());
        if (need > f->cap_scratch3) {
            if (f->d_scratch3) (void)hipFree(f->d_scratch3);
            f->d_scratch3 = nullptr;
            f->cap_scratch3 = 0;
            INFCHK(f, hipMalloc(&f->d_scratch3, need + need / 4));
            f->cap_scratch3 = need + need / 4;
        }
        INFCHK(f, qd_launch_inflate3(f->d_comp, (size_t)comp_len, f->d_blk, (uint32_t)blk.size(), f->d_out, f->d_st, f->d_scratch3, f->stream));
        form3_ran = true;
    } else if (f->form == 2 && qd_inflate2_lds(longest) <= 160 * 1024) {  // (a run with a payload beyond ~52 KB -- stored blocks -- takes the first form)
        if (blk.size() > f->cap_matches) {
            if (f->d_matches) (void)hipFree(f->d_matches);
            f->d_matches = nullptr;
            f->cap_matches = 0;
            const size_t nb = blk.size() + blk.size() / 4 + 16;
            INFCHK(f, hipMalloc((void**)&f->d_matches, nb * (size_t)QD_INFLATE_MATCHES_PER_BLOCK * 8));
            f->cap_matches = nb;
        }
        static const bool debug_rounds = getenv("QUADE_INFLATE_DEBUG") != nullptr;  // measurement: rounds per block on stderr
        uint32_t* d_rounds = nullptr;
        if (debug_rounds) INFCHK(f, hipMalloc((void**)&d_rounds, blk.size() * 32));
        INFCHK(f, qd_launch_inflate2(f->d_comp, f->d_blk, (uint32_t)blk.size(), f->d_out, f->d_st, f->d_matches, QD_INFLATE_MATCHES_PER_BLOCK, longest,
                                     f->stream, d_rounds));
        if (debug_rounds) {
            std::vector<uint32_t> r(blk.size() * 8);
            INFCHK(f, hipMemcpy(r.data(), d_rounds, blk.size() * 32, hipMemcpyDeviceToHost));
            (void)hipFree(d_rounds);
            uint64_t sum = 0, db = 0, tk[8] = {0};
            uint32_t mx = 0, over8 = 0;
            for (size_t q = 0; q < blk.size(); ++q) {
                const uint32_t v = r[8 * q];
                sum += v & 0xFFFF;
                db += v >> 16;
                mx = std::max(mx, v & 0xFFFF);
                over8 += (v & 0xFFFF) > 8;
                for (int k = 1; k < 8; ++k) tk[k] += r[8 * q + k];
            }
            const double us = 0.01 / (double)blk.size();  // 100 MHz ticks -> microseconds per block
            fprintf(stderr, "[inflate form 2] %zu blocks: %.2f rounds and %.2f deflate blocks per block, most %u, %u blocks over 8 rounds; us per block: stage %.0f "
                            "header+tables %.0f rounds %.0f scan+write %.0f matches %.0f flush %.0f; %.0f matches per block\n", blk.size(), (double)sum / blk.size(),
                    (double)db / blk.size(), mx, over8, tk[1] * us, tk[2] * us, tk[3] * us, tk[4] * us, tk[5] * us, tk[6] * us, (double)tk[7] / blk.size());
        }
    } else {
        INFCHK(f, qd_launch_inflate(f->d_comp, f->d_blk, (uint32_t)blk.size(), f->d_out, f->d_st, f->stream));
    }
    INFCHK(f, hipMemcpyAsync(f->h_st, f->d_st, blk.size() * 4, hipMemcpyDeviceToHost, f->stream));
    if (form3_ran) {
        // blocks whose Huffman codes hold more long symbols than a lane's table takes (byte soup, not fastq) go through the second
        // form -- or the first, when their payload leaves no room for it: a second launch over just those blocks
        INFCHK(f, hipEventRecord(f->done, f->stream));
        INFCHK(f, wait_event_napping(f->done));
        std::vector<qd_inflate_block> redo;
        std::vector<size_t> redo_at;
        uint32_t redo_longest = 0;
        for (size_t i = 0; i < blk.size(); ++i)
            if (f->h_st[i] == QD_INFLATE_TABLE_SPACE) {
                redo.push_back(blk[i]);
                redo_at.push_back(i);
                redo_longest = std::max(redo_longest, blk[i].in_len);
            }
        if (!redo.empty()) {
            qd_inflate_block* d_redo = nullptr;
            int32_t* d_rst = nullptr;
            INFCHK(f, hipMalloc((void**)&d_redo, redo.size() * sizeof(qd_inflate_block)));
            INFCHK(f, hipMalloc((void**)&d_rst, redo.size() * 4));
            INFCHK(f, hipMemcpy(d_redo, redo.data(), redo.size() * sizeof(qd_inflate_block), hipMemcpyHostToDevice));
            hipError_t e2;
            if (qd_inflate2_lds(redo_longest) <= 160 * 1024) {
                if (redo.size() > f->cap_matches) {
                    if (f->d_matches) (void)hipFree(f->d_matches);
                    f->d_matches = nullptr;
                    f->cap_matches = 0;
                    INFCHK(f, hipMalloc((void**)&f->d_matches, (redo.size() + 16) * (size_t)QD_INFLATE_MATCHES_PER_BLOCK * 8));
                    f->cap_matches = redo.size() + 16;
                }
                e2 = qd_launch_inflate2(f->d_comp, d_redo, (uint32_t)redo.size(), f->d_out, d_rst, f->d_matches, QD_INFLATE_MATCHES_PER_BLOCK, redo_longest, f->stream);
            } else {
                e2 = qd_launch_inflate(f->d_comp, d_redo, (uint32_t)redo.size(), f->d_out, d_rst, f->stream);
            }
            std::vector<int32_t> rst(redo.size());
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(rst.data(), d_rst, redo.size() * 4, hipMemcpyDeviceToHost, f->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(f->stream);
            (void)hipFree(d_redo);
            (void)hipFree(d_rst);
            INFCHK(f, e2);
            for (size_t k = 0; k < redo.size(); ++k) f->h_st[redo_at[k]] = rst[k];
        }
    }
    if (out_len) INFCHK(f, hipMemcpyAsync(text, f->d_out, (size_t)out_len, hipMemcpyDeviceToHost, f->stream));
    INFCHK(f, hipEventRecord(f->done, f->stream));
    INFCHK(f, wait_event_napping(f->done));
    // 3. every block: decoder status, then the CRC32 of its text
    for (size_t i = 0; i < blk.size(); ++i) {
        if (f->h_st[i] != 0 || qd_io_crc32(text + blk[i].out_off, blk[i].out_len) != f->crc[i]) {
            if (bad_block) *bad_block = (int32_t)i;
            char m[128];
            if (f->h_st[i])
                snprintf(m, sizeof m, "BGZF block %zu: did not inflate (decoder status %d)", i, (int)f->h_st[i]);
            else
                snprintf(m, sizeof m, "BGZF block %zu: CRC32 mismatch", i);
            return inf_fail(f, QD_ERR_FORMAT, m);
        }
    }
    if (!out_pinned) memcpy(out, f->h_out, (size_t)out_len);
    return QD_OK;
}

}  // extern "C"

// ---- Huffman-only gzip members on the device (include/quade_hip.h; kernel: quade_deflate.hip) -------------------------
struct qd_deflater {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::string err;
    // grow-only staging: pinned host + device, for the text, the members, the piece table and the member lengths
    uint8_t *h_text = nullptr, *d_text = nullptr, *h_out = nullptr, *d_out = nullptr;
    size_t cap_text = 0, cap_out = 0;
    qd_deflate_piece *h_pc = nullptr, *d_pc = nullptr;
    uint32_t *h_len = nullptr, *d_len = nullptr;
    size_t cap_pc = 0, cap_len = 0;
    // level 1 (LZ77 + dynamic Huffman, quade_deflate.hip): sub-block table, piece -> first sub-block, device scratch
    int level = -1;
    qd_lz_sub *h_sub = nullptr, *d_sub = nullptr;
    uint32_t *h_first = nullptr, *d_first = nullptr;
    size_t cap_sub = 0, cap_first = 0;
    uint32_t *d_tokens = nullptr, *d_sub_bytes = nullptr;
    uint8_t* d_sub_out = nullptr;
    size_t cap_scratch = 0;  // sub-blocks the three scratch arrays hold
};

namespace {
thread_local std::string g_deflater_error;
// The coder's launches are short (2 ms per 64 MB) and everything downstream waits for them: its stream gets the device's
// highest priority, so that its workgroups are placed before those of long-running kernels that share the GPU (the BGZF
// inflater's launches hold their CUs for ~16 ms).
hipError_t make_priority_stream(hipStream_t* st) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}
int def_fail(qd_deflater* f, int code, const std::string& msg) {
    if (f) f->err = msg;
    else g_deflater_error = msg;
    return code;
}
#define DEFCHK(f, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return def_fail((f), QD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
}  // namespace

extern "C" {

int64_t qd_huffman_member_bound(int64_t text_len) {
    // a code for 257 symbols never needs more than 9 bits per byte on average (the fixed 9-bit code is a candidate); the
    // 15-bit limit's repair and the header (149 bytes), end-of-block code and trailer ride in the slack
    return text_len < 0 ? -1 : ((text_len * 9 + 7) / 8 + text_len / 64 + 1024 + 3) & ~(int64_t)3;
}

int qd_deflater_create(int device_id, qd_deflater** out) {
    if (!out) return QD_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return def_fail(nullptr, QD_ERR_NO_DEVICE, "no such HIP device");
    qd_deflater* f = new qd_deflater();
    f->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || make_priority_stream(&f->stream) != hipSuccess ||
        hipEventCreateWithFlags(&f->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
        delete f;
        return def_fail(nullptr, QD_ERR_HIP, "hipStreamCreate failed");
    }
    *out = f;
    return QD_OK;
}

const char* qd_deflater_last_error(const qd_deflater* f) { return f ? f->err.c_str() : g_deflater_error.c_str(); }

int qd_deflater_destroy(qd_deflater* f) {
    if (!f) return QD_OK;
    (void)hipSetDevice(f->device);
    if (f->stream) {
        (void)hipStreamSynchronize(f->stream);
        (void)hipStreamDestroy(f->stream);
    }
    if (f->done) (void)hipEventDestroy(f->done);
    if (f->h_text) (void)hipHostFree(f->h_text);
    if (f->d_text) (void)hipFree(f->d_text);
    if (f->h_out) (void)hipHostFree(f->h_out);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->h_pc) (void)hipHostFree(f->h_pc);
    if (f->d_pc) (void)hipFree(f->d_pc);
    if (f->h_len) (void)hipHostFree(f->h_len);
    if (f->d_len) (void)hipFree(f->d_len);
    if (f->h_sub) (void)hipHostFree(f->h_sub);
    if (f->d_sub) (void)hipFree(f->d_sub);
    if (f->h_first) (void)hipHostFree(f->h_first);
    if (f->d_first) (void)hipFree(f->d_first);
    if (f->d_tokens) (void)hipFree(f->d_tokens);
    if (f->d_sub_bytes) (void)hipFree(f->d_sub_bytes);
    if (f->d_sub_out) (void)hipFree(f->d_sub_out);
    delete f;
    return QD_OK;
}

int qd_deflater_set_level(qd_deflater* f, int32_t level) {
    if (!f || (level != -1 && level != 1)) return def_fail(f, QD_ERR_INVALID, "the device makes members at gzip_level -1 (Huffman only) and 1 (LZ77 + Huffman)");
    f->level = level;
    return QD_OK;
}

int qd_deflater_run(qd_deflater* f, int32_t n_pieces, const uint8_t* const* text, const int64_t* text_len, const uint32_t* crc32,
                    int32_t text_pinned, uint8_t* out, int64_t out_stride, int64_t* member_len) {
    if (!f || n_pieces < 0 || (n_pieces && (!text || !text_len || !crc32 || !out || !member_len)) || out_stride < 64 || (out_stride & 3))
        return def_fail(f, QD_ERR_INVALID, "bad arguments");
    if (n_pieces == 0) return QD_OK;
    size_t total = 0;
    for (int i = 0; i < n_pieces; ++i) {
        if (text_len[i] < 0 || text_len[i] > (int64_t)0x7FFFFFFF || (text_len[i] && !text[i])) return def_fail(f, QD_ERR_INVALID, "bad piece");
        total += ((size_t)text_len[i] + 15) & ~(size_t)15;
    }
    DEFCHK(f, hipSetDevice(f->device));
    DEFCHK(f, grow_pair(f->h_pc, f->d_pc, f->cap_pc, (size_t)n_pieces));
    DEFCHK(f, grow_pair(f->h_len, f->d_len, f->cap_len, (size_t)n_pieces));
    DEFCHK(f, grow_pair(f->h_out, f->d_out, f->cap_out, (size_t)n_pieces * (size_t)out_stride));
    if (total + 16 > f->cap_text) {  // device text always; the pinned staging copy only for pageable callers
        if (f->h_text) (void)hipHostFree(f->h_text);
        if (f->d_text) (void)hipFree(f->d_text);
        f->h_text = f->d_text = nullptr;
        f->cap_text = 0;
        const size_t n = total + total / 4 + 4096;
        DEFCHK(f, hipMalloc((void**)&f->d_text, n));
        f->cap_text = n;
    }
    if (!text_pinned && !f->h_text) DEFCHK(f, hipHostMalloc((void**)&f->h_text, f->cap_text, hipHostMallocDefault));
    size_t at = 0;
    for (int i = 0; i < n_pieces; ++i) {
        f->h_pc[i] = qd_deflate_piece{(uint64_t)at, (uint32_t)text_len[i], crc32[i]};
        if (text_len[i]) {
            const uint8_t* src = text[i];
            if (!text_pinned) {
                memcpy(f->h_text + at, text[i], (size_t)text_len[i]);
                src = f->h_text + at;
            }
            DEFCHK(f, hipMemcpyAsync(f->d_text + at, src, (size_t)text_len[i], hipMemcpyHostToDevice, f->stream));
        }
        at += ((size_t)text_len[i] + 15) & ~(size_t)15;
    }
    DEFCHK(f, hipMemcpyAsync(f->d_pc, f->h_pc, (size_t)n_pieces * sizeof(qd_deflate_piece), hipMemcpyHostToDevice, f->stream));
    if (f->level == 1) {  // LZ77 + dynamic Huffman: one workgroup per 64 KiB sub-block, then the members are strung together
        size_t n_subs = 0;
        for (int i = 0; i < n_pieces; ++i) n_subs += ((size_t)text_len[i] + QD_LZ_SUB - 1) / QD_LZ_SUB;
        const int64_t sub_stride = qd_huffman_member_bound(QD_LZ_SUB);
        DEFCHK(f, grow_pair(f->h_sub, f->d_sub, f->cap_sub, n_subs + 1));
        DEFCHK(f, grow_pair(f->h_first, f->d_first, f->cap_first, (size_t)n_pieces + 1));
        if (n_subs > f->cap_scratch) {
            if (f->d_tokens) (void)hipFree(f->d_tokens);
            if (f->d_sub_bytes) (void)hipFree(f->d_sub_bytes);
            if (f->d_sub_out) (void)hipFree(f->d_sub_out);
            f->d_tokens = f->d_sub_bytes = nullptr;
            f->d_sub_out = nullptr;
            f->cap_scratch = 0;
            const size_t n = n_subs + n_subs / 4 + 16;
            DEFCHK(f, hipMalloc((void**)&f->d_tokens, n * (size_t)QD_LZ_SUB * 4));
            DEFCHK(f, hipMalloc((void**)&f->d_sub_bytes, n * 4));
            DEFCHK(f, hipMalloc((void**)&f->d_sub_out, n * (size_t)sub_stride));
            f->cap_scratch = n;
        }
        size_t js = 0, off = 0;
        for (int i = 0; i < n_pieces; ++i) {
            f->h_first[i] = (uint32_t)js;
            for (int64_t a = 0; a < text_len[i]; a += QD_LZ_SUB)
                f->h_sub[js++] = qd_lz_sub{(uint64_t)(off + (size_t)a), (uint32_t)std::min<int64_t>(QD_LZ_SUB, text_len[i] - a), (uint32_t)i};
            off += ((size_t)text_len[i] + 15) & ~(size_t)15;
        }
        f->h_first[n_pieces] = (uint32_t)js;
        if (js) DEFCHK(f, hipMemcpyAsync(f->d_sub, f->h_sub, js * sizeof(qd_lz_sub), hipMemcpyHostToDevice, f->stream));
        DEFCHK(f, hipMemcpyAsync(f->d_first, f->h_first, ((size_t)n_pieces + 1) * 4, hipMemcpyHostToDevice, f->stream));
        DEFCHK(f, qd_launch_lz(f->d_text, f->d_pc, (uint32_t)n_pieces, f->d_sub, f->d_first, (uint32_t)js, f->d_tokens, f->d_sub_out, sub_stride,
                               f->d_sub_bytes, f->d_out, out_stride, f->d_len, f->stream));
    } else {
        DEFCHK(f, qd_launch_huffman(f->d_text, f->d_pc, (uint32_t)n_pieces, f->d_out, out_stride, f->d_len, f->stream));
    }
    DEFCHK(f, hipMemcpyAsync(f->h_len, f->d_len, (size_t)n_pieces * 4, hipMemcpyDeviceToHost, f->stream));
    DEFCHK(f, hipEventRecord(f->done, f->stream));
    DEFCHK(f, wait_event_napping(f->done));
    for (int i = 0; i < n_pieces; ++i) {  // the used bytes of every member
        if (f->h_len[i] > (uint64_t)out_stride) return def_fail(f, QD_ERR_HIP, "member longer than its slot");
        if (f->h_len[i])
            DEFCHK(f, hipMemcpyAsync(f->h_out + (size_t)i * (size_t)out_stride, f->d_out + (size_t)i * (size_t)out_stride, f->h_len[i],
                                     hipMemcpyDeviceToHost, f->stream));
    }
    DEFCHK(f, hipEventRecord(f->done, f->stream));
    DEFCHK(f, wait_event_napping(f->done));
    for (int i = 0; i < n_pieces; ++i) {
        member_len[i] = f->h_len[i];  // 0: this member did not fit out_stride (the caller makes it itself)
        if (f->h_len[i]) memcpy(out + (size_t)i * (size_t)out_stride, f->h_out + (size_t)i * (size_t)out_stride, f->h_len[i]);
    }
    return QD_OK;
}

}  // extern "C"

// ---- a whole gzip file image through the device's gzip inflater (include/quade_hip.h: qd_dev_gunzip) -----------------------------------
namespace {
// bytes of a gzip member's header at p[0 .. n) (RFC 1952), 0 when it is none or does not end inside n
size_t gzip_header_bytes(const uint8_t* p, size_t n) {
    if (n < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
    const int flg = p[3];
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return 0;
        at += 2 + (p[at] | ((size_t)p[at + 1] << 8));
    }
    for (int bit : {8, 16})
        if (flg & bit) {
            while (at < n && p[at]) ++at;
            ++at;
        }
    if (flg & 2) at += 2;
    return at <= n ? at : 0;
}
}  // namespace

extern "C" int qd_dev_gunzip(int device_id, const uint8_t* gz, int64_t gz_len, uint8_t* out, int64_t out_cap, int64_t* out_len, int64_t step_bytes,
                             int64_t stretch_bytes, int64_t unit_text, int64_t* stats) {
    if (!gz || gz_len < 0 || !out_len || (out_cap > 0 && !out) || step_bytes < 4096) return QD_ERR_INVALID;
    *out_len = 0;
    if (hipSetDevice(device_id) != hipSuccess) return QD_ERR_NO_DEVICE;
    hipStream_t st = nullptr;
    uint8_t *d_gz = nullptr, *d_out = nullptr, *d_win = nullptr;
    int rc = QD_OK;
    qd_gz G;
    if (stretch_bytes > 0) G.stretch_bytes = (uint64_t)stretch_bytes;
    if (unit_text > 0) G.unit_text = (uint64_t)unit_text, G.unit_text_given = true;
    int64_t members = 0, steps_run = 0;
    auto done = [&](int code) {
        if (st) (void)hipStreamSynchronize(st);
        if (d_gz) (void)hipFree(d_gz);
        if (d_out) (void)hipFree(d_out);
        if (d_win) (void)hipFree(d_win);
        if (st) (void)hipStreamDestroy(st);
        if (stats) {
            const qd_gz_stats s = G.stats();
            stats[0] = members;
            stats[1] = steps_run;
            stats[2] = s.stretches;
            stats[3] = s.units;
            stats[4] = s.chain_retries;
            stats[5] = s.partial_last;
            stats[6] = s.plain_probes;
            stats[7] = 0;
        }
        return code;
    };
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return done(QD_ERR_HIP);
    if (hipMalloc((void**)&d_gz, (size_t)gz_len + 4096) != hipSuccess || hipMalloc((void**)&d_out, (size_t)std::max<int64_t>(out_cap, 1) + 4096) != hipSuccess ||
        hipMalloc((void**)&d_win, 32768) != hipSuccess)
        return done(QD_ERR_HIP);
    if (hipMemset(d_gz + gz_len, 0, 4096) != hipSuccess || (gz_len && hipMemcpy(d_gz, gz, (size_t)gz_len, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemset(d_win, 0, 32768) != hipSuccess)
        return done(QD_ERR_HIP);
    int64_t pos = 0, opos = 0;  // next member's header; text so far
    while (pos < gz_len && rc == QD_OK) {
        const size_t hb = gzip_header_bytes(gz + pos, (size_t)(gz_len - pos));
        if (!hb) {
            rc = QD_ERR_FORMAT;
            break;
        }
        uint64_t bit = 8 * (uint64_t)(pos + (int64_t)hb);  // absolute bit position of the next block header
        uint32_t crc = 0;
        uint64_t member_text = 0;
        qd_gz_step s{};
        s.carried = d_win;
        s.carried_valid = 0;
        int64_t step = step_bytes;
        bool ended = false;
        while (!ended) {
            const uint64_t byte0 = (bit >> 3) & ~(uint64_t)15;
            s.comp = d_gz + byte0;
            s.comp_bytes = (uint64_t)std::min<int64_t>(step, gz_len - (int64_t)byte0);
            s.bit_start = bit - 8 * byte0;
            s.at_end = (int64_t)byte0 + (int64_t)s.comp_bytes >= gz_len;
            if (G.decode(&s, 1, st) != hipSuccess) return done(QD_ERR_HIP);
            ++steps_run;
            if (s.failed) {
                rc = QD_ERR_FORMAT;
                break;
            }
            if (s.bit_next == s.bit_start && !s.member_end && !s.starved) {  // a block longer than the step: more input, if there is any
                if (s.at_end) {
                    rc = QD_ERR_FORMAT;
                    break;
                }
                step *= 2;
                continue;
            }
            if (opos + (int64_t)s.text_len > out_cap) {
                rc = QD_ERR_INVALID;
                break;
            }
            uint8_t* dst = d_out + opos;
            if (G.resolve(&s, 1, &dst, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess || G.finish(&s, 1) != hipSuccess) return done(QD_ERR_HIP);
            if (s.failed) {
                rc = QD_ERR_FORMAT;
                break;
            }
            crc = member_text ? qd_crc32_combine_host(crc, s.crc32, s.text_len) : s.crc32;
            if (s.text_len == 0 && member_text == 0) crc = 0;
            member_text += s.text_len;
            opos += (int64_t)s.text_len;
            bit = 8 * byte0 + s.bit_next;
            ended = s.member_end != 0;
        }
        if (rc != QD_OK) break;
        const int64_t tr = (int64_t)((bit + 7) >> 3);  // the trailer: CRC-32 and ISIZE of the member's text
        if (tr + 8 > gz_len) {
            rc = QD_ERR_FORMAT;
            break;
        }
        const uint32_t want_crc = gz[tr] | ((uint32_t)gz[tr + 1] << 8) | ((uint32_t)gz[tr + 2] << 16) | ((uint32_t)gz[tr + 3] << 24);
        const uint32_t want_len = gz[tr + 4] | ((uint32_t)gz[tr + 5] << 8) | ((uint32_t)gz[tr + 6] << 16) | ((uint32_t)gz[tr + 7] << 24);
        if (want_crc != crc || want_len != (uint32_t)member_text) {
            if (getenv("QUADE_GZ_DEBUG")) fprintf(stderr, "[qd_dev_gunzip] trailer at %lld: crc %08x (text's %08x), isize %u (text %llu)\n", (long long)tr, want_crc, crc, want_len, (unsigned long long)member_text);
            rc = QD_ERR_FORMAT;
            break;
        }
        ++members;
        pos = tr + 8;
    }
    if (rc == QD_OK && opos && hipMemcpy(out, d_out, (size_t)opos, hipMemcpyDeviceToHost) != hipSuccess) rc = QD_ERR_HIP;
    if (rc == QD_OK) *out_len = opos;
    return done(rc);
}
