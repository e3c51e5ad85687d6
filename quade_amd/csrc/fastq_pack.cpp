// Host-only helpers of libquade_hip.so: fastq text -> record index / packed index rows.
// They restate what the hot path consumed from the reference's (un-vendored) pyFastq reader:
// 4-line records, and a record whose sequence and quality lengths differ is dropped inside its
// own stream (pinned by the reference's bundled golden run).  Also the row layout of
// a plan (qd_plan_layout).  No GPU calls here: this file builds with plain g++ (sanitizer tests).
#include <algorithm>
#include <cstring>

#include "../../include/quade_hip.h"
#include "fastq_scan.h"

namespace {

inline void pack_row(const qd_layout* L, int k, const uint8_t* seq, const uint8_t* qual, int64_t len,
                     uint8_t* srow, uint8_t* qrow) {
    const int so = L->seq_off[k], sw = L->seq_width[k], ss = L->seq_stride[k];
    const int qo = L->qual_off[k], qw = L->qual_width[k], qs = L->qual_stride[k];
    int avail = (int)(len - so < 0 ? 0 : (len - so > sw ? sw : len - so));
    if (avail > 0) memcpy(srow, seq + so, (size_t)avail);
    memset(srow + avail, 0x00, (size_t)(ss - avail));
    avail = (int)(len - qo < 0 ? 0 : (len - qo > qw ? qw : len - qo));
    if (avail > 0) memcpy(qrow, qual + qo, (size_t)avail);
    memset(qrow + avail, 0xFF, (size_t)(qs - avail));
}

// row stride of a window: its width rounded up to an even number of bytes (so that the rows of two
// consecutive reads start 4-byte aligned), at least 2
int32_t even_stride(int32_t width) {
    const int32_t s = (width + 1) & ~1;
    return s < 2 ? 2 : s;
}

}  // namespace

extern "C" {

int qd_plan_layout(const qd_plan* P, qd_layout* L) {
    if (!P || !L) return QD_ERR_INVALID;
    memset(L, 0, sizeof *L);
    if (P->min_qual < 0 || P->min_qual > 40) return QD_ERR_INVALID;  // src/Quade.py:262
    const int is[2] = {P->idx1_start, P->idx2_start}, ie[2] = {P->idx1_end, P->idx2_end};
    const int ms[2] = {P->mol1_start, P->mol2_start}, me[2] = {P->mol1_end, P->mol2_end};
    L->n_streams = P->dual ? 2 : 1;
    for (int k = 0; k < 2; ++k) {
        L->seq_stride[k] = 8;
        L->qual_stride[k] = 8;
        if (k >= L->n_streams) continue;
        if (is[k] < 0 || ie[k] < is[k] || ms[k] < 0 || me[k] < ms[k]) return QD_ERR_INVALID;  // Quade.py:277-279
        if (ie[k] > 255 || me[k] > 255) return QD_ERR_UNSUPPORTED;
        const int iw = ie[k] - is[k], mw = me[k] - ms[k];
        int lo = 0, hi = 0;
        if (iw > 0 && mw > 0) {
            lo = std::min(is[k], ms[k]);
            hi = std::max(ie[k], me[k]);
        } else if (iw > 0) {
            lo = is[k];
            hi = ie[k];
        } else if (mw > 0) {
            lo = ms[k];
            hi = me[k];
        }
        if (hi - lo > QD_MAX_WINDOW) return QD_ERR_UNSUPPORTED;
        L->seq_off[k] = lo;
        L->seq_width[k] = hi - lo;
        L->seq_stride[k] = even_stride(hi - lo);
        L->qual_off[k] = is[k];
        L->qual_width[k] = iw;
        L->qual_stride[k] = even_stride(iw);
        L->key_width += iw;
        L->mol_width += mw;
    }
    if (L->key_width > QD_MAX_KEY) return QD_ERR_UNSUPPORTED;
    return QD_OK;
}


int64_t qd_fastq_index(const uint8_t* text, int64_t text_len, int64_t max_records, int64_t* rec_off,
                       int64_t* consumed) {
    if (!text || text_len < 0 || max_records < 0 || !rec_off) return QD_ERR_INVALID;
    int64_t pos = 0, n = 0;
    Rec r;
    while (n < max_records && next_record(text, text_len, pos, r)) {
        if (r.seq_end - r.seq == r.qual_end - r.qual) rec_off[n++] = r.head;
        pos = r.next;
    }
    rec_off[n] = pos;
    if (consumed) *consumed = pos;
    return n;
}

int64_t qd_pack_index_fastq(const qd_layout* L, int32_t k, const uint8_t* text, int64_t text_len,
                            int64_t max_records, uint8_t* seq_rows, uint8_t* qual_rows, uint8_t* len_rows,
                            int32_t* all_full, int64_t* consumed, uint32_t* short_idx, int64_t short_cap,
                            int64_t* n_short) {
    if (!L || k < 0 || k >= L->n_streams || !text || text_len < 0 || max_records < 0 || !seq_rows || !qual_rows)
        return QD_ERR_INVALID;
    const int ss = L->seq_stride[k], qs = L->qual_stride[k];
    const int64_t need = (int64_t)L->seq_off[k] + L->seq_width[k];
    int64_t pos = 0, n = 0, ns = 0;
    int32_t full = 1;
    Rec r;
    while (n < max_records && next_record(text, text_len, pos, r)) {
        const int64_t len = r.seq_end - r.seq;
        if (len == r.qual_end - r.qual) {
            pack_row(L, k, text + r.seq, text + r.qual, len, seq_rows + n * ss, qual_rows + n * qs);
            if (len_rows) len_rows[n] = (uint8_t)(len > 255 ? 255 : len);
            if (len < need) {
                full = 0;
                if (short_idx && ns < short_cap) short_idx[ns] = (uint32_t)n;
                ++ns;
            }
            ++n;
        }
        pos = r.next;
    }
    if (all_full) *all_full = full;
    if (consumed) *consumed = pos;
    if (n_short) *n_short = ns;
    return n;
}

int qd_pack_index_reads(const qd_layout* L, int32_t k, int64_t n, const uint8_t* seq, const uint8_t* qual,
                        const int64_t* offsets, uint8_t* seq_rows, uint8_t* qual_rows, uint8_t* len_rows,
                        int32_t* all_full) {
    if (!L || k < 0 || k >= L->n_streams || n < 0 || !seq || !qual || !offsets || !seq_rows || !qual_rows)
        return QD_ERR_INVALID;
    const int ss = L->seq_stride[k], qs = L->qual_stride[k];
    const int64_t need = (int64_t)L->seq_off[k] + L->seq_width[k];
    int32_t full = 1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (len < 0) return QD_ERR_INVALID;
        pack_row(L, k, seq + offsets[i], qual + offsets[i], len, seq_rows + i * ss, qual_rows + i * qs);
        if (len_rows) len_rows[i] = (uint8_t)(len > 255 ? 255 : len);
        if (len < need) full = 0;
    }
    if (all_full) *all_full = full;
    return QD_OK;
}

int qd_build_tags(const qd_layout* L, const qd_plan* P, int64_t n, const uint8_t* const seq_rows[2],
                  const uint8_t* const len_rows[2], const uint8_t* mol_rows, uint8_t* tag_rows, int32_t tag_stride,
                  uint8_t* tag_len) {
    if (!L || !P || n < 0 || !seq_rows || !tag_rows || !tag_len) return QD_ERR_INVALID;
    if (tag_stride < 2 + L->key_width + L->mol_width || 2 + L->key_width + L->mol_width > 255) return QD_ERR_INVALID;
    const int is[2] = {P->idx1_start, P->idx2_start}, ie[2] = {P->idx1_end, P->idx2_end};
    const int ms[2] = {P->mol1_start, P->mol2_start}, me[2] = {P->mol1_end, P->mol2_end};
    for (int k = 0; k < L->n_streams; ++k)
        if (!seq_rows[k]) return QD_ERR_INVALID;
    for (int64_t r = 0; r < n; ++r) {
        uint8_t* t = tag_rows + r * tag_stride;
        int o = 0;
        t[o++] = ':';
        for (int k = 0; k < L->n_streams; ++k) {  // IDX: I1 part then I2 part (Quade.py:217)
            const int len = (len_rows && len_rows[k]) ? len_rows[k][r] : 0x7FFFFFFF;
            const uint8_t* row = seq_rows[k] + r * L->seq_stride[k];
            const int e = ie[k] < len ? ie[k] : len;
            for (int c = is[k]; c < e; ++c) t[o++] = row[c - L->seq_off[k]];
        }
        const int mark = o;
        t[o++] = ':';
        int mo = 0;  // bytes of the device's mol row consumed so far
        for (int k = 0; k < L->n_streams; ++k) {  // MOL (Quade.py:218)
            const int len = (len_rows && len_rows[k]) ? len_rows[k][r] : 0x7FFFFFFF;
            const uint8_t* row = seq_rows[k] + r * L->seq_stride[k];
            const int e = me[k] < len ? me[k] : len;
            for (int c = ms[k]; c < e; ++c) t[o++] = mol_rows ? mol_rows[r * L->mol_width + mo++] : row[c - L->seq_off[k]];
        }
        if (o == mark + 1) o = mark;  // empty molecular index: ":IDX" only
        tag_len[r] = (uint8_t)o;
    }
    return QD_OK;
}

int64_t qd_format_records(const uint8_t* text, const int64_t* rec_off, const int64_t* sel, int64_t n_sel,
                          const uint8_t* tag_rows, int32_t tag_stride, const uint8_t* tag_len, uint8_t* out,
                          int64_t out_cap) {
    if (!text || !rec_off || (!sel && n_sel) || n_sel < 0 || !tag_rows || !tag_len || (!out && out_cap)) return QD_ERR_INVALID;
    int64_t need = 0;
    for (int64_t i = 0; i < n_sel; ++i) {
        const int64_t r = sel[i];
        need += (rec_off[r + 1] - rec_off[r]) + tag_len[r] + 8;  // upper bound per record
    }
    if (need > out_cap) return -need;
    int64_t o = 0;
    // The selected records lie all over the batch's text (they were scattered by routing code): every record starts with a
    // cache miss.  The lines of the record a few places ahead are requested while this one is copied (format was 0.61 of the
    // 2.64 core-seconds per M pairs of the end-to-end run, bound by those misses: profiles/r03_e2e_16m_level1_stages.txt).
    constexpr int64_t AHEAD = 6;
    for (int64_t i = 0; i < n_sel; ++i) {
        if (i + AHEAD < n_sel) {
            const uint8_t* q = text + rec_off[sel[i + AHEAD]];
            const int64_t len = rec_off[sel[i + AHEAD] + 1] - rec_off[sel[i + AHEAD]];
            for (int64_t b = 0; b < len && b < 512; b += 64) __builtin_prefetch(q + b, 0, 0);
            __builtin_prefetch(tag_rows + sel[i + AHEAD] * tag_stride, 0, 0);
        }
        const int64_t r = sel[i];
        const uint8_t* p = text + rec_off[r];
        const uint8_t* end = text + rec_off[r + 1];
        // line 1: header
        const uint8_t* nl = (const uint8_t*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) return QD_ERR_FORMAT;
        const uint8_t* h = p + (nl > p ? 1 : 0);  // drop the first byte ('@')
        auto is_ws = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
        while (h < nl && is_ws(*h)) ++h;  // str.split() semantics: leading blanks are skipped
        const uint8_t* he = h;
        while (he < nl && !is_ws(*he)) ++he;
        out[o++] = '@';
        memcpy(out + o, h, (size_t)(he - h));
        o += he - h;
        memcpy(out + o, tag_rows + r * tag_stride, tag_len[r]);
        o += tag_len[r];
        out[o++] = '\n';
        // lines 2-4.  The usual record -- a bare "+" line, "\n" line ends, sequence and quality of one length (which the
        // reader checked) -- is laid out by arithmetic alone: header, S bytes, "\n+\n", S bytes, "\n"; anything else
        // (text behind the '+', "\r\n", a record the arithmetic does not fit) takes the searching path below.
        const uint8_t* s0 = nl + 1;
        const int64_t rest = end - s0;  // S + 1 + 1 + 1 + S + 1
        if (rest >= 4 && !(rest & 1)) {
            const int64_t S = (rest - 4) / 2;
            if (s0[S] == '\n' && s0[S + 1] == '+' && s0[S + 2] == '\n' && end[-1] == '\n' && (S == 0 || (s0[S - 1] != '\r' && end[-2] != '\r')) &&
                !memchr(s0, '\n', (size_t)S) && !memchr(s0 + S + 3, '\n', (size_t)S)) {
                memcpy(out + o, s0, (size_t)S + 3);  // sequence, "\n+\n"
                o += S + 3;
                memcpy(out + o, s0 + S + 3, (size_t)S + 1);  // quality, "\n"
                o += S + 1;
                continue;
            }
        }
        // line 2: sequence
        const uint8_t* s1 = (const uint8_t*)memchr(s0, '\n', (size_t)(end - s0));
        if (!s1) return QD_ERR_FORMAT;
        const uint8_t* se = (s1 > s0 && s1[-1] == '\r') ? s1 - 1 : s1;
        memcpy(out + o, s0, (size_t)(se - s0));
        o += se - s0;
        out[o++] = '\n';
        out[o++] = '+';
        out[o++] = '\n';
        // line 3: separator (dropped), line 4: quality
        const uint8_t* p3 = (const uint8_t*)memchr(s1 + 1, '\n', (size_t)(end - (s1 + 1)));
        if (!p3) return QD_ERR_FORMAT;
        const uint8_t* q0 = p3 + 1;
        const uint8_t* q1 = (const uint8_t*)memchr(q0, '\n', (size_t)(end - q0));
        if (!q1) return QD_ERR_FORMAT;
        const uint8_t* qe = (q1 > q0 && q1[-1] == '\r') ? q1 - 1 : q1;
        memcpy(out + o, q0, (size_t)(qe - q0));
        o += qe - q0;
        out[o++] = '\n';
    }
    return o;
}

}  // extern "C"
