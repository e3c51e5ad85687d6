// Record scanner shared by the host helpers (fastq_pack.cpp) and the native reader (quade_io.cpp).
#pragma once
#include <stdint.h>
#include <string.h>

namespace {

struct Rec {
    int64_t head, seq, seq_end, qual, qual_end, next;
};

// Parses one record starting at `pos`.  Returns false when fewer than four newline-terminated
// lines remain.  A trailing '\r' is not part of a line.
inline bool next_record(const uint8_t* t, int64_t len, int64_t pos, Rec& r) {
    int64_t p = pos;
    int64_t starts[4], ends[4];
    for (int i = 0; i < 4; ++i) {
        if (p >= len) return false;
        const void* nl = memchr(t + p, '\n', (size_t)(len - p));
        if (!nl) return false;
        const int64_t e = (const uint8_t*)nl - t;
        starts[i] = p;
        ends[i] = (e > p && t[e - 1] == '\r') ? e - 1 : e;
        p = e + 1;
    }
    r.head = starts[0];
    r.seq = starts[1];
    r.seq_end = ends[1];
    r.qual = starts[3];
    r.qual_end = ends[3];
    r.next = p;
    return true;
}

}  // namespace
