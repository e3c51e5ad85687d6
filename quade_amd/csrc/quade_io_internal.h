// Seams of the host I/O (quade_io.cpp) that the device chunk pipeline (quade_pipe.cpp) builds on.  Internal: not part of the ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

struct qd_sink;
struct qd_reader;

namespace qdio {

// ---- input: a fastq(.gz) file as pieces of TEXT in file order (gzip members of any kind inflated by the reader's own
// thread / the library's pool, plain text read as is) -- what the native reader feeds its record scanner.
// start_offset > 0: the file is read from that byte on (a BGZF file that turns into ordinary gzip members half way).
qd_reader* raw_open(const char* path, int64_t start_offset, std::string* err);
// 1: *ptr / *len are the next piece, valid until the next call; 0: end of the file; -1: error (*err)
int raw_next(qd_reader* r, const uint8_t** ptr, size_t* len, std::string* err);
void raw_close(qd_reader* r);

// total size of the BGZF block that starts at p (header .. ISIZE), 0 when p does not start with a complete BGZF header
size_t bgzf_block_size(const uint8_t* p, size_t avail);
// whole BGZF blocks (or any gzip members) back to back -> exactly out_len bytes of text; false: damaged
bool host_inflate_members(const uint8_t* comp, size_t comp_len, uint8_t* out, size_t out_len);
// one gzip member of `level` (-1 = Huffman only) from n bytes of text, by the host's coder
bool host_gzip_member(const uint8_t* text, size_t n, int level, std::vector<uint8_t>* out);
uint32_t crc32(const uint8_t* p, size_t n);

// ---- output: the files of a sink
struct SinkInfo {
    int level;
    bool write_pass, write_fail, write_undet;
    uint32_t n_samples;
};
SinkInfo sink_info(const qd_sink* s);
// file k (0 = R1, 1 = R2) of routing code `code`; the destination's two files are created (truncated, announced) at the
// first call for either, as at its first routed pair (src/FastqWriter.py:55-57).  nullptr: could not create (sink error set).
void* sink_file(qd_sink* s, uint32_t code, int k);
// appends on the calling thread; false: the sink's error is set
bool sink_append(qd_sink* s, void* file, const uint8_t* data, size_t n);
void sink_account(qd_sink* s, int64_t members, int64_t device_members, int64_t text_bytes, int64_t gzip_bytes);
void sink_fail(qd_sink* s, const std::string& msg);

void pool_submit(std::function<void()> fn, bool urgent = false);  // urgent: ahead of the jobs already queued
int pool_size();

}  // namespace qdio
