// Parallel inflate of ONE gzip stream (any number of members, members of any size) on host threads.
//
// What it replaces: the gunzip inside pyFastq.FastqReader that the reference's chunk loops draw their records from
// (src/Quade.py:203-206, 234-236 open ordinary .fastq.gz files; its own fixtures test/dataset/*.fastq.gz are single
// gzip members) -- a serial bit stream that one thread inflates at a few hundred MB/s, whatever the machine.
//
// How (the published two-pass scheme of pugz / rapidgzip, restated for this reader):
//   1. the compressed file is cut at fixed byte offsets; chunk j looks for the first bit position >= its offset at
//      which a dynamic-Huffman DEFLATE block begins (every bit offset is tried: 3 header bits, the code-length code
//      must be a complete prefix code, both alphabets must build, the block must decode to text and be followed by
//      another valid block header);
//   2. from there the chunk is inflated WITHOUT its 32 KiB history: the output is 16-bit symbols, a literal byte or
//      a marker "byte i of the unknown window" -- the window is pre-filled with markers, so a match that reaches
//      into it simply copies markers; the chunk stops at the first dynamic block at or beyond the next offset;
//   3. in file order: a chunk is accepted only when it began exactly where its predecessor stopped (so every
//      accepted boundary is proven by the chain from the true start of the stream, never guessed: a false
//      candidate, or a search that found nothing, makes the coordinator inflate that stretch itself with the known
//      window); its last 32 KiB are resolved against the predecessor's window (the only serial step, ~30 us), and
//      the chunk's symbols are translated to bytes by a pool job; CRC-32 and ISIZE of every member are checked on
//      the translated text (per-piece CRCs combined in order).
// The result is byte for byte what zlib produces, or an error -- there is no heuristic in what is delivered.
// No GPU calls here: this file builds with plain g++ (sanitizer tests).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

namespace qdpgz {

struct Text {  // a piece of inflated text; pieces come in file order
    const uint8_t* data = nullptr;
    size_t len = 0;
    std::shared_ptr<void> owner;  // whatever holds the bytes (returned to the buffer pool with the last reference)
};

using Submit = std::function<void(std::function<void()>)>;  // runs a job on a worker thread, some time

struct Options {
    size_t chunk_bytes = 4u << 20;  // compressed bytes per chunk
    int in_flight = 8;              // chunks being inflated / translated at once
};

struct Stats {
    int64_t chunks = 0;        // pieces delivered
    int64_t parallel = 0;      // of them: inflated speculatively and proven by the chain
    int64_t serial = 0;        // inflated by the coordinator with the known window (first chunk, failed candidates)
    int64_t members = 0;       // gzip members whose CRC-32 and ISIZE were checked
    int64_t search_bits = 0;   // bit positions tried by the block searches
};

struct State;

class Gunzip {
  public:
    // data[0..size) must stay valid until the object is destroyed (the destructor waits for outstanding jobs)
    Gunzip(const uint8_t* data, size_t size, const Options& opt, Submit submit);
    ~Gunzip();
    Gunzip(const Gunzip&) = delete;
    Gunzip& operator=(const Gunzip&) = delete;
    // 1: *out holds the next piece of text; 0: clean end of the file; -1: error() says why (and stays that way)
    int next(std::shared_ptr<Text>* out);
    const std::string& error() const;
    Stats stats() const;
    // how far the compressed input has been consumed for certain (byte offset behind the last accepted piece)
    size_t consumed() const;

  private:
    std::shared_ptr<State> s_;
};

// first bytes of a gzip member (RFC 1952): magic, CM = 8, no reserved flag bits
inline bool looks_like_gzip(const uint8_t* p, size_t n) { return n >= 10 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && !(p[3] & 0xe0); }

}  // namespace qdpgz
