/*
 * quade_hip.h -- C ABI of the MI355X (gfx950) demultiplexing library, libquade_hip.so.
 *
 * Drop-in boundary for the per-read hot path of a-slide/Quade 0.3.2.  The reference has no FFI:
 * its seam is one Python classmethod call per read pair.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference).  The library is
 * batch-granular: one call processes n read pairs whose index reads were packed into
 * fixed-stride rows (layout: qd_layout).
 *
 * Conventions
 *   - plain C types only; every function returns QD_OK (0) or a negative QD_ERR_* code;
 *     qd_last_error() gives the text.  No exceptions or callbacks cross the boundary.
 *   - a qd_ctx is bound to one HIP device and must be driven by one thread at a time;
 *     different contexts may be driven concurrently from different threads.
 *   - there is NO CPU fallback: qd_create() fails with QD_ERR_NO_DEVICE when no gfx950 device is
 *     usable.  The host-only helpers (qd_plan_layout, qd_pack_*, qd_fastq_*, qd_build_tags,
 *     qd_format_records, qd_version, qd_strerror) never touch the GPU.
 *
 * Routing code (uint16) written per pair -- src/Sample.py:65-91:
 *     0xFFFF            index matches no sample            (Sample.py:86-91, "Undetermined")
 *     2*ordinal         matched, min phred >= minimal_qual (Sample.py:70-75, "<name>_pass")
 *     2*ordinal + 1     matched, quality gate failed       (Sample.py:78-83, "<name>_fail")
 *   ordinal = position of the sample in SAMPLE_LIST, i.e. order of the [sampleN] sections
 *   (src/Quade.py:133, src/Sample.py:153).
 *
 * Counter vector (uint64[2*S+4]) -- src/Sample.py:32,62,71-72,79-80,88:
 *     [0] TOTAL  [1] PASS_QUAL  [2] FAIL_QUAL  [3] UNDETERMINED
 *     [4+2*i] sample i pass_qual   [5+2*i] sample i fail_qual
 */
#ifndef QUADE_HIP_H
#define QUADE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_ABI_VERSION 6

#define QD_OK 0
#define QD_ERR_INVALID (-1)     /* bad argument (NULL pointer, misaligned buffer, size out of range)   */
#define QD_ERR_NO_DEVICE (-2)   /* no usable HIP device / not gfx950 -- the library never falls back   */
#define QD_ERR_HIP (-3)         /* a HIP runtime call failed; text in qd_last_error()                   */
#define QD_ERR_STATE (-4)       /* call order: plan and barcodes must be set before demux/submit        */
#define QD_ERR_UNSUPPORTED (-5) /* plan outside the supported envelope (window > QD_MAX_WINDOW ...)     */
#define QD_ERR_BARCODE (-6)     /* barcode table rejected (duplicate index: src/Sample.py:140)          */
#define QD_ERR_FORMAT (-7)      /* malformed fastq text handed to a host helper                         */

#define QD_CODE_UNDETERMINED 0xFFFFu
#define QD_MAX_WINDOW 64   /* widest index-read window (idx U mol) kept per row, bytes                  */
#define QD_MAX_KEY 32      /* longest fused barcode the match kernels compare, bytes                    */
#define QD_MAX_SAMPLES 32767

typedef struct qd_ctx qd_ctx;

/* ---- plan: the parameters the hot path reads (src/Quade.py:96-116) -------------------------------
 * start values are 0-based (the conf's 1-based start minus 1, Quade.py:106), end values are the
 * conf's 1-based inclusive ends, so a slice is read[start:end] exactly as Quade.py:217-218,246-247.
 * Disabled parts are 0:0 (Quade.py:109-116).  min_qual is the phred threshold (Quade.py:96,
 * Sample.py:70), 0..40 (Quade.py:262). */
typedef struct qd_plan {
    int32_t dual;     /* [index] index2 (Quade.py:100): 1 = two index reads fused, 0 = one */
    int32_t min_qual; /* [quality] minimal_qual */
    int32_t idx1_start, idx1_end;
    int32_t idx2_start, idx2_end;
    int32_t mol1_start, mol1_end;
    int32_t mol2_start, mol2_end;
} qd_plan;

/* ---- row layout derived from a plan --------------------------------------------------------------
 * For index stream k (0 = index_R1, 1 = index_R2):
 *   seq row  : seq_stride[k] bytes = columns [seq_off[k], seq_off[k]+seq_width[k]) of the read's
 *              sequence line, as read (case preserved), zero-padded (0x00) to the stride and where
 *              the read is shorter than the window.
 *   qual row : qual_stride[k] bytes = quality characters (Phred+33 text, as read) of columns
 *              [qual_off[k], qual_off[k]+qual_width[k]) = the barcode slice only (Sample.py:70 takes
 *              the minimum over the fused barcode positions, nothing else), padded with 0xFF.
 *   len row  : optional uint8 per read = min(255, read length).  Omitted (NULL) when every read of
 *              the batch covers its whole window -- then the fast kernels run.
 * Outputs: codes = uint16 per pair; mol = mol_width bytes per pair (raw case, I1 part then I2 part
 * packed together, zero padded when reads are short) -- Quade.py:218.  A stride is its window's width
 * rounded up to an even number of bytes (minimum 2), so rows carry at most one byte of padding.     */
typedef struct qd_layout {
    int32_t n_streams;
    int32_t seq_off[2], seq_width[2], seq_stride[2];
    int32_t qual_off[2], qual_width[2], qual_stride[2];
    int32_t key_width; /* K = fused barcode slice width = sum of (idx_end-idx_start)            */
    int32_t mol_width; /* M = fused molecular slice width; 0 when both molecular flags are off  */
} qd_layout;

/* Host only.  Replaces nothing in the reference; states the packing contract. */
int qd_plan_layout(const qd_plan* plan, qd_layout* out);

/* ---- library ------------------------------------------------------------------------------------*/
int qd_version(void);
const char* qd_strerror(int code);
/* Text of the last error on this context (ctx == NULL: last error of qd_create on this thread). */
const char* qd_last_error(const qd_ctx* ctx);

/* ---- context --------------------------------------------------------------------------------------
 * qd_create replaces the implicit process-global state of src/Sample.py:32-44 (one run per
 * process) with an explicit context bound to HIP device `device_id`. */
int qd_device_count(int32_t* n_devices); /* gfx950 or not; QD_ERR_NO_DEVICE when the runtime finds none */
int qd_create(int device_id, qd_ctx** out);
int qd_destroy(qd_ctx* ctx);
/* name: >= 64 bytes.  Any output pointer may be NULL. */
int qd_device_info(const qd_ctx* ctx, char* name, int32_t name_cap, int32_t* compute_units,
                   int64_t* total_mem_bytes);

/* Replaces Sample.CLASS_INIT(min_qual=...) (src/Sample.py:48-54, called at src/Quade.py:125-129)
 * and the position fields of Quade.__init__ (src/Quade.py:99-116).  The write_* flags are not a
 * device concern: counters move regardless (Sample.py:71-91) and routing is decided by code. */
int qd_set_plan(qd_ctx* ctx, const qd_plan* plan);
int qd_get_layout(const qd_ctx* ctx, qd_layout* out);

/* Replaces the registration side of Sample.__init__ (src/Sample.py:132-153, called at
 * src/Quade.py:137,139): n_samples upper-case barcodes in ordinal order, concatenated in
 * `barcodes`, barcode i = barcodes[offsets[i] .. offsets[i+1]).  Lengths may differ from the slice
 * width and from each other (the reference never checks, Sample.py:132-141); such a barcode can
 * only match reads whose clamped slice has exactly that length.  Duplicate -> QD_ERR_BARCODE.
 * Alphabet checks stay on the Python side (message text parity, Sample.py:141). Resets counters. */
int qd_set_barcodes(qd_ctx* ctx, int32_t n_samples, const uint8_t* barcodes, const int32_t* offsets);

/* ---- device-resident hot path ---------------------------------------------------------------------
 * Replaces, for n_pairs reads at once: index/molecular extraction and fusion (src/Quade.py:217-218,
 * 246-247) + Sample.FINDER (src/Sample.py:56-91).  All pointers are DEVICE pointers, 16-byte
 * aligned, laid out as qd_layout says.  seq[1]/qual[1]/len[1] are ignored for a single-index plan.
 * codes_dev: n_pairs uint16.  mol_dev: n_pairs*mol_width bytes, may be NULL when mol_width == 0.
 * `stream` is a hipStream_t: NULL is HIP's null (default) stream, ordered with the caller's other
 * default-stream work; QD_STREAM_CONTEXT is the context's own non-blocking stream.  Asynchronous:
 * returns after the launch.  Counters accumulate in the context.  The library never synchronises the
 * whole device: rows written by work on ANOTHER stream must be complete (or ordered by the caller)
 * before the launch on `stream` reads them. */
#define QD_STREAM_CONTEXT ((void*)(intptr_t)-1)
typedef struct qd_rows {
    const uint8_t* seq[2];
    const uint8_t* qual[2];
    const uint8_t* len[2]; /* NULL: every read covers its window */
} qd_rows;
int qd_demux_device(qd_ctx* ctx, int64_t n_pairs, const qd_rows* rows, uint16_t* codes_dev,
                    uint8_t* mol_dev, void* stream);

/* Ragged batch: some reads are shorter than their window (Python slice clamping of a short index
 * read, src/Quade.py:217-218), and the caller knows which.  rows->len must be given;
 * short_idx_dev[0..n_short) = the pair indices (device memory, unique, any order) at which ANY stream's
 * read is shorter than seq_off+seq_width of that stream.  When the plan is eligible for the fast
 * kernels, the whole batch runs on them and only the listed pairs are redone with the length-aware
 * byte-granular path (results and counters are those of running every pair that way); otherwise, or
 * when more than half of the pairs are listed, the generic kernel takes the batch. */
int qd_demux_device_ragged(qd_ctx* ctx, int64_t n_pairs, const qd_rows* rows, uint16_t* codes_dev, uint8_t* mol_dev,
                           int64_t n_short, const uint32_t* short_idx_dev, void* stream);

/* Which kernel qd_demux_device would launch for a batch: 1 = fast (LDS table, vector rows),
 * 2 = generic.  Informational (tests, bench). */
int qd_kernel_kind(const qd_ctx* ctx, int has_len);

/* Tuning / test knobs (no reference counterpart).  Names:
 *   "fast_workgroups_per_cu"  0 = automatic (default), 1..4096 = fixed
 *   "fast_block"              0 = automatic (default), 256 / 512 / 1024 threads per workgroup
 *   "mol_strips"              1 = stage molecular bytes through LDS for 16-byte stores (default), 0 = off
 *   "force_generic"           1 = always launch the generic kernel
 *   "kernel"                  0 = automatic (default), 1 = fast (when the plan is eligible), 2 = generic
 *   "fold_pairs"              the device counts into 32-bit per-workgroup rows that are folded into 64-bit totals
 *                             before this many pairs have been launched since the last fold (default and
 *                             maximum 2^32 - 1; tests lower it) */
int qd_set_option(qd_ctx* ctx, const char* name, int64_t value);

/* ---- counters: replace the class counters of src/Sample.py:32,144 and feed Sample.REPORT ---------
 * qd_get_counts waits for outstanding work of this context (only), then writes 2*S+4 values. */
int qd_get_counts(qd_ctx* ctx, uint64_t* out, int32_t n_values);
int qd_reset_counts(qd_ctx* ctx);
/* Adds a counter vector (layout of qd_get_counts, e.g. another context's) to this context's totals.  The reference
 * keeps ONE set of class counters per run (src/Sample.py:32,144); a process that drives several contexts on one
 * device (chunk workers) folds the others into the context that is the member of its communicator before
 * qd_reduce_counts, or into any one of them before the report.  Waits for this context's outstanding work.
 * QD_ERR_INVALID when the vector's aggregates do not add up (TOTAL = PASS + FAIL + UNDETERMINED over the samples). */
int qd_add_counts(qd_ctx* ctx, const uint64_t* counts, int32_t n_values);
/* Waits for every outstanding launch / copy this context issued (on its own stream, its slots'
 * streams and the caller's streams it was handed) -- not for other contexts' work on the device. */
int qd_synchronize(qd_ctx* ctx);

/* ---- host-staged streaming: pinned slots, H2D || kernel || D2H -------------------------------------
 * Replaces the per-read loop body of Quade.double_index_parser / simple_index_parser
 * (src/Quade.py:210-221, 240-250) for host-resident rows.  The library owns pinned host and device
 * buffers of n_slots slots x max_pairs.  A slot's host buffers are caller-writable from qd_wait()
 * (or creation) until the next qd_submit() of that slot. */
typedef struct qd_slot_buffers {
    uint8_t* seq[2];
    uint8_t* qual[2];
    uint8_t* len[2];
    uint16_t* codes; /* valid after qd_wait */
    uint8_t* mol;    /* valid after qd_wait; NULL when mol_width == 0 */
    int64_t max_pairs;
    uint32_t* short_idx[2]; /* per stream: indices of the reads shorter than their window (qd_submit_ragged) */
    int64_t short_cap;      /* entries each short_idx array holds */
} qd_slot_buffers;
int qd_slots_create(qd_ctx* ctx, int32_t n_slots, int64_t max_pairs);
int qd_slots_destroy(qd_ctx* ctx);
int qd_slot_get(qd_ctx* ctx, int32_t slot, qd_slot_buffers* out);
/* has_len != 0: the len rows were filled and the generic kernel runs. Non-blocking. */
int qd_submit(qd_ctx* ctx, int32_t slot, int64_t n_pairs, int32_t has_len);
/* The len rows were filled and stream k's short reads are listed (ascending) in the slot's
 * short_idx[k][0..n_short[k]) -- what qd_pack_index_fastq writes.  Indices >= n_pairs are ignored.
 * Runs like qd_demux_device_ragged; a count above short_cap means "too many to list": the generic
 * kernel takes the batch.  Non-blocking. */
int qd_submit_ragged(qd_ctx* ctx, int32_t slot, int64_t n_pairs, const int64_t n_short[2]);
int qd_wait(qd_ctx* ctx, int32_t slot);

/* ---- host helpers: fastq text -> rows (replace what the path consumed from pyFastq, a9) -----------
 * Scans decompressed fastq text (4-line records).  A record whose sequence and quality lengths
 * differ is skipped inside its own stream (pinned by the reference's golden run);
 * a trailing partial record ends the scan.  For every kept record r (r < max_records):
 *   rec_off[r]  = byte offset of its '@' line,   rec_off[n] = offset one past the last kept record
 * Returns the number of kept records (>= 0) or a negative error code.  *consumed = bytes scanned
 * (start of the first record NOT consumed) so a caller can stream a file in pieces. */
int64_t qd_fastq_index(const uint8_t* text, int64_t text_len, int64_t max_records, int64_t* rec_off,
                       int64_t* consumed);

/* Packs stream `k` (0/1) of `layout` from fastq text: for kept record r writes seq row r, qual row
 * r and len row r (len_rows may be NULL).  *all_full is set to 0 when any read is shorter than its
 * window (then the caller must pass len rows to the device).  short_idx (may be NULL): the indices
 * r of those short reads, ascending, at most short_cap of them stored; *n_short = how many there
 * were (may exceed short_cap).  Same skipping rule and return value as qd_fastq_index. */
int64_t qd_pack_index_fastq(const qd_layout* layout, int32_t k, const uint8_t* text, int64_t text_len,
                            int64_t max_records, uint8_t* seq_rows, uint8_t* qual_rows,
                            uint8_t* len_rows, int32_t* all_full, int64_t* consumed,
                            uint32_t* short_idx, int64_t short_cap, int64_t* n_short);

/* Packs rows from already separated reads (concatenated sequence and quality bytes + offsets):
 * used by tests and by callers that hold records rather than text. */
int qd_pack_index_reads(const qd_layout* layout, int32_t k, int64_t n, const uint8_t* seq,
                        const uint8_t* qual, const int64_t* offsets, uint8_t* seq_rows,
                        uint8_t* qual_rows, uint8_t* len_rows, int32_t* all_full);

/* ---- host helpers: routed records -> output text (replace FastqWriter.__call__'s formatting) --------
 * qd_build_tags: the name suffix of every pair, ":IDX" or ":IDX:MOL" (src/FastqWriter.py:61-66):
 * IDX = fused barcode slice as read (NOT case folded), MOL = fused molecular slice; ":MOL" is
 * omitted when the molecular slice is empty (empty molecular index is falsy).  Slices
 * are clamped to the read length when len rows are given.  mol_rows (optional): the molecular
 * bytes written by the device (n x mol_width); when NULL they are sliced from seq_rows on the
 * host -- both give the same bytes.  tag row r = tag_rows + r*tag_stride,
 * tag_stride >= 2 + key_width + mol_width; tag_len[r] = bytes used. */
int qd_build_tags(const qd_layout* layout, const qd_plan* plan, int64_t n, const uint8_t* const seq_rows[2],
                  const uint8_t* const len_rows[2], const uint8_t* mol_rows, uint8_t* tag_rows,
                  int32_t tag_stride, uint8_t* tag_len);

/* qd_format_records: for the n_sel records sel[0..n_sel) (indices into rec_off, input order) of
 * fastq `text`, writes "@" + name + tag + "\n" + seq + "\n+\n" + qual + "\n" -- the record format
 * of the reference's output (src/FastqWriter.py:68-69 via FastqSeq.fastqstr; pinned by the bundled
 * goldens): name = header line without its first byte, cut at the first ASCII whitespace.
 * Returns bytes written, or -(bytes needed) when out_cap is too small. */
int64_t qd_format_records(const uint8_t* text, const int64_t* rec_off, const int64_t* sel, int64_t n_sel,
                          const uint8_t* tag_rows, int32_t tag_stride, const uint8_t* tag_len, uint8_t* out,
                          int64_t out_cap);

/* ---- multi-GPU: the one exchange of the path ------------------------------------------------------------
 * Read pairs are independent (src/Sample.py:56-91 touches nothing but counters) and chunks are
 * independent files (src/Quade.py:198,229), so work shards across GPUs with no data-path collective.
 * What is exchanged is what the reference keeps in class-level counters (src/Sample.py:32,144): one
 * sum of the uint64[2S+1] counters + TOTAL over all member contexts, an RCCL all-reduce over xGMI
 * (librccl.so.1 is loaded on first use; single-GPU users never load it).
 *   one process, several devices : qd_comm_create_local(contexts, n)            (ncclCommInitAll)
 *   one process per device       : rank 0 calls qd_comm_unique_id and hands the 128 bytes to the other
 *                                  ranks by any means; every rank calls qd_comm_create_rank (ncclCommInitRank)
 * qd_reduce_counts waits for the member contexts' outstanding work, sums, and gives every caller the
 * total in the layout of qd_get_counts.  Member contexts must hold the same sample table. */
typedef struct qd_comm qd_comm;
#define QD_UNIQUE_ID_BYTES 128
int qd_comm_unique_id(uint8_t id[QD_UNIQUE_ID_BYTES]);
int qd_comm_create_local(qd_ctx* const* contexts, int32_t n_contexts, qd_comm** out);
int qd_comm_create_rank(qd_ctx* ctx, int32_t world_size, int32_t rank, const uint8_t id[QD_UNIQUE_ID_BYTES],
                        qd_comm** out);
int qd_comm_world(const qd_comm* comm);
int qd_reduce_counts(qd_comm* comm, uint64_t* out, int32_t n_values);
int qd_comm_destroy(qd_comm* comm);
const char* qd_comm_last_error(void); /* text of the last qd_comm_* / qd_reduce_counts error on this thread */

/* ---- host I/O: routed records -> per-destination fastq.gz files ---------------------------------------
 * A sink is one output directory's set of destinations: "<name>_pass", "<name>_fail" per sample and
 * "Undetermined", each a pair of files <dest>_R1.fastq.gz / <dest>_R2.fastq.gz (src/FastqWriter.py:29-31,
 * src/Sample.py:44,147-148).  qd_sink_route replaces, for a whole batch, the routing tail of
 * Sample.FINDER (src/Sample.py:74-75,82-83,90-91: writer called when the write_* flag of the pair's
 * category is set) and FastqWriter.__call__/flush_buffers (src/FastqWriter.py:48-90): name tag,
 * record format (as qd_format_records), lazy creation of a destination's two files at its first
 * routed pair (truncating), gzip members appended in input order.  Compression runs on a thread pool
 * owned by the library (libdeflate when libdeflate.so.0 loads, zlib otherwise); no file descriptor
 * is held between members.  qd_sink_route returns once the batch's text has been consumed (the
 * caller's buffers are free again); compression and the appends continue behind it.  A sink is driven
 * by one thread at a time; different sinks may be driven concurrently. */
typedef struct qd_sink qd_sink;
typedef struct qd_text_batch qd_text_batch; /* a batch of the native reader, below */
/* Threads of the I/O pool: n_threads > 0 sets it (before the pool's first use), 0 = one per hardware
 * core this process may use (qd_host_cores), < 0 = query only.  Returns the size in effect. */
int qd_io_threads(int32_t n_threads);
int qd_io_backend(void); /* 1 = libdeflate, 0 = zlib */
int qd_host_cores(void); /* cores this process may use: affinity mask capped by the cgroup CPU quota */
/* names: n_samples sample names in ordinal order (SAMPLE_LIST order, src/Sample.py:153); gzip_level -1..9 (-1 = Huffman coding only, no string matching: ~3x the speed of level 1, larger files);
 * write_*: the [output] flags (src/Quade.py:125-129 -> Sample.CLASS_INIT). */
int qd_sink_create(const char* outdir, int32_t n_samples, const char* const* names, int32_t gzip_level,
                   int32_t write_pass, int32_t write_fail, int32_t write_undetermined, qd_sink** out);
int qd_sink_set_quiet(qd_sink* sink, int32_t quiet); /* 1: no "Create ... file" lines on stdout */
/* codes: the device's routing codes of the batch's n pairs; r1/r2 text + rec_off: the insert reads
 * as qd_fastq_index gives them (record i of both = pair i); tags as qd_build_tags gives them. */
int qd_sink_route(qd_sink* sink, int64_t n_pairs, const uint16_t* codes, const uint8_t* r1_text,
                  const int64_t* r1_rec_off, const uint8_t* r2_text, const int64_t* r2_rec_off,
                  const uint8_t* tag_rows, int32_t tag_stride, const uint8_t* tag_len);
/* Same, for insert reads that came from the native reader (qd_reader_next): the sink takes the two text
 * batches over (the caller must NOT free their handles afterwards, whatever the return code) and copies the
 * tags, so the call returns right after the scatter; formatting and compression run behind it. */
int qd_sink_route_batches(qd_sink* sink, int64_t n_pairs, const uint16_t* codes, const qd_text_batch* r1,
                          const qd_text_batch* r2, const uint8_t* tag_rows, int32_t tag_stride,
                          const uint8_t* tag_len);
int qd_sink_flush(qd_sink* sink); /* waits until every member is in its file (src/Sample.py:93-102 FLUSH_ALL) */
int qd_sink_stats(qd_sink* sink, int64_t* members, int64_t* text_bytes, int64_t* gzip_bytes, int64_t* files);
const char* qd_sink_last_error(const qd_sink* sink);
int qd_sink_close(qd_sink* sink); /* flush + destroy */

/* Writes n bytes as a gzip file, compressed on the library's pool: member_bytes > 0 = members of that much
 * text, 0 = one member, -1 = BGZF blocks (bgzip / htslib layout: 64 KiB members that carry their size in a
 * 'BC' extra subfield, closed by the empty end-of-file block).  Tooling (synthetic inputs); no reference
 * counterpart. */
/* Where the host's CPU time goes (measurement): thread-CPU seconds per stage of the reader, the sink and the pool since the
 * process started or since the last call with reset != 0, summed over all threads of the library.  names[i] (static strings)
 * and seconds[i] for i < the return value (<= cap); either array may be NULL. */
int qd_io_stage_seconds(const char** names, double* seconds, int32_t cap, int32_t reset);
int qd_write_gzip_file(const char* path, const uint8_t* data, int64_t n_bytes, int32_t level, int64_t member_bytes);

/* ---- host I/O: fastq(.gz) file -> batches of whole records ----------------------------------------------
 * Replaces pyFastq.FastqReader as the reference uses it (src/Quade.py:203-214: one .next() per record):
 * a reader owns two threads that read + inflate ("*.gz": gzip, any number of members; BGZF / bgzip files
 * are cut into blocks by their header fields and inflated in parallel on the library's pool; else plain
 * text) and scan + batch the file ahead of the consumer.  A batch holds exactly batch_records kept records
 * (fewer at the end of the file only); a record whose sequence and quality lengths differ is skipped
 * inside its own stream, a trailing partial record ends the stream, a last line
 * without newline counts.  text/rec_off stay valid until qd_text_batch_free(handle). */
typedef struct qd_reader qd_reader;
struct qd_text_batch {
    const uint8_t* text;    /* whole records, 4 lines each */
    int64_t text_len;
    const int64_t* rec_off; /* n_records + 1 offsets into text, as qd_fastq_index gives them */
    int64_t n_records;      /* 0 = end of the stream (handle is NULL then) */
    void* handle;
};
int qd_reader_open(const char* path, int64_t batch_records, int32_t queue_depth, qd_reader** out);
int qd_reader_next(qd_reader* reader, qd_text_batch* out); /* blocks until a batch is ready */
int qd_text_batch_free(void* handle);
int qd_reader_close(qd_reader* reader);
const char* qd_reader_last_error(const qd_reader* reader); /* reader == NULL: why qd_reader_open failed on this thread */

/* ---- ordinary gzip files: parallel inflate on the library's pool ------------------------------------------------
 * The reference opens ordinary .fastq.gz files (src/Quade.py:203-206, 234-236; its fixtures under test/dataset
 * are single gzip members): one DEFLATE stream, which a single thread inflates at a few hundred MB/s.  The reader
 * cuts such a file at fixed offsets, finds a block start behind every cut by trying every bit offset, inflates
 * the chunks in parallel into 16-bit symbols over a window of markers, proves every boundary by the chain from the
 * start of the stream, resolves the markers in file order and checks every member's CRC-32 and ISIZE
 * (quade_amd/csrc/quade_pgz.h).  Result: the bytes zlib would give, or an error.
 * qd_io_set_option names (process-wide; tests and tuning):
 *   "parallel_gunzip"        1 (default) / 0 = one thread per file (libdeflate per member, streaming zlib beyond 32 MB)
 *   "gunzip_chunk_bytes"     compressed bytes per chunk (default 4 MiB, at least 64 KiB)
 *   "gunzip_min_file_bytes"  smaller files are inflated by one thread (default 8 MiB)
 *   "gunzip_in_flight"       chunks in flight per file, 0 (default) = half the pool's threads, at least 4
 *   "test_deflate_fail_after" / "test_inflate_fail_after"  tests: a device lane of the sink / of a reader treats the device as
 *                            failed after this many batches / runs (-1, the default: never) */
int qd_io_set_option(const char* name, int64_t value);
/* Chunks of this reader's file that were inflated speculatively and proven / inflated by the coordinator itself. */
int qd_reader_gunzip_stats(const qd_reader* reader, int64_t* parallel_chunks, int64_t* serial_chunks);
/* A whole gzip file image in memory -> its text, by the same parallel inflater (chunk_bytes 0 = the option's value).
 * stats (may be NULL): int64[5] = pieces, of them speculative and proven, inflated by the coordinator, members whose
 * trailer was checked, bit offsets tried by the block searches.  QD_ERR_FORMAT: damaged or truncated stream
 * (qd_gunzip_last_error); QD_ERR_INVALID: out_cap too small. */
int qd_gunzip_buffer(const uint8_t* comp, int64_t comp_len, int64_t chunk_bytes, uint8_t* out, int64_t out_cap,
                     int64_t* out_len, int64_t* stats);
const char* qd_gunzip_last_error(void);

/* ---- BGZF inflate on the device ------------------------------------------------------------------------
 * The gunzip inside pyFastq.FastqReader (src/Quade.py:203-214), for BGZF (bgzip) files: their blocks are
 * independent gzip members of <= 64 KiB that carry their compressed and inflated sizes, so a run of them is
 * inflated one block per GPU lane.  An inflater owns a stream and grow-only staging buffers on one device and
 * is driven by one thread at a time (the native reader gives each of its files one).
 * qd_inflater_run: comp[0..comp_len) = whole BGZF blocks back to back, out_len = the sum of their ISIZE
 * fields; blocking.  Every block's CRC32 and length are checked on the host.  QD_ERR_FORMAT: not whole BGZF
 * blocks, sizes inconsistent, or a block did not inflate / check (*bad_block = its index, else -1) -- the
 * caller may inflate that run itself (the reader does).  No reference counterpart beyond the gunzip. */
typedef struct qd_inflater qd_inflater;
int qd_inflater_create(int device_id, qd_inflater** out);
int qd_inflater_run(qd_inflater* inflater, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len,
                    int32_t* bad_block);
/* The same with `out` in page-locked memory from qd_pinned_alloc: the text is copied from the device straight
 * into it (no staging copy on the host). */
int qd_inflater_run_pinned(qd_inflater* inflater, const uint8_t* comp, int64_t comp_len, uint8_t* out, int64_t out_len,
                           int32_t* bad_block);
void* qd_pinned_alloc(int64_t bytes); /* NULL on failure */
void qd_pinned_free(void* p);
/* ABI v4 (form 3: v6).  Which kernel inflates the blocks: 3 (the default) = one LANE decodes a block's symbols once into tokens, 64 blocks
 * per wave, then a workgroup per block resolves the tokens (quade_inflate3.hip); 2 = 1 024 lanes per block (spans decoded from guessed starts
 * that synchronise, matches resolved by pointer jumping: quade_inflate.hip); 1 = one wave per block, one symbol after the other (r02).  Same results, same status codes.
 * The environment variable QUADE_INFLATE_FORM sets the form new inflaters start with. */
int qd_inflater_set_form(qd_inflater* inflater, int32_t form);
int qd_inflater_destroy(qd_inflater* inflater);
const char* qd_inflater_last_error(const qd_inflater* inflater);
/* A reader whose BGZF runs go through an inflater on `device_id` (< 0: host threads, as qd_reader_open). */
int qd_reader_open_on(const char* path, int64_t batch_records, int32_t queue_depth, int32_t device_id, qd_reader** out);
/* BGZF runs of this reader inflated by the device / by host threads so far (read it after the last batch). */
int qd_reader_inflate_stats(const qd_reader* reader, int64_t* device_runs, int64_t* host_runs);

/* ---- Huffman-only gzip members on the device --------------------------------------------------------------------
 * The gzip inside FastqWriter.flush_buffers (src/FastqWriter.py:83-90 appends gzip members to the destination files),
 * for the driver's `gzip_level : -1`: a member is one dynamic-Huffman DEFLATE block of literals (a byte histogram, a
 * length-limited Huffman code, one table lookup per byte; no string matching) -- any gunzip reads it.  One workgroup
 * codes one piece of text (quade_amd/csrc/quade_deflate.hip); the CRC-32 of every piece is made by the caller.
 * A deflater owns a stream and grow-only staging buffers on one device and is driven by one thread at a time.
 * qd_deflater_run: piece i = text[i][0 .. text_len[i]) (text_pinned != 0: every text[i] is page-locked memory from
 * qd_pinned_alloc and is copied to the device without a staging copy); member i lands at out + i * out_stride
 * (ordinary memory, out_stride >= qd_huffman_member_bound(longest piece), a multiple of 4), its length in
 * member_len[i]; member_len[i] == 0: that member did not fit out_stride (make it on the host).  Blocking.
 * qd_sink_set_device_deflate: the sink's Huffman-only members are made on `device_id` while the process has
 * page-locked buffers to spare (576 x 2.75 MB, made as its six lanes come up); a piece that finds none is coded on its pool thread as before, so the
 * host and the device share the work; device_id < 0: host only (the default).  gzip_level -1 and 1 are affected (the levels
 * the device implements, qd_deflater_set_level); sinks at other levels ignore the device. */
typedef struct qd_deflater qd_deflater;
int qd_deflater_create(int device_id, qd_deflater** out);
int qd_deflater_run(qd_deflater* deflater, int32_t n_pieces, const uint8_t* const* text, const int64_t* text_len,
                    const uint32_t* crc32, int32_t text_pinned, uint8_t* out, int64_t out_stride, int64_t* member_len);
int64_t qd_huffman_member_bound(int64_t text_len);
/* ABI v4.  Which members the deflater makes: -1 (the default) = Huffman only, as above; 1 = LZ77 + dynamic Huffman, the
 * driver's `gzip_level : 1` (greedy parse, 4-byte hash of last positions + runs, 32 KiB window; the piece is coded as 64 KiB
 * dynamic-Huffman blocks joined by empty stored blocks in ONE member).  Same slots, same bound, same fallback rule
 * (member_len 0); other levels stay with the host's libdeflate / zlib: QD_ERR_INVALID. */
int qd_deflater_set_level(qd_deflater* deflater, int32_t level);
int qd_deflater_destroy(qd_deflater* deflater);
const char* qd_deflater_last_error(const qd_deflater* deflater);
int qd_sink_set_device_deflate(qd_sink* sink, int32_t device_id);
int qd_sink_device_members(qd_sink* sink, int64_t* device_members); /* of qd_sink_stats' members: made on the device */

/* ---- ordinary gzip files on the device (ABI v6) ----------------------------------------------------------------------
 * What it replaces: the gunzip inside pyFastq.FastqReader for the reference's real input format -- src/Quade.py:203-206, 234-236 open
 * plain .fastq.gz files, and its fixtures test/dataset/[*].fastq.gz are single gzip members.  The pipeline (qd_pipe_run) feeds such
 * files through these kernels; this entry point inflates a whole file image for tests and measurements: gz[0 .. gz_len) = one or more
 * gzip members -> out (at most out_cap bytes), *out_len = the text's size.  Every member's CRC-32 and ISIZE are checked.  step_bytes:
 * compressed bytes per step (what the pipeline takes per batch and stream); stretch_bytes / unit_text: 0 = the defaults (16 KiB of
 * compressed bytes per probed stretch, at most 2 MiB of text per resolving workgroup).  stats (may be NULL): int64[8] = members, steps,
 * stretches probed, units decoded, units dropped (a false block start: their predecessor ran through them), steps that ended inside
 * a block, decodes done again without the probe's text filter (the stream is not text), 0.
 * QD_ERR_FORMAT: not gzip, damaged, or beyond what the device decodes (the pipeline then hands the file to the host's inflater). */
int qd_dev_gunzip(int device_id, const uint8_t* gz, int64_t gz_len, uint8_t* out, int64_t out_cap, int64_t* out_len, int64_t step_bytes,
                  int64_t stretch_bytes, int64_t unit_text, int64_t* stats);

/* ---- device buffers kept across pipelines (ABI v6) -------------------------------------------------------------------
 * No counterpart in the reference (its buffers are Python objects).  The large device buffers of a pipeline -- windows, token slots,
 * upload rings -- go to a per-device free list of the process when the pipeline is destroyed, and the next pipeline is served from
 * it: hipMalloc of ~25 GB takes between 50 ms and over a second on this pool's boxes, which a process that runs several jobs pays
 * once.  The list holds at most QUADE_POOL_GB gigabytes (environment; default 64, 0 = no list); page-locked host buffers (the
 * feeders' read buffers, the collector's slabs) have a list of their own (QUADE_POOL_PINNED_GB, default 4).  qd_pool_trim gives
 * everything on both lists back to the driver (memory in use by live objects is not touched); always QD_OK. */
int qd_pool_trim(void);

/* ---- device-resident chunk pipeline (ABI v5) ---------------------------------------------------------------------
 * Replaces, for whole chunks, the per-pair loop of Quade.double_index_parser / simple_index_parser and everything it
 * calls (src/Quade.py:195-254: four FastqReader.next(), slice + fuse, Sample.FINDER; src/FastqWriter.py:48-90: name tag,
 * record format, gzip append) with the fastq TEXT staying on the device between the stages: BGZF blocks are inflated
 * there, records are found (a record whose sequence and quality lengths differ is dropped inside its own stream,
 * SURVEY.md F6), pairs are formed in lock step (the chunk ends with its first exhausted stream, Quade.py:223-224), index
 * rows are packed and matched (the context's counters move), records are scattered by routing code in input order,
 * formatted ("@name:IDX[:MOL]", FastqWriter.py:61-69), CRC-32'd and coded into gzip members; only compressed bytes
 * cross PCIe.  Inputs the device cannot inflate (ordinary gzip members, plain text) are inflated by the host's readers
 * and join the same path as text.  The sink must be at gzip_level 1 or -1 (the levels the device codes).
 * qd_pipe_run processes the chunks in order (input for the next chunks is read ahead) and returns when every member is
 * in its file; it is driven by one thread, the context must not be used by another meanwhile.  begin_message /
 * end_message (may be NULL) are printed to stdout when a chunk starts / when its last record is in its files. */
typedef struct qd_pipe qd_pipe;
typedef struct qd_pipe_chunk {
    const char* r1;  /* seq_R1 file of the chunk (src/Quade.py:119) */
    const char* r2;  /* seq_R2 */
    const char* i1;  /* index_R1 */
    const char* i2;  /* index_R2; NULL with a single-index plan */
    qd_sink* sink;   /* where the chunk's records go (one sink for the run, or one per chunk part directory) */
    const char* begin_message;
    const char* end_message;
    /* One part of a chunk that several ranks share (all zero: the whole chunk).  Stream s (r1, r2, i1, i2) is read from file offset
     * start_offset[s] on -- a BGZF block boundary --, the first skip_bytes[s] bytes of its text and then its first skip_kept[s] kept
     * records are dropped, and the part is over after max_pairs pairs (0: at the first exhausted stream).  qd_pipe_index gives the numbers. */
    int64_t start_offset[4];
    int64_t skip_bytes[4];
    int64_t skip_kept[4];
    int64_t max_pairs;
} qd_pipe_chunk;
typedef struct qd_pipe_stats {
    int64_t pairs, batches;
    int64_t bgzf_blocks;         /* inflated (and CRC-checked) on the device */
    int64_t host_inflated_runs;  /* runs of BGZF blocks the device refused and the host inflated */
    int64_t text_segments;       /* uploads of text inflated by the host's readers (ordinary gzip, plain files) */
    int64_t pieces;              /* gzip members made */
    int64_t host_coded_pieces;   /* of them by the host (a member that did not fit its slot on the device) */
    int64_t text_in_bytes, text_out_bytes, gzip_bytes;
    int64_t rescans;             /* window scans repeated (line table too small, host-inflated text) */
    /* wall seconds: the whole call; the driver waiting for input from the feeders, for its read-backs from the device, for an
     * output set the collector still holds, in device allocations; the collector waiting for the device, downloading, appending */
    double run_s, wait_input_s, wait_sync_s, wait_out_set_s, alloc_s, collector_wait_s, download_s, append_s;
    /* ABI v6: ordinary gzip files (the reference's input format) inflated on the device */
    int64_t gzip_steps;      /* inflate steps of such streams (one per stream and top-up) */
    int64_t gzip_units;      /* stretches decoded by a lane each, from a block start that the chain from the member's start proved */
    int64_t gzip_members;    /* members whose CRC-32 and ISIZE were checked */
    int64_t gzip_fallbacks;  /* streams the device gave up (damaged, or beyond what it decodes): the host's inflater took them over */
} qd_pipe_stats;
int qd_pipe_create(qd_ctx* ctx, qd_pipe** out); /* the context holds plan and barcodes */
/* A chunk that several ranks share ("large single chunks are split into contiguous row ranges").  The reference pairs
 * record j of every stream counted from the start of the chunk (src/Quade.py:210-221), and a dropped record shifts its stream, so a
 * rank cannot start in the middle of a file without knowing how many lines and kept records lie before.  qd_pipe_index is the first
 * pass: the BGZF file `path` is cut into world x grains_per_rank grains at block boundaries, this rank inflates its grains on the
 * device and reports, for each and for each residue of (lines before the grain) mod 4: the kept records that start in it, and how
 * many bytes into the grain the first of them starts (0xFFFFFFFF: none).  The ranks exchange these tables (any transport: they are a
 * few hundred bytes per grain), add the line counts up -- which fixes every grain's residue -- and the kept counts, and derive the
 * start_offset / skip_bytes / skip_kept / max_pairs of every rank's part (quade_amd/dist.py: plan_parts).  incomplete[q] != 0: a record
 * of the grain reaches beyond the text this rank looked at -- the caller falls back to one rank per chunk.  QD_ERR_UNSUPPORTED: the
 * file is not BGZF throughout, or a rank's share exceeds one window. */
typedef struct qd_grain_info {
    int64_t file_offset;   /* of the grain's first BGZF block */
    uint32_t n_lines;      /* newlines inside the grain */
    uint32_t kept[4];      /* kept records whose header line starts in the grain, by residue */
    uint32_t skip_bytes[4]; /* from the grain's first byte of text to its first kept record's header */
    uint32_t incomplete[4];
} qd_grain_info;
int qd_pipe_index(qd_pipe* pipe, const char* path, int32_t world, int32_t rank, int32_t grains_per_rank, qd_grain_info* out, int32_t cap,
                  int32_t* n_out);
/* "batch_pairs": pairs per batch at most (default 2 000 000; windows of text are sized from it, at most 1 GiB per stream);
 * "member_slots_bytes": device memory the member slots of one batch may take (default 12 GiB): a member holds 1 MiB of one destination's
 * text, less (down to 64 KiB) when thousands of destinations would need more slots than that;
 * "device_gunzip": 1 (default) = ordinary gzip files (one or more members that are not BGZF blocks) are inflated on the device too
 * (gz_probe / inflate3_tokens / gz_resolve / gz_windows / gz_fixup); 0 = by the host's parallel inflater, uploaded as text;
 * "inflate_form": 3 (default) = every DEFLATE symbol decoded once, one lane per block, the blocks of all four streams' uploads in one
 * launch, a workgroup per block resolves the tokens (quade_inflate3.hip); 2 = speculative spans, a launch per stream and eight uploads;
 * "inflate_streams" (form 2): 1 (default) = the BGZF inflate launches go down the compute stream one after the other; 2 = they alternate between
 * two streams of their own, so one launch's last blocks and the next one's first share the device (measured slower: DESIGN.md 4.4);
 * "test_fail_inflate_batch": tests -- the device's BGZF result of that batch is treated as refused;
 * "test_host_code_every": tests -- every k-th member is coded by the host, as one that did not fit its slot on the device would be */
int qd_pipe_set_option(qd_pipe* pipe, const char* name, int64_t value);
int qd_pipe_run(qd_pipe* pipe, const qd_pipe_chunk* chunks, int32_t n_chunks, qd_pipe_stats* stats);
const char* qd_pipe_last_error(const qd_pipe* pipe);
int qd_pipe_destroy(qd_pipe* pipe);
/* ---- the pipeline's device stages one at a time, over host buffers (bindings that hold text in memory; tests) ----------
 * qd_dev_fastq_scan: the record scan of qd_pipe_run on `device_id` (what qd_fastq_index does on the host, same rules): text
 * = whole lines from a record start on; at_eof: a last line without newline counts; need: reads shorter than this count as
 * short; line_cap: lines the scan's table holds (result[5] != 0: more than that -- result[0] says how many).  recs_out:
 * 6 uint32 per kept record (head, name_off, name_len, seq, seq_len, qual -- offsets into text), at most recs_cap records;
 * result_out: uint32[8] = n_lines, n_records, n_kept, n_short, tail_start, overflow, -, -.  Returns the kept records. */
int64_t qd_dev_fastq_scan(int device_id, const uint8_t* text, int64_t text_len, int32_t at_eof, int32_t want_names, int32_t need,
                          int64_t line_cap, uint32_t* recs_out, int64_t recs_cap, uint32_t* result_out);
/* CRC-32 (gzip's) of n bytes, made on the device from ranges of range_bytes (<= 65536) whose CRCs are combined */
int qd_dev_crc32(int device_id, const uint8_t* data, int64_t n, int64_t range_bytes, uint32_t* crc_out);
/* the stable sort by destination of the scatter stage: perm_out[k] = the pair at sorted position k; with len / offsets_out
 * (n + 1 values) also the exclusive sums of len in sorted order, i.e. where every pair's output record starts */
int qd_dev_sort_by_dest(int device_id, const uint16_t* dest, int64_t n, int32_t n_dest, const uint32_t* len, uint32_t* perm_out,
                        uint32_t* offsets_out);
/* what a context was made with / holds (the pipeline reads them; bindings may too) */
int qd_get_plan(const qd_ctx* ctx, qd_plan* out);
int qd_context_device(const qd_ctx* ctx, int32_t* device_id);

#ifdef __cplusplus
}
#endif
#endif /* QUADE_HIP_H */
