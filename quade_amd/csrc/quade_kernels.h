// Kernel parameter block and launch entry points (internal to libquade_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>

#define QD_CODE_UNDET 0xFFFFu
#define QD_MAX_KEY_BYTES 32

#ifndef QD_FAST_BLOCK
#define QD_FAST_BLOCK 512  /* default threads per workgroup, fast kernel */
#endif
#ifndef QD_FAST_UNITS
#define QD_FAST_UNITS 1    /* 2-pair units per lane per tile          */
#endif
#ifndef QD_FAST_NT
#define QD_FAST_NT 1       /* non-temporal row loads                   */
#endif
#ifndef QD_FASTX_NT
#define QD_FASTX_NT QD_FAST_NT /* exact-width rows: their lines are touched by two load instructions */
#endif
#ifndef QD_FASTX_TIGHT
#define QD_FASTX_TIGHT 1   /* exact-width rows: a lane's second 16-byte block ends with its second row */
#endif
#ifndef QD_FAST_RUNS
#define QD_FAST_RUNS 4     /* R > 0: a wave walks R * 128 consecutive pairs in R steps (see demux_fast)   */
#endif
#ifndef QD_FAST_SMALL_BATCH
#define QD_FAST_SMALL_BATCH (16ll << 20) /* pairs: at most this many -> 256-thread workgroups               */
#endif
#ifndef QD_GENERIC_SPECIAL
#define QD_GENERIC_SPECIAL 1 /* full-length batches on the generic path run its specialised forms (A/B: 0 = the catch-all) */
#endif
#ifndef QD_GENERIC_ALIGNED
#define QD_GENERIC_ALIGNED 1 /* the specialised generic forms load 4- / 8-byte aligned slices without shifts (A/B: 0)       */
#endif
#ifndef QD_WIDE_PREFETCH
#define QD_WIDE_PREFETCH 0 /* A/B: register double buffering for the static wide shapes */
#endif
#ifndef QD_GENERIC_LDS_TABLE
#define QD_GENERIC_LDS_TABLE (24 * 1024) /* specialised generic kernels: the table of every barcode is staged in LDS while histogram + table fit this many bytes (A/B: 0 = global memory, r02 / early r03) */
#endif
#ifndef QD_STRIPS_LDS_BUDGET
#define QD_STRIPS_LDS_BUDGET 0 /* A/B: LDS a CU may spend on two workgroups incl. their code strips (0: small tables only) */
#endif
#ifndef QD_SLOT_FACTOR
#define QD_SLOT_FACTOR 4 /* open-addressing slots per barcode, before rounding up to a power of two (load <= 1/4) */
#endif
#ifndef QD_STATIC80_ALWAYS
#define QD_STATIC80_ALWAYS 1 /* the static 8+0 shape for small tables too (0: large tables only, r02's choice)     */
#endif
#ifndef QD_FAST_CODE_STRIPS
#define QD_FAST_CODE_STRIPS 1 /* with runs: codes leave through the wave's LDS strip, 16 B per lane          */
#endif
#ifndef QD_MOL_RUN_STRIPS
#define QD_MOL_RUN_STRIPS 1 /* with runs: the molecular bytes of a whole run leave in one burst (else: per step)   */
#endif
#ifndef QD_FAST_GRID_FILLS
#define QD_FAST_GRID_FILLS 1 /* automatic grid = a whole number of device fills                     */
#endif
#ifndef QD_FAST_PREFETCH
#define QD_FAST_PREFETCH 1 /* register double buffering of tiles       */
#endif
#ifndef QD_FAST_MINWAVES
#define QD_FAST_MINWAVES 0 /* 2nd argument of __launch_bounds__ (waves per SIMD); 0 = unset */
#endif
#ifndef QD_FAST_PROBE2
#define QD_FAST_PROBE2 1 /* probe the table for a lane's two pairs in lockstep (full tiles) */
#endif
#ifndef QD_FAST_WT_STORES
#define QD_FAST_WT_STORES 1 /* write-through (sc1) output stores */
#endif
#ifndef QD_FASTX_PREFETCH_MAXNL
#define QD_FASTX_PREFETCH_MAXNL 2 /* RowsX: double-buffer when the seq rows take <= this many 16-byte loads per lane */
#endif
#ifndef QD_FAST_BIG_LDS
#define QD_FAST_BIG_LDS (32 * 1024) /* LDS image above which 1024-thread workgroups are used */
#endif
#define QD_GEN_BLOCK 256

// counter rows the kernels flush into: 32-bit (folded into 64-bit totals by the host), or 64-bit for A/B builds
#ifdef QD_ROWS64
typedef unsigned long long qd_row_t;
#else
typedef uint32_t qd_row_t;
#endif

struct DemuxParams {
    // packed rows (device)
    const uint8_t* seq[2];
    const uint8_t* qual[2];
    const uint8_t* len[2];
    // outputs (device)
    uint16_t* codes;
    uint8_t* mol;
    qd_row_t* partial;  // [partial_rows][cnt_stride] per-workgroup counter rows (folded into 64-bit totals by the host)
    uint64_t* adjust;   // [cnt_stride] the 64-bit totals: demux_fixup moves counts here
    // barcode table (device, global memory)
    const uint32_t* slots;              // [slot_mask+1]  (fingerprint<<16 | ordinal), 0xFFFFFFFF empty
    const uint64_t* bk16;     // [S][2]  canonical keys of the barcodes whose length == K
    const uint64_t* bk32;     // [S][4]  canonical keys of every barcode (generic kernel)
    const uint8_t* blen;                // [S]     barcode lengths
    int64_t n;
    uint32_t slot_mask, seed, n_samples, cnt_stride, partial_rows;
    uint32_t lds_bk_off, lds_hist_off;
    uint32_t mol_strip_off;  // LDS offset of the per-wave molecular staging strips, 0 = not used
    uint32_t mol_run_strips; // 1: the molecular strips hold a whole wave run (R x 128 x M bytes per wave), flushed by its last step; set by the launcher
    uint32_t code_strip_off; // LDS offset of the per-wave code strips (wave runs), 0 = one dword store per step; set by the launcher
    uint32_t thr;  // minimal_qual + 33, compared with raw quality bytes
    int32_t n_streams, K, M;
    int32_t seq_stride[2], qual_stride[2];
    int32_t idx_off[2], idx_w[2];  // byte offset of the barcode slice inside a seq row, width
    int32_t mol_off[2], mol_w[2];
    int32_t idx_col[2], mol_col[2];  // absolute 0-based column of the slices (length clamping)
    uint64_t idx_mask[2], mol_mask[2];  // (1 << 8*w) - 1
    // wide plans (16 < K <= 32, slices of up to 16 bytes per index read): the fast table holds nibble-packed
    // keys (quade_common.h), a hit is confirmed against the barcode's bytes
    int32_t wide;
    uint64_t idx_mask_hi[2];  // mask of slice bytes 8..15
    const uint64_t* bkv;      // [S][4]  per barcode: slice of index read 1 (lo, hi), slice of index read 2 (lo, hi)
    // table of EVERY barcode (any length), global memory: generic kernel and the exception pairs redone
    // after a fast launch (the LDS table of the fast kernel holds the barcodes of length K only)
    const uint32_t* gslots;
    uint32_t gmask, gseed;
    // exception pairs (reads shorter than their window) redone by demux_fixup after a fast launch
    const uint32_t* exc;
    const uint8_t* exc_len[2];  // lengths of the exception pairs' reads, compact (NULL: taken from len[k][pair])
    uint32_t n_exc;
};

// Per-context (= per device, one driving thread) memo of what a kernel instantiation was told and
// asked: the dynamic-LDS attribute and the occupancy answer for the last LDS size.
struct QdKernelCache {
    struct Entry {
        bool attr_set = false;
        size_t occ_lds = ~(size_t)0;
        int occ_blocks = 1;
    };
    std::map<const void*, Entry> entries;
};

// wg_per_cu <= 0: automatic (see launch_fast_t); block_override: 0 = automatic, else 256/512/1024
hipError_t qd_launch_fast(const DemuxParams& p, QdKernelCache& cache, int cus, int wg_per_cu, int block_override,
                          size_t lds_bytes, size_t strip_bytes_per_wave, hipStream_t st);
hipError_t qd_launch_generic(const DemuxParams& p, int grid, hipStream_t st);
hipError_t qd_launch_fixup(const DemuxParams& p, hipStream_t st);
hipError_t qd_launch_reduce(const qd_row_t* partial, uint32_t rows, uint32_t cnt_stride, uint32_t ncnt,
                            const uint64_t* base, uint64_t* out, hipStream_t st);
