// Device allocations that outlive the pipeline that made them (r05).
//
// hipMalloc of the pipeline's buffers -- ~25 GB for a batch of 2 M pairs: windows, token slots, records, coder output -- takes 50 ms
// on one box and over a second on the next (profiles/r05_cfg_ab.txt: alloc_s 0.05 ... 1.2 of runs that otherwise last 0.5 s).
// A process that runs several jobs (a service; bench.py's repeated runs) pays it once: buffers released by a pipeline go to a
// per-device free list and the next pipeline's requests are served from it.  What the list holds is bounded (QUADE_POOL_GB,
// default 64; 0: no pool) and qd_pool_trim() gives all of it back to the driver.
// Nothing here is on a kernel's path; the reference has no counterpart (its buffers are Python objects).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// an allocation of at least `want` bytes: *cap says how large it is
hipError_t qd_pool_get(size_t want, void** p, size_t* cap);
// back to the list (or to the driver); the device is drained first -- nothing queued may still use it
void qd_pool_put(void* p, size_t cap);
// the same for page-locked host buffers (hipHostMalloc: the feeders' 16 MB read buffers, the collector's slabs -- pinning 200 MB
// takes tens of milliseconds at the start of every run); the list holds at most QUADE_POOL_PINNED_GB gigabytes (default 4)
hipError_t qd_pool_get_pinned(size_t want, void** p, size_t* cap);
void qd_pool_put_pinned(void* p, size_t cap);
