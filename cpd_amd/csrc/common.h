// common.h -- shared helpers for libcpd_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cpd_hip.h"

#define CPD_WAVE 64

extern thread_local int g_cpd_last_hip_error;

static inline int cpd_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_cpd_last_hip_error = (int)e;
        return CPD_ERR_LAUNCH;
    }
    return CPD_OK;
}
#define CPD_HIP_TRY(expr)                       \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) {                 \
            g_cpd_last_hip_error = (int)_e;     \
            return CPD_ERR_LAUNCH;              \
        }                                       \
    } while (0)

// Tuning / diagnostic knobs (CPD_GC_*, CPD_WGRAD_*) are read from the environment only when CPD_TUNE=1 is set:
// a production launch costs one getenv instead of fifteen.
static inline bool cpd_tuning() {
    const char *e = getenv("CPD_TUNE");
    return e && e[0] == '1';
}
static inline const char *cpd_knob(bool tuning, const char *name) { return tuning ? getenv(name) : nullptr; }

static inline hipStream_t cpd_s(cpd_stream_t s) { return (hipStream_t)s; }
// Fill of a large 16-byte-aligned buffer with one 32-bit pattern (occupancy bitmaps: 11.6 MB per frame; voxel slot tables).
// hipMemsetAsync's fill kernel reached 2.3 TB/s on the 556 MB level-0 bitmap of a 48-frame batch; a plain grid-stride loop of
// 16-byte stores runs at the write rate of the memory system (7 TB/s for that buffer).
static __global__ void __launch_bounds__(256) cpd_fill_kernel(uint4 *__restrict__ p, size_t n16, uint32_t v) {
    const uint4 z = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}
static inline int cpd_fill_bytes(void *p, uint8_t byte, size_t bytes, hipStream_t s) {   // memset semantics
    if (bytes == 0) return 0;
    if (((uintptr_t)p & 15) || (bytes & 15) || bytes < (1u << 20)) return (int)hipMemsetAsync(p, byte, bytes, s);
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    cpd_fill_kernel<<<(unsigned)blocks, 256, 0, s>>>(static_cast<uint4 *>(p), n16, 0x01010101u * byte);
    return (int)hipGetLastError();
}
static inline int cpd_zero_fill(void *p, size_t bytes, hipStream_t s) { return cpd_fill_bytes(p, 0, bytes, s); }

static inline size_t cpd_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int cpd_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// Device-wide exclusive scan over a virtual sequence value(i), i in [0,n), in three launches:
//   scan_reduce  : per-block sums            (blocks of SCAN_BLOCK threads x SCAN_ITEMS items)
//   scan_spine   : exclusive scan of the block sums by one workgroup, total -> *total_out
//   scan_apply   : per-item exclusive prefix handed to a consumer functor
// Thread t of a block owns SCAN_ITEMS consecutive items, so prefixes follow index order.
// ---------------------------------------------------------------------------------------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024).
// Returns the exclusive prefix; *block_total receives the sum. `sm` needs 17 uint32.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sm, uint32_t *block_total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) sm[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        uint32_t w = lane < nw ? sm[lane] : 0;
        uint32_t winc = wave_incl_scan(w);
        if (lane < nw) sm[lane] = winc - w;  // exclusive per-wave offset
        if (lane == nw - 1) sm[16] = winc;
    }
    __syncthreads();
    uint32_t res = sm[wid] + inc - v;
    *block_total = sm[16];
    __syncthreads();
    return res;
}

// Item <-> thread mapping (coalesced): a block owns SCAN_TILE consecutive items, each of its 4 waves a
// contiguous quarter, and in iteration k lane l of a wave touches item wave_base + 64*k + l -- so
// every wave access is 64 consecutive items and prefixes still follow index order.
template <class ValueFn>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(long long n, ValueFn value, uint32_t *block_sums) {
    __shared__ uint32_t sm[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wbase = (long long)blockIdx.x * SCAN_TILE + (long long)wave * (SCAN_TILE / 4);
    uint32_t s = 0;
#pragma unroll 4
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        if (i < n) s += value(i);
    }
    uint32_t tot;
    block_excl_scan(s, sm, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// (template only to give the kernel inline linkage: every translation unit carries its own copy)
template <int kUnused>
__global__ void __launch_bounds__(1024) scan_spine_kernel(uint32_t *block_sums, int nb, int32_t *total_out,
                                                          int32_t total_cap) {
    __shared__ uint32_t sm[17];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_excl_scan(v, sm, &tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) {
        int32_t t = (int32_t)carry;
        if (total_cap >= 0 && t > total_cap) t = total_cap;
        *total_out = t;
    }
}

template <class ValueFn, class ConsumeFn>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(long long n, ValueFn value, const uint32_t *block_sums,
                                                                ConsumeFn consume) {
    __shared__ uint32_t sm[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wbase = (long long)blockIdx.x * SCAN_TILE + (long long)wave * (SCAN_TILE / 4);
    // pass 1: this wave's total (values are cheap to recompute: popcounts / flag tests)
    uint32_t s = 0;
#pragma unroll 4
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        if (i < n) s += value(i);
    }
    uint32_t tot;
    const uint32_t incl = wave_incl_scan(s);
    const uint32_t wave_total = __shfl(incl, 63, 64);
    if (lane == 0) sm[wave] = wave_total;
    __syncthreads();
    uint32_t carry = block_sums[blockIdx.x];
    for (int w = 0; w < wave; ++w) carry += sm[w];
    (void)tot;
    // pass 2: in index order, 64 items at a time
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        const uint32_t v = i < n ? value(i) : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (i < n) consume(i, v, carry + inc - v);
        carry += __shfl(inc, 63, 64);
    }
}

// ---- the same scan over a sequence that is ZERO outside a few TOUCHED blocks (round 5: occupancy bitmaps -- a 160k-point frame sets
// bits in a few per cent of the 32 KB blocks of its 11.6 MB bitmap, 69 MB at config 5's 0.05 m voxels). Protocol: the caller zeroes
// block_sums[0 .. nb] (nb + 1 words), whoever writes a non-zero item of block b stores a non-zero word to block_sums[b] (any value,
// plain store), then device_scan_touched: the reduce pass computes the sums of the marked blocks only, the spine scans all nb sums
// and leaves the TOTAL at block_sums[nb], the apply pass visits only blocks whose sum is non-zero (block_sums[b + 1] != block_sums[b]).
// Items of an untouched block are NOT handed to the consumer: for a popcount prefix that is the entry of a word without bits, which
// no lookup reads (a lookup tests the word's bit first); scan_block_rank() gives the running count at the start of any block.
template <class ValueFn>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_touched_kernel(long long n, ValueFn value, uint32_t *block_sums) {
    if (block_sums[blockIdx.x] == 0) return;                          // untouched: its sum is the zero already there
    __shared__ uint32_t sm[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wbase = (long long)blockIdx.x * SCAN_TILE + (long long)wave * (SCAN_TILE / 4);
    uint32_t s = 0;
#pragma unroll 4
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        if (i < n) s += value(i);
    }
    uint32_t tot;
    block_excl_scan(s, sm, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
template <int kUnused>
__global__ void __launch_bounds__(1024) scan_spine_tail_kernel(uint32_t *block_sums, int nb, int32_t *total_out, int32_t total_cap) {
    __shared__ uint32_t sm[17];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t tot;
        uint32_t ex = block_excl_scan(v, sm, &tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        block_sums[nb] = carry;                                       // the total, so that block b's sum = block_sums[b + 1] - block_sums[b]
        if (total_out) {
            int32_t t = (int32_t)carry;
            if (total_cap >= 0 && t > total_cap) t = total_cap;
            *total_out = t;
        }
    }
}
template <class ValueFn, class ConsumeFn>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_touched_kernel(long long n, ValueFn value, const uint32_t *block_sums, ConsumeFn consume) {
    uint32_t carry = block_sums[blockIdx.x];
    if (block_sums[blockIdx.x + 1] == carry) return;                  // nothing in this block
    __shared__ uint32_t sm[17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wbase = (long long)blockIdx.x * SCAN_TILE + (long long)wave * (SCAN_TILE / 4);
    uint32_t s = 0;
#pragma unroll 4
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        if (i < n) s += value(i);
    }
    const uint32_t incl = wave_incl_scan(s);
    const uint32_t wave_total = __shfl(incl, 63, 64);
    if (lane == 0) sm[wave] = wave_total;
    __syncthreads();
    for (int w = 0; w < wave; ++w) carry += sm[w];
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long i = wbase + k * 64 + lane;
        const uint32_t v = i < n ? value(i) : 0u;
        const uint32_t inc = wave_incl_scan(v);
        if (i < n) consume(i, v, carry + inc - v);
        carry += __shfl(inc, 63, 64);
    }
}
// running count at item `i` of a touched-scan result when the items of i's block up to i are known to the caller (`before`), or
// -- the common use -- the count at the START of i's block
__device__ __forceinline__ uint32_t scan_block_start(const uint32_t *block_sums, long long i) { return block_sums[i / SCAN_TILE]; }
__device__ __forceinline__ bool scan_block_empty(const uint32_t *block_sums, long long i) {
    const long long b = i / SCAN_TILE;
    return block_sums[b + 1] == block_sums[b];
}
template <class ValueFn, class ConsumeFn>
static inline int device_scan_touched(long long n, ValueFn value, ConsumeFn consume, uint32_t *block_sums, int32_t *total_out,
                                      int32_t total_cap, hipStream_t s) {
    int nb = n <= 0 ? 1 : (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    scan_reduce_touched_kernel<<<nb, SCAN_BLOCK, 0, s>>>(n, value, block_sums);
    scan_spine_tail_kernel<0><<<1, 1024, 0, s>>>(block_sums, nb, total_out, total_cap);
    scan_apply_touched_kernel<<<nb, SCAN_BLOCK, 0, s>>>(n, value, block_sums, consume);
    return cpd_check_launch();
}

// Host driver: block_sums must hold scan_num_blocks(n) uint32. total_out (device i32) may be null;
// the total written there is clamped to total_cap when total_cap >= 0.
static inline int scan_num_blocks(long long n) { return (n <= 0 ? 1 : (int)((n + SCAN_TILE - 1) / SCAN_TILE)) + 1; }   // (+ 1: the total's slot of the touched-block scan)

template <class ValueFn, class ConsumeFn>
static inline int device_scan(long long n, ValueFn value, ConsumeFn consume, uint32_t *block_sums, int32_t *total_out,
                              int32_t total_cap, hipStream_t s) {
    int nb = scan_num_blocks(n) - 1;
    scan_reduce_kernel<<<nb, SCAN_BLOCK, 0, s>>>(n, value, block_sums);
    scan_spine_kernel<0><<<1, 1024, 0, s>>>(block_sums, nb, total_out, total_cap);
    scan_apply_kernel<<<nb, SCAN_BLOCK, 0, s>>>(n, value, block_sums, consume);
    return cpd_check_launch();
}
