// site_index_layout.h -- memory layout of a site index (cpd_index_build / cpd_conv_outset) and the
// device lookup shared by the kernels that consume one: flags (256 B) | occupancy bitmap | per-word
// popcount prefix | scan spine | rank -> row permutation.
#pragma once
#include "common.h"

namespace {

struct IndexView {
    uint64_t *bitmap;
    uint32_t *base;
    uint32_t *bsum;
    int32_t *perm;    // rank -> row id
    int32_t *flags;   // [0]: 0 canonical, 1 perm in use, 2 external rank -> row map at [2..3] (index_order below)
    long long cells, words;
    size_t bytes;
};

static IndexView index_carve(void *mem, int batch, const int32_t shape[3], int n_cap) {
    IndexView v;
    size_t off = 0;
    char *b = (char *)mem;
    auto take = [&](size_t bytes) {
        void *p = b ? (void *)(b + off) : nullptr;
        off += cpd_align(bytes);
        return p;
    };
    v.cells = (long long)batch * shape[0] * shape[1] * shape[2];
    v.words = (v.cells + 63) / 64;
    v.flags = (int32_t *)take(256);
    v.bitmap = (uint64_t *)take((size_t)v.words * 8);
    v.base = (uint32_t *)take((size_t)v.words * 4);
    v.bsum = (uint32_t *)take((size_t)scan_num_blocks(v.words) * 4);
    v.perm = (int32_t *)take((size_t)(n_cap > 0 ? n_cap : 1) * 4);
    v.bytes = off;
    return v;
}

struct Grid {
    int32_t b, d, h, w;
    __host__ __device__ long long key(int bi, int z, int y, int x) const {
        return (((long long)bi * d + z) * h + y) * w + x;
    }
};

// Which rank -> row map an index uses is recorded in its own flags block and read on the device: flags[0] = 0 canonical
// (row id = rank), 1 the map stored in the index (cpd_index_build over an arbitrary-order list), 2 a caller-owned map whose
// device address sits in flags[2..3] (cpd_index_set_order: rows of a level re-ordered after the index was built).
__device__ __forceinline__ const int32_t *index_order(const int32_t *__restrict__ flags, const int32_t *__restrict__ own_perm) {
    const int f = flags[0];
    if (f == 0) return nullptr;
    if (f == 1) return own_perm;
    return *reinterpret_cast<const int32_t *const *>(flags + 2);
}

__device__ __forceinline__ int32_t site_lookup(const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                                               const int32_t *__restrict__ perm, long long key) {
    uint64_t w = bitmap[key >> 6];
    uint64_t bit = 1ull << (key & 63);
    if (!(w & bit)) return -1;
    int32_t r = (int32_t)(base[key >> 6] + __popcll(w & (bit - 1ull)));
    return perm ? perm[r] : r;
}

}  // namespace
