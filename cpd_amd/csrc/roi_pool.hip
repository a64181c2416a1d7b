// roi_pool.hip -- RoI-head feature pooling of the two-stage CPD model:
//   * cpd_voxel2pinds        = generate_voxel2pinds                     (cpd/utils/spconv_utils.py:4-21)
//   * cpd_voxel_query        = voxel_query_kernel_stack                 (pointnet2_stack/src/voxel_query_gpu.cu:10-87)
//   * cpd_voxel_query_index  = the same query through the occupancy-bitmap site index of the sparse
//                              tensor instead of a dense (B,Z,Y,X) int32 volume (no volume to fill)
//   * cpd_group_points       = group_points_kernel_stack                (pointnet2_stack/src/group_points_gpu.cu:69-99)
//   * cpd_voxel_pool_max     = grouping + position encoding + ReLU + max-pool of
//                              NeighborVoxelSAModuleMSG.forward         (voxel_pool_modules.py:96-117), fused
// The reference runs the query with one thread per grid point walking up to 9^3 = 729 cells serially.
// Here 16 lanes share a grid point: they test 16 cells of the scan at a time, a ballot keeps the
// reference's dz, dy, dx order ("first nsample hits"), and the group stops as soon as nsample
// neighbours are found. With the bitmap index a cell test is one bit of an L2-resident word.
#include "site_index_layout.h"

// Several kernels below must reproduce fp32 torch arithmetic op for op (the anchor assigner compares IoUs
// for exact equality; points_in_boxes decides on face distances): this file is compiled with
// -ffp-contract=off (csrc/Makefile). Without OCML_BASIC_ROUNDED_OPERATIONS the __fmul_rn/__fadd_rn
// intrinsics are plain operators and do not stop contraction, and `#pragma clang fp contract(off)` at file
// scope was observed not to reach the inlined device helpers (fused area / limit_period ops in the ISA).

namespace {

__global__ void __launch_bounds__(256) v2p_scatter_kernel(const int32_t *__restrict__ idx, int n, Grid g, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 q = reinterpret_cast<const int4 *>(idx)[i];
    if ((unsigned)q.x >= (unsigned)g.b || (unsigned)q.y >= (unsigned)g.d || (unsigned)q.z >= (unsigned)g.h ||
        (unsigned)q.w >= (unsigned)g.w)
        return;
    out[g.key(q.x, q.y, q.z, q.w)] = i;
}

struct DenseLookup {
    const int32_t *vol;
    __device__ __forceinline__ int32_t operator()(long long key) const { return vol[key]; }
};
struct IndexLookup {
    const uint64_t *bitmap;
    const uint32_t *base;
    const int32_t *perm;
    const int32_t *flags;     // the index's own record of its order: flags[0] != 0 -> row id = perm[rank], else row id = rank
    __device__ __forceinline__ int32_t operator()(long long key) const {
        return site_lookup(bitmap, base, index_order(flags, perm), key);
    }
};

struct QueryParams {
    int m, r1, r2, r3, nsample;
    float radius2;
    int zr, yr, xr;
    const float *new_xyz, *xyz;
    const int32_t *new_coords;
    int32_t *idx;
};

// 16 lanes per query point, 16 query points per 256-thread block.
template <class Lookup>
__global__ void __launch_bounds__(256) voxel_query_kernel(QueryParams p, Lookup lookup) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int pt = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = pt < p.m;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int b = 0, cz = 0, cy = 0, cx = 0;
    if (live) {
        nx = p.new_xyz[3 * (size_t)pt]; ny = p.new_xyz[3 * (size_t)pt + 1]; nz = p.new_xyz[3 * (size_t)pt + 2];
        const int4 c = reinterpret_cast<const int4 *>(p.new_coords)[pt];
        b = c.x; cz = c.y; cy = c.z; cx = c.w;
    }
    const int wy = 2 * p.yr + 1, wx = 2 * p.xr + 1;
    const int ncell = (2 * p.zr + 1) * wy * wx;
    int cnt = 0, first = -1;
    for (int c0 = 0; c0 < ncell; c0 += 16) {
        const bool group_active = live && cnt < p.nsample;
        if (!__any(group_active)) break;                       // every group of the wave is finished
        int nb = -1;
        const int c = c0 + sub;
        if (group_active && c < ncell) {
            const int dz = c / (wy * wx) - p.zr, rem = c % (wy * wx);
            const int dy = rem / wx - p.yr, dx = rem % wx - p.xr;
            const int z = cz + dz, y = cy + dy, x = cx + dx;
            if (z >= 0 && z < p.r1 && y >= 0 && y < p.r2 && x >= 0 && x < p.r3) {
                const int32_t cand = lookup((((long long)b * p.r1 + z) * p.r2 + y) * p.r3 + x);
                if (cand >= 0) {
                    const float xp = p.xyz[3 * (size_t)cand], yp = p.xyz[3 * (size_t)cand + 1], zp = p.xyz[3 * (size_t)cand + 2];
                    const float d2 = (xp - nx) * (xp - nx) + (yp - ny) * (yp - ny) + (zp - nz) * (zp - nz);
                    if (!(d2 > p.radius2)) nb = cand;
                }
            }
        }
        const unsigned long long bal = __ballot(nb >= 0);
        const unsigned hits = (unsigned)((bal >> (16 * grp)) & 0xffffull);       // this group's 16 cells, scan order
        if (group_active && hits) {
            const int pos = cnt + __popc(hits & ((1u << sub) - 1u));
            if (nb >= 0 && pos < p.nsample) p.idx[(size_t)pt * p.nsample + pos] = nb;
            if (first < 0) first = __shfl(nb, 16 * grp + __ffs(hits) - 1, 64);
            cnt += __popc(hits);
        }
    }
    if (live) {
        if (cnt > p.nsample) cnt = p.nsample;
        if (cnt == 0) {
            if (sub == 0) p.idx[(size_t)pt * p.nsample] = -1;          // other slots keep the caller's zeros
        } else {
            for (int l = cnt + sub; l < p.nsample; l += 16) p.idx[(size_t)pt * p.nsample + l] = first;   // the pre-fill
        }
    }
}

// The same query for the bitmap index, ROW-WISE (round 4): the x-run [cx - xr, cx + xr] of one (dz, dy) window row is <= 32 consecutive
// bits of the occupancy bitmap -- one or two 64-bit words -- so a lane tests a whole ROW of the window per step instead of one cell:
// 81 rows instead of 729 cells at range 4, 289 instead of 4913 at range 8, and no rank lookup / coordinate load for an empty row (most
// of them: a grid point in free space walked the entire window before). The set bits of 16 rows, in (dz, dy) row order and dx order
// inside a row, are the candidates in the reference's scan order; a prefix sum of the rows' popcounts deals them to the group's 16
// lanes -- "first nsample hits" (voxel_query_gpu.cu:41-77) exactly.
__global__ void __launch_bounds__(256) voxel_query_rows_kernel(QueryParams p, IndexLookup lookup) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int pt = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = pt < p.m;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int b = 0, cz = 0, cy = 0, cx = 0;
    if (live) {
        nx = p.new_xyz[3 * (size_t)pt]; ny = p.new_xyz[3 * (size_t)pt + 1]; nz = p.new_xyz[3 * (size_t)pt + 2];
        const int4 c = reinterpret_cast<const int4 *>(p.new_coords)[pt];
        b = c.x; cz = c.y; cy = c.z; cx = c.w;
    }
    const int wy = 2 * p.yr + 1;
    const int nrow = (2 * p.zr + 1) * wy;
    const int x0 = cx - p.xr < 0 ? 0 : cx - p.xr, x1 = cx + p.xr >= p.r3 ? p.r3 - 1 : cx + p.xr;
    const int nbits = x1 - x0 + 1;                                   // <= 32 (the launcher checks 2 xr + 1 <= 32); <= 0: nothing in range
    const int32_t *const perm = index_order(lookup.flags, lookup.perm);
    int cnt = 0, first = -1;
    for (int r0 = 0; r0 < nrow; r0 += 16) {
        const bool group_active = live && cnt < p.nsample;
        if (!__any(group_active)) break;
        uint32_t mask = 0;
        long long key0 = 0;
        const int rr = r0 + sub;
        if (group_active && rr < nrow && nbits > 0) {
            const int qz = rr / wy;
            const int z = cz + qz - p.zr, y = cy + (rr - qz * wy) - p.yr;
            if (z >= 0 && z < p.r1 && y >= 0 && y < p.r2) {
                key0 = (((long long)b * p.r1 + z) * p.r2 + y) * p.r3 + x0;
                const long long w0 = key0 >> 6;
                const int off = (int)(key0 & 63);
                uint64_t bits = lookup.bitmap[w0] >> off;
                if (off + nbits > 64) bits |= lookup.bitmap[w0 + 1] << (64 - off);
                mask = (uint32_t)bits & (nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u));
            }
        }
        // the set bits of the group's 16 rows = its candidates in scan order; they are dealt to the 16 lanes sixteen at a time (a row
        // with one site and a row with nine cost the same: what is serial is ceil(candidates / 16), not the number of non-empty rows)
        int incl = __popc(mask);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int up = __shfl_up(incl, d, 16);
            if (sub >= d) incl += up;
        }
        const int excl = incl - __popc(mask);
        const int total = __shfl(incl, 16 * grp + 15, 64);
        for (int j0 = 0; __any(group_active && j0 < total && cnt < p.nsample); j0 += 16) {
            const int j = j0 + sub;
            const bool mine = group_active && j < total && cnt < p.nsample;
            int rl = 0;                                       // the row (lane of the group) candidate j lies in: # rows whose inclusive prefix <= j
#pragma unroll
            for (int l = 0; l < 15; ++l) rl += __shfl(incl, 16 * grp + l, 64) <= j ? 1 : 0;
            const int src = 16 * grp + rl;
            uint32_t m = (uint32_t)__shfl((int)mask, src, 64);
            int k = j - __shfl(excl, src, 64);
            const long long rkey = ((long long)__shfl((int)(key0 >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)key0, src, 64);
            int nb = -1;
            if (mine) {
                int pos = 0;                                  // position of the k-th set bit of m
#pragma unroll
                for (int sft = 16; sft; sft >>= 1) {
                    const int c = __popc(m & ((1u << sft) - 1u));
                    if (k >= c) { k -= c; m >>= sft; pos += sft; }
                }
                const int32_t cand = site_lookup(lookup.bitmap, lookup.base, perm, rkey + pos);
                const float xp = p.xyz[3 * (size_t)cand], yp = p.xyz[3 * (size_t)cand + 1], zp = p.xyz[3 * (size_t)cand + 2];
                const float d2 = (xp - nx) * (xp - nx) + (yp - ny) * (yp - ny) + (zp - nz) * (zp - nz);
                if (!(d2 > p.radius2)) nb = cand;
            }
            const unsigned long long hb = __ballot(nb >= 0);
            const unsigned hits = (unsigned)((hb >> (16 * grp)) & 0xffffull);
            const int f = __shfl(nb, 16 * grp + (hits ? __ffs(hits) - 1 : 0), 64);
            if (hits) {                                       // (group-uniform; lanes past `total` of a group with hits just count)
                const int pos = cnt + __popc(hits & ((1u << sub) - 1u));
                if (nb >= 0 && pos < p.nsample) p.idx[(size_t)pt * p.nsample + pos] = nb;
                if (first < 0) first = f;
                cnt += __popc(hits);
            }
        }
    }
    if (live) {
        if (cnt > p.nsample) cnt = p.nsample;
        if (cnt == 0) {
            if (sub == 0) p.idx[(size_t)pt * p.nsample] = -1;
        } else {
            for (int l = cnt + sub; l < p.nsample; l += 16) p.idx[(size_t)pt * p.nsample + l] = first;
        }
    }
}

// ... and for a VOXEL level, whose point coordinates are by construction its cells' centres (get_voxel_centers, common_utils.py:66-82:
// (index + 0.5) * cell + origin, three fp32 operations per axis): the kernel evaluates the reference's distance test from the cell
// coordinates -- same operations, same order, this file is compiled without contraction -- instead of loading xyz[candidate], so a
// candidate costs no memory access until it is a HIT (rank -> row). Rounding is monotonic: d2 = fl(fl(a + b) + c) >= fl(b + c) for a >= 0,
// so a window row whose (dy, dz) part alone exceeds radius^2 is skipped before its bitmap words are fetched; and only the rows the ball can
// reach at all are dealt to the lanes (<= 6 of 25 at range 2 / radius = one cell, <= 15 of 81 at range 4 / radius = two cells -- the shipped
// yaml's pairs: ONE 16-row step instead of two / six). A lane writes the hits of its own row at the positions a prefix sum gives.
struct CellGeom { float sx, sy, sz, ox, oy, oz; };
// -> [lo, hi]: the cells c of [c0 - range, c0 + range] n [0, n) whose centre (c + 0.5) * s + o can lie within `radius` of q: a SUPERSET
// (1e-3 cells of slack, far above the fp32 error of this expression for any grid CPD has; the exact test runs on every row / cell kept)
__device__ __forceinline__ void cells_within(float q, float radius, float s, float o, int c0, int range, int n, int &lo, int &hi) {
    const float inv = 1.0f / s;
    const float a = (q - radius - o) * inv - 0.5f - 1e-3f, b = (q + radius - o) * inv - 0.5f + 1e-3f;
    lo = c0 - range < 0 ? 0 : c0 - range;
    hi = c0 + range >= n ? n - 1 : c0 + range;
    if (a > (float)lo) lo = (int)ceilf(a);          // (a, b beyond the int range only for coordinates no grid has; the comparisons guard them)
    if (b < (float)hi) hi = (int)floorf(b);
}
__global__ void __launch_bounds__(256) voxel_query_grid_kernel(QueryParams p, IndexLookup lookup, CellGeom cg, float radius) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int pt = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
    const bool live = pt < p.m;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    int b = 0, cz = 0, cy = 0, cx = 0;
    if (live) {
        nx = p.new_xyz[3 * (size_t)pt]; ny = p.new_xyz[3 * (size_t)pt + 1]; nz = p.new_xyz[3 * (size_t)pt + 2];
        const int4 c = reinterpret_cast<const int4 *>(p.new_coords)[pt];
        b = c.x; cz = c.y; cy = c.z; cx = c.w;
    }
    // the part of the window the ball can reach: rows [zlo, zhi] x [ylo, yhi] in (z, y) order -- the reference's scan order restricted to
    // them --, cells [xlo, xhi] of each (<= 2 xr + 1 <= 32 bits of the bitmap)
    int zlo, zhi, ylo, yhi, xlo, xhi;
    cells_within(nz, radius, cg.sz, cg.oz, cz, p.zr, p.r1, zlo, zhi);
    cells_within(ny, radius, cg.sy, cg.oy, cy, p.yr, p.r2, ylo, yhi);
    cells_within(nx, radius, cg.sx, cg.ox, cx, p.xr, p.r3, xlo, xhi);
    const int wy = yhi - ylo + 1, nbits = xhi - xlo + 1;
    const int nrow = (live && wy > 0 && nbits > 0 && zhi >= zlo) ? (zhi - zlo + 1) * wy : 0;
    const float inv_wy = 1.0f / (float)(wy > 0 ? wy : 1);
    const int32_t *const perm = index_order(lookup.flags, lookup.perm);
    int cnt = 0, first = -1;
    for (int r0 = 0; __any(r0 < nrow && cnt < p.nsample); r0 += 16) {
        const bool group_active = r0 < nrow && cnt < p.nsample;
        uint32_t mask = 0;                                          // this lane's row: the cells within the radius
        long long key0 = 0;
        const int rr = r0 + sub;
        if (group_active && rr < nrow) {
            const int qz = (int)(((float)rr + 0.5f) * inv_wy);      // rr / wy (exact: both far below 2^20)
            const int z = zlo + qz, y = ylo + (rr - qz * wy);
            const float yp = ((float)y + 0.5f) * cg.sy + cg.oy, zp = ((float)z + 0.5f) * cg.sz + cg.oz;
            const float b2 = (yp - ny) * (yp - ny), c2 = (zp - nz) * (zp - nz);
            if (!(b2 + c2 > p.radius2)) {
                key0 = (((long long)b * p.r1 + z) * p.r2 + y) * p.r3 + xlo;
                const long long w0 = key0 >> 6;
                const int off = (int)(key0 & 63);
                uint64_t bits = lookup.bitmap[w0] >> off;
                if (off + nbits > 64) bits |= lookup.bitmap[w0 + 1] << (64 - off);
                uint32_t occ = (uint32_t)bits & (nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u));
                while (occ) {                                       // the reference's test on every occupied cell of the row
                    const int pos = __ffs(occ) - 1;
                    occ &= occ - 1u;
                    const float xp = ((float)(xlo + pos) + 0.5f) * cg.sx + cg.ox;
                    const float d2 = (xp - nx) * (xp - nx) + b2 + c2;
                    if (!(d2 > p.radius2)) mask |= 1u << pos;
                }
            }
        }
        int incl = __popc(mask);                                    // hits before this row, in scan order
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int up = __shfl_up(incl, d, 16);
            if (sub >= d) incl += up;
        }
        const int total = __shfl(incl, 16 * grp + 15, 64);
        int at = cnt + incl - __popc(mask), fc = -1;
        while (mask && at < p.nsample) {                            // a lane writes its own row's hits: rank -> row id
            const int pos = __ffs(mask) - 1;
            mask &= mask - 1u;
            const int32_t cand = site_lookup(lookup.bitmap, lookup.base, perm, key0 + pos);
            p.idx[(size_t)pt * p.nsample + at] = cand;
            if (fc < 0) fc = cand;
            ++at;
        }
        const unsigned long long hb = __ballot(fc >= 0);
        const unsigned rows = (unsigned)((hb >> (16 * grp)) & 0xffffull);
        const int f = __shfl(fc, 16 * grp + (rows ? __ffs(rows) - 1 : 0), 64);
        if (group_active) {
            if (first < 0 && rows) first = f;
            cnt += total;
        }
    }
    if (live) {
        if (cnt > p.nsample) cnt = p.nsample;
        if (cnt == 0) {
            if (sub == 0) p.idx[(size_t)pt * p.nsample] = -1;
        } else {
            for (int l = cnt + sub; l < p.nsample; l += 16) p.idx[(size_t)pt * p.nsample + l] = first;
        }
    }
}

__global__ void __launch_bounds__(256) group_points_kernel(int nb, int m, int c, int nsample, const float *__restrict__ feat,
                                                           const int32_t *__restrict__ feat_cnt, const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ idx_cnt, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * c * nsample) return;
    const int s = (int)(i % nsample), ci = (int)((i / nsample) % c), pt = (int)(i / nsample / c);
    int bs = 0, cnt = idx_cnt[0];
    for (int k = 1; k < nb; ++k) {
        if (pt < cnt) break;
        cnt += idx_cnt[k];
        bs = k;
    }
    long long start = 0;
    for (int k = 0; k < bs; ++k) start += feat_cnt[k];
    out[i] = feat[(start + idx[(size_t)pt * nsample + s]) * c + ci];
}

// group_points_grad_kernel_stack (group_points_gpu.cu:9-36): grad_features[start + idx[pt][s]][c] += grad_out[pt][c][s].
// Thread = (pt, s, c) with c fastest, so the atomics of a wave land on consecutive floats of one feature row.
__global__ void __launch_bounds__(256) group_points_grad_kernel(int nb, int m, int c, int nsample, const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ feat_cnt, const int32_t *__restrict__ idx,
                                                                const int32_t *__restrict__ idx_cnt, float *__restrict__ grad_feat) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * c * nsample) return;
    const int ci = (int)(i % c), s = (int)((i / c) % nsample), pt = (int)(i / c / nsample);
    int bs = 0, cnt = idx_cnt[0];
    for (int k = 1; k < nb; ++k) {
        if (pt < cnt) break;
        cnt += idx_cnt[k];
        bs = k;
    }
    long long start = 0;
    for (int k = 0; k < bs; ++k) start += feat_cnt[k];
    atomicAdd(&grad_feat[(start + idx[(size_t)pt * nsample + s]) * c + ci], grad_out[((size_t)pt * c + ci) * nsample + s]);
}

// out[m][ch] = max_s relu(fin[idx[m][s]][ch] + (xyz[idx[m][s]] - new_xyz[m]) . Wpos[:, ch] + bpos[ch]);
// an empty ball (idx[m][0] < 0) gives relu(bpos[ch]) -- grouped features and offsets are zeroed
// (voxel_pool_modules.py:99,105) before the position MLP. One thread per (m, ch).
__global__ void __launch_bounds__(256) voxel_pool_max_kernel(int m, int c, int nsample, const float *__restrict__ fin, int fin_ld,
                                                             const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                             const int32_t *__restrict__ idx, const float *__restrict__ wpos,
                                                             const float *__restrict__ bpos, float *__restrict__ out, int out_ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * c) return;
    const int pt = (int)(i / c), ch = (int)(i - (long long)pt * c);
    const int32_t *id = idx + (size_t)pt * nsample;
    const float w0 = wpos[ch], w1 = wpos[c + ch], w2 = wpos[2 * c + ch], b0 = bpos[ch];
    float best;
    if (id[0] < 0) {
        best = b0 > 0.f ? b0 : 0.f;
    } else {
        const float nx = new_xyz[3 * (size_t)pt], ny = new_xyz[3 * (size_t)pt + 1], nz = new_xyz[3 * (size_t)pt + 2];
        best = 0.f;                                            // ReLU output is >= 0
        for (int s = 0; s < nsample; ++s) {
            const int32_t j = id[s];
            const float dx = xyz[3 * (size_t)j] - nx, dy = xyz[3 * (size_t)j + 1] - ny, dz = xyz[3 * (size_t)j + 2] - nz;
            const float pos = ((dx * w0 + dy * w1) + dz * w2) + b0;
            const float v = fin[(size_t)j * fin_ld + ch] + pos;
            best = v > best ? v : best;
        }
    }
    out[(size_t)pt * out_ld + ch] = best;
}

// ---- dataloader pre-filter --------------------------------------------------------
// ... followed by the module's output MLP (mlps_out: 1 x 1 conv + eval BatchNorm + ReLU, voxel_pool_modules.py:118-121) in the same
// kernel: out[m][co] = relu(sum_ch pooled[m][ch] * w_out[ch][co] + t_out[co]) with the BatchNorm scale folded into w_out; the pooled
// [M, C] tensor never reaches memory (C = 16 / 32 / 64, C2 <= 2 C).
template <int C, int NCOL>
__global__ void __launch_bounds__(256) voxel_pool_max_mlp_kernel(int m, int nsample, const float *__restrict__ fin, int fin_ld,
                                                                 const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                 const int32_t *__restrict__ idx, const float *__restrict__ wpos,
                                                                 const float *__restrict__ bpos, const float *__restrict__ wout,
                                                                 const float *__restrict__ tout, int c2, int relu,
                                                                 float *__restrict__ out, int out_ld, uint32_t *__restrict__ out_absmax) {
    // thread = (point slot, channel); a block walks CPD_POOL_ITERS x (256 / C) consecutive points. The pooled channels of the block's
    // points go through LDS (a broadcast read per four channels); the thread's column(s) of w_out stay in registers over the walk.
    constexpr int PPB = 256 / C, ITERS = 8;                          // NCOL columns of w_out per thread: c2 <= NCOL * C (the launcher's choice)
    __shared__ __attribute__((aligned(16))) float pooled[PPB][C];
    const int slot = threadIdx.x / C, ch = threadIdx.x - slot * C;
    float wcol[NCOL][C], tcol[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) {
        const int co = ch + q * C;
        tcol[q] = co < c2 ? tout[co] : 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) wcol[q][k] = co < c2 ? wout[k * c2 + co] : 0.f;
    }
    const float w0 = wpos[ch], w1 = wpos[C + ch], w2 = wpos[2 * C + ch], b0 = bpos[ch];
    uint32_t vmax = 0;
    for (int it = 0; it < ITERS; ++it) {
        const int pt = (blockIdx.x * ITERS + it) * PPB + slot;
        if (blockIdx.x * ITERS * PPB + it * PPB >= m) break;         // (block-uniform)
        const bool live = pt < m;
        float best = 0.f;
        if (live) {
            const int32_t *id = idx + (size_t)pt * nsample;
            if (id[0] < 0) {
                best = b0 > 0.f ? b0 : 0.f;
            } else {
                const float nx = new_xyz[3 * (size_t)pt], ny = new_xyz[3 * (size_t)pt + 1], nz = new_xyz[3 * (size_t)pt + 2];
                const int32_t j0 = id[0];
#pragma unroll 8
                for (int s = 0; s < nsample; ++s) {                 // (unrolled: the samples' loads in flight together)
                    const int32_t j = id[s];
                    // the query pre-fills every slot with the first hit (voxel_query_gpu.cu:62-66) and real hits are distinct voxels: a later
                    // slot equal to slot 0 is that pre-fill -- its value is already in `best` (round 5: a third of the gathers at 16 samples)
                    if (s > 0 && j == j0) continue;
                    const float dx = xyz[3 * (size_t)j] - nx, dy = xyz[3 * (size_t)j + 1] - ny, dz = xyz[3 * (size_t)j + 2] - nz;
                    const float pos = ((dx * w0 + dy * w1) + dz * w2) + b0;
                    const float v = fin[(size_t)j * fin_ld + ch] + pos;
                    best = v > best ? v : best;
                }
            }
        }
        __syncthreads();                                             // the previous iteration's readers are done
        pooled[slot][ch] = best;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NCOL; ++q) {
            const int co = ch + q * C;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < C; k += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(&pooled[slot][k]);
                acc += v.x * wcol[q][k];
                acc += v.y * wcol[q][k + 1];
                acc += v.z * wcol[q][k + 2];
                acc += v.w * wcol[q][k + 3];
            }
            acc += tcol[q];
            if (relu) acc = acc > 0.f ? acc : 0.f;
            if (live && co < c2) {
                out[(size_t)pt * out_ld + co] = acc;
                const uint32_t vb = __float_as_uint(acc) & 0x7fffffffu;
                vmax = vb > vmax ? vb : vmax;
            }
        }
    }
    if (out_absmax) {                                                // the range block of the rows written (cpd_gather_conv's `in_absmax` downstream)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
            vmax = t > vmax ? t : vmax;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t *slot = out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
            if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
        }
    }
}

struct RangeFlagFn {      // mask_points_by_range (common_utils.py:60-63): x, y inside the closed range
    const float *pts;
    int c;
    float x0, y0, x1, y1;
    __device__ uint32_t operator()(long long i) const {
        const float x = pts[(size_t)i * c], y = pts[(size_t)i * c + 1];
        return (x >= x0 && x <= x1 && y >= y0 && y <= y1) ? 1u : 0u;
    }
};
struct CompactRowsFn {    // stable compaction: kept row i goes to row prefix
    const float *pts;
    float *out;
    int c;
    __device__ void operator()(long long i, uint32_t flag, uint32_t prefix) const {
        if (!flag) return;
        for (int k = 0; k < c; ++k) out[(size_t)prefix * c + k] = pts[(size_t)i * c + k];
    }
};

// roiaware_pool3d_kernel.cu:23-35, 313-336: first box (in box order) whose z-extent and MARGIN-grown
// rotated rectangle contain the point. One thread per point; the boxes of a sample sit in LDS.
__global__ void __launch_bounds__(256) points_in_boxes_kernel(int boxes_num, int pts_num, const float *__restrict__ boxes,
                                                              const float *__restrict__ pts, int pts_ld, float margin,
                                                              int32_t *__restrict__ out) {
    extern __shared__ float sbox[];            // per box: cx, cy, cz, dx/2 + margin, dy/2 + margin, dz/2, cos(-rz), sin(-rz)
    const int b = blockIdx.y;
    const float *bx = boxes + (size_t)b * boxes_num * 7;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    if (pt < pts_num) {
        const float *p = pts + ((size_t)b * pts_num + pt) * pts_ld;
        x = p[0]; y = p[1]; z = p[2];
    }
    int32_t hit = -1;
    for (int k0 = 0; k0 < boxes_num; k0 += 512) {
        const int nk = min(512, boxes_num - k0);
        __syncthreads();
        for (int k = threadIdx.x; k < nk; k += blockDim.x) {
            const float *q = bx + (size_t)(k0 + k) * 7;
            float *o = sbox + 8 * k;
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            o[3] = (float)(q[3] / 2.0 + margin); o[4] = (float)(q[4] / 2.0 + margin); o[5] = (float)(q[5] / 2.0);
            o[6] = cosf(-q[6]); o[7] = sinf(-q[6]);
        }
        __syncthreads();
        if (hit < 0 && pt < pts_num) {
            for (int k = 0; k < nk; ++k) {
                const float *o = sbox + 8 * k;
                if (fabsf(z - o[2]) > o[5]) continue;
                const float sx = x - o[0], sy = y - o[1];
                const float lx = __fadd_rn(__fmul_rn(sx, o[6]), __fmul_rn(sy, -o[7]));
                const float ly = __fadd_rn(__fmul_rn(sx, o[7]), __fmul_rn(sy, o[6]));
                if (fabsf(lx) < o[3] && fabsf(ly) < o[4]) { hit = k0 + k; break; }
            }
        }
    }
    if (pt < pts_num) out[(size_t)b * pts_num + pt] = hit;
}

// ---- anchor head -----------------------------------------------------------------
// fp32 op-for-op restatement of box_utils.boxes3d_lidar_to_aligned_bev_boxes / boxes_iou_normal
// (cpd/utils/box_utils.py:238-272): the assigner compares IoUs for exact equality, so nothing here
// may be contracted or reassociated.
// correctly rounded fp32 quotient: the double quotient of two floats rounds to the same float as the
// exact quotient does (53 >= 2*24 + 2 bits), independent of how the compiler lowers a float division
__device__ __forceinline__ float div_rn(float x, float y) { return (float)((double)x / (double)y); }
__device__ __forceinline__ float limit_period_f(float val, float offset, float period) {
    return __fsub_rn(val, __fmul_rn(floorf(__fadd_rn(div_rn(val, period), offset)), period));
}
__device__ __forceinline__ float4 aligned_bev(const float *b) {
    const float rot = fabsf(limit_period_f(b[6], 0.5f, 3.14159265358979323846f));
    const float quarter = (float)(3.14159265358979323846 / 4);
    const float d0 = rot < quarter ? b[3] : b[4], d1 = rot < quarter ? b[4] : b[3];
    return make_float4(__fsub_rn(b[0], div_rn(d0, 2.f)), __fsub_rn(b[1], div_rn(d1, 2.f)),
                       __fadd_rn(b[0], div_rn(d0, 2.f)), __fadd_rn(b[1], div_rn(d1, 2.f)));
}
__device__ __forceinline__ float iou_normal(const float4 a, const float4 b) {
    const float x_len = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
    const float y_len = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    const float inter = __fmul_rn(x_len, y_len);
    return div_rn(inter, fmaxf(__fsub_rn(__fadd_rn(area_a, area_b), inter), 1e-6f));
}

__global__ void __launch_bounds__(256) nearest_bev_iou_kernel(const float *__restrict__ a, int n, const float *__restrict__ b, int m,
                                                              float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * m) return;
    const int ia = (int)(i / m), ib = (int)(i - (long long)ia * m);
    out[i] = iou_normal(aligned_bev(a + 7 * (size_t)ia), aligned_bev(b + 7 * (size_t)ib));
}

// Pass 1 of the assigner: per anchor max / first argmax over the GT boxes (no n x m matrix in HBM), and the
// per-GT maximum over anchors through an order-preserving atomicMax on the float bits (IoUs are >= 0).
__global__ void __launch_bounds__(256) assign_pass1_kernel(const float *__restrict__ anchors, int n, const float *__restrict__ gt, int m,
                                                           float *__restrict__ amax_iou, int32_t *__restrict__ amax_idx,
                                                           int32_t *__restrict__ gmax_bits) {
    extern __shared__ float4 sgt[];
    for (int j = threadIdx.x; j < m; j += blockDim.x) sgt[j] = aligned_bev(gt + 7 * (size_t)j);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = aligned_bev(anchors + 7 * (size_t)i);
    float best = -1.f;
    int bj = 0;
    for (int j = 0; j < m; ++j) {
        const float v = iou_normal(a, sgt[j]);
        if (v > best) { best = v; bj = j; }                       // first maximum, like numpy argmax
        if (v > 0.f) atomicMax(&gmax_bits[j], __float_as_int(v));
    }
    amax_iou[i] = best;
    amax_idx[i] = bj;
}

// Pass 2: force-match anchors that attain some GT's maximum (exact float equality, l.178), thresholds,
// background, regression targets (ResidualCoder.encode_torch) -- axis_aligned_target_assigner.py:178-243
// with POS_FRACTION < 0.
__global__ void __launch_bounds__(256) assign_pass2_kernel(const float *__restrict__ anchors, int n, const float *__restrict__ gt, int m,
                                                           const int32_t *__restrict__ gt_classes, float matched, float unmatched,
                                                           const float *__restrict__ amax_iou, const int32_t *__restrict__ amax_idx,
                                                           const int32_t *__restrict__ gmax_bits, int32_t *__restrict__ labels,
                                                           float *__restrict__ targets, float *__restrict__ gt_ious,
                                                           int32_t *__restrict__ n_examples) {
    extern __shared__ float4 sgt[];
    float *sgm = reinterpret_cast<float *>(sgt + m);
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        sgt[j] = aligned_bev(gt + 7 * (size_t)j);
        const int bits = gmax_bits[j];
        sgm[j] = bits == 0 ? -1.f : __int_as_float(bits);        // empty_gt_mask: a GT no anchor overlaps matches nothing
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *an = anchors + 7 * (size_t)i;
    const float4 a = aligned_bev(an);
    bool forced = false;
    for (int j = 0; j < m && !forced; ++j) forced = iou_normal(a, sgt[j]) == sgm[j];
    const float iou = amax_iou[i];
    const int j = amax_idx[i];
    int32_t lab = -1;
    if (forced || iou >= matched) lab = gt_classes[j];
    if (iou < unmatched) lab = 0;
    if (forced) lab = gt_classes[j];
    labels[i] = lab;
    gt_ious[i] = iou;
    float *o = targets + 7 * (size_t)i;
    if (lab > 0) {
        const float *g = gt + 7 * (size_t)j;
        const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
        const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
        const float diag = sqrtf(__fadd_rn(__fmul_rn(dxa, dxa), __fmul_rn(dya, dya)));
        o[0] = div_rn(__fsub_rn(g[0], an[0]), diag);
        o[1] = div_rn(__fsub_rn(g[1], an[1]), diag);
        o[2] = div_rn(__fsub_rn(g[2], an[2]), dza);
        o[3] = logf(div_rn(dxg, dxa)); o[4] = logf(div_rn(dyg, dya)); o[5] = logf(div_rn(dzg, dza));
        o[6] = __fsub_rn(g[6], an[6]);
    } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = 0.f;
    }
    if (lab >= 0) atomicAdd(n_examples, 1);
}

__global__ void __launch_bounds__(256) assign_weights_kernel(int n, const int32_t *__restrict__ labels, const int32_t *n_examples,
                                                             int norm, float *__restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ne = *n_examples;
    w[i] = labels[i] > 0 ? (norm ? 1.0f / (float)(ne > 1 ? ne : 1) : 1.0f) : 0.f;
}

// ResidualCoder.decode_torch + direction classifier (anchor_head_template.py:363-376)
__global__ void __launch_bounds__(256) anchor_decode_kernel(const float *__restrict__ box_preds, const float *__restrict__ anchors,
                                                            const float *__restrict__ dir_cls, long long total, int n, int nbins,
                                                            float dir_offset, float dir_limit_offset, float period,
                                                            float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float *t = box_preds + 7 * (size_t)i, *a = anchors + 7 * (size_t)(i % n);
    float *o = out + 7 * (size_t)i;
    const float diag = sqrtf(__fadd_rn(__fmul_rn(a[3], a[3]), __fmul_rn(a[4], a[4])));
    o[0] = __fadd_rn(__fmul_rn(t[0], diag), a[0]);
    o[1] = __fadd_rn(__fmul_rn(t[1], diag), a[1]);
    o[2] = __fadd_rn(__fmul_rn(t[2], a[5]), a[2]);
    o[3] = __fmul_rn(expf(t[3]), a[3]); o[4] = __fmul_rn(expf(t[4]), a[4]); o[5] = __fmul_rn(expf(t[5]), a[5]);
    float rg = __fadd_rn(t[6], a[6]);
    if (dir_cls) {
        const float *d = dir_cls + (size_t)i * nbins;
        int lab = 0;
        for (int k = 1; k < nbins; ++k)
            if (d[k] > d[lab]) lab = k;
        const float dir_rot = limit_period_f(__fsub_rn(rg, dir_offset), dir_limit_offset, period);
        rg = __fadd_rn(__fadd_rn(dir_rot, dir_offset), __fmul_rn(period, (float)lab));
    }
    o[6] = rg;
}

}  // namespace

extern "C" int cpd_nearest_bev_iou(const float *a, int n, const float *b, int m, float *out, cpd_stream_t st) {
    if (n < 0 || m < 0 || ((long long)n * m > 0 && (!a || !b || !out))) return CPD_ERR_ARG;
    if ((long long)n * m == 0) return CPD_OK;
    nearest_bev_iou_kernel<<<cpd_div_up((long long)n * m, 256), 256, 0, cpd_s(st)>>>(a, n, b, m, out);
    return cpd_check_launch();
}

extern "C" size_t cpd_anchor_assign_workspace_bytes(int n, int m) {
    return cpd_align((size_t)(n > 0 ? n : 1) * 8) + cpd_align((size_t)(m > 0 ? m : 1) * 4 + 16);
}

extern "C" int cpd_anchor_assign(const float *anchors, int n, const float *gt, int m, const int32_t *gt_classes, float matched_thr,
                                 float unmatched_thr, int norm_by_num_examples, int32_t *labels, float *bbox_targets,
                                 float *reg_weights, float *gt_ious, void *workspace, size_t workspace_bytes, cpd_stream_t st) {
    if (n < 0 || m < 0 || m > 2048 || !workspace || (n > 0 && (!anchors || !labels || !bbox_targets || !reg_weights || !gt_ious)) ||
        (m > 0 && (!gt || !gt_classes)))
        return CPD_ERR_ARG;
    if (workspace_bytes < cpd_anchor_assign_workspace_bytes(n, m)) return CPD_ERR_WORKSPACE;
    if (n == 0) return CPD_OK;
    hipStream_t s = cpd_s(st);
    if (m == 0) {                                              // no GT of this class: everything is background
        CPD_HIP_TRY(hipMemsetAsync(labels, 0, (size_t)n * 4, s));
        CPD_HIP_TRY(hipMemsetAsync(bbox_targets, 0, (size_t)n * 28, s));
        CPD_HIP_TRY(hipMemsetAsync(reg_weights, 0, (size_t)n * 4, s));
        CPD_HIP_TRY(hipMemsetAsync(gt_ious, 0, (size_t)n * 4, s));
        return CPD_OK;
    }
    float *amax_iou = (float *)workspace;
    int32_t *amax_idx = (int32_t *)(amax_iou + n);
    int32_t *gmax_bits = (int32_t *)((char *)workspace + cpd_align((size_t)n * 8));
    int32_t *n_examples = gmax_bits + m;
    CPD_HIP_TRY(hipMemsetAsync(gmax_bits, 0, (size_t)(m + 1) * 4, s));
    const int nb = cpd_div_up(n, 256);
    assign_pass1_kernel<<<nb, 256, (size_t)m * 16, s>>>(anchors, n, gt, m, amax_iou, amax_idx, gmax_bits);
    assign_pass2_kernel<<<nb, 256, (size_t)m * 20, s>>>(anchors, n, gt, m, gt_classes, matched_thr, unmatched_thr, amax_iou, amax_idx,
                                                       gmax_bits, labels, bbox_targets, gt_ious, n_examples);
    assign_weights_kernel<<<nb, 256, 0, s>>>(n, labels, n_examples, norm_by_num_examples, reg_weights);
    return cpd_check_launch();
}

extern "C" int cpd_anchor_decode(const float *box_preds, const float *anchors, const float *dir_cls_preds, int batch, int n,
                                 int num_dir_bins, float dir_offset, float dir_limit_offset, float *out, cpd_stream_t st) {
    if (batch < 0 || n < 0 || ((long long)batch * n > 0 && (!box_preds || !anchors || !out)) || (dir_cls_preds && num_dir_bins <= 0))
        return CPD_ERR_ARG;
    const long long total = (long long)batch * n;
    if (total == 0) return CPD_OK;
    const float period = (float)(2 * 3.14159265358979323846 / (num_dir_bins > 0 ? num_dir_bins : 1));
    anchor_decode_kernel<<<cpd_div_up(total, 256), 256, 0, cpd_s(st)>>>(box_preds, anchors, dir_cls_preds, total, n, num_dir_bins,
                                                                       dir_offset, dir_limit_offset, period, out);
    return cpd_check_launch();
}

extern "C" size_t cpd_mask_points_workspace_bytes(int n) {
    return cpd_align((size_t)scan_num_blocks(n) * 4 + 16);
}

extern "C" int cpd_mask_points_by_range(const float *points, int n, int c, const float range_xyz[6], float *out,
                                        int32_t *n_out, void *workspace, size_t workspace_bytes, cpd_stream_t st) {
    if (n < 0 || c < 2 || !range_xyz || !n_out || !workspace || (n > 0 && (!points || !out))) return CPD_ERR_ARG;
    if (workspace_bytes < cpd_mask_points_workspace_bytes(n)) return CPD_ERR_WORKSPACE;
    if (n == 0) {
        CPD_HIP_TRY(hipMemsetAsync(n_out, 0, 4, cpd_s(st)));
        return CPD_OK;
    }
    return device_scan(n, RangeFlagFn{points, c, range_xyz[0], range_xyz[1], range_xyz[3], range_xyz[4]},
                       CompactRowsFn{points, out, c}, (uint32_t *)workspace, n_out, -1, cpd_s(st));
}

// Multi-sweep merge (waymo_unsupervised_dataset.py:192-202 points_rigid_transform, 333-360 get_frame): every sweep's points go
// sweep -> world (its own pose) and world -> current frame (inverse of the current pose), each product formed in float64 from
// the float32 coordinates and rounded to float32 -- the reference's np.mat arithmetic on float32 clouds --, intensity
// (column 3) and the last column are zeroed, sweeps are concatenated in order.
#define CPD_MAX_SWEEPS 16
struct SweepPoses {
    int32_t n;
    int32_t off[CPD_MAX_SWEEPS + 1];
    double pose[CPD_MAX_SWEEPS][12];      // rows 0..2 of the 4x4 sweep -> world matrices
    double cur_inv[12];                   // rows 0..2 of inverse(current pose)
};
__device__ __forceinline__ void rigid3(const double (&m)[12], float x, float y, float z, float &ox, float &oy, float &oz) {
    // no fused multiply-add: (a*b) rounded, then added, like a scalar dgemm without FMA contraction
    const double dx = x, dy = y, dz = z;
    ox = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[0], dx), __dmul_rn(m[1], dy)), __dmul_rn(m[2], dz)), m[3]));
    oy = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[4], dx), __dmul_rn(m[5], dy)), __dmul_rn(m[6], dz)), m[7]));
    oz = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[8], dx), __dmul_rn(m[9], dy)), __dmul_rn(m[10], dz)), m[11]));
}
__global__ void __launch_bounds__(256) merge_sweeps_kernel(const float *__restrict__ pts, int n, int c, SweepPoses sp,
                                                           float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = 0;
    while (s + 1 < sp.n && i >= sp.off[s + 1]) ++s;
    const float *p = pts + (size_t)i * c;
    float x, y, z, x2, y2, z2;
    rigid3(sp.pose[s], p[0], p[1], p[2], x, y, z);
    rigid3(sp.cur_inv, x, y, z, x2, y2, z2);
    float *o = out + (size_t)i * c;
    o[0] = x2; o[1] = y2; o[2] = z2;
    for (int k = 3; k < c; ++k) o[k] = (k == 3 || k == c - 1) ? 0.f : p[k];
}

extern "C" int cpd_merge_sweeps(const float *points, const int32_t *sweep_offsets, int n_sweeps, int c, const double *poses,
                                const double *cur_pose_inv, float *out, cpd_stream_t st) {
    if (!sweep_offsets || n_sweeps <= 0 || c < 4 || !poses || !cur_pose_inv) return CPD_ERR_ARG;
    if (n_sweeps > CPD_MAX_SWEEPS) return CPD_ERR_UNSUPPORTED;
    SweepPoses sp;
    sp.n = n_sweeps;
    for (int s = 0; s <= n_sweeps; ++s) {
        sp.off[s] = sweep_offsets[s];
        if (sweep_offsets[s] < 0 || (s > 0 && sweep_offsets[s] < sweep_offsets[s - 1])) return CPD_ERR_ARG;
    }
    if (sweep_offsets[0] != 0) return CPD_ERR_ARG;
    for (int s = 0; s < n_sweeps; ++s)
        for (int k = 0; k < 12; ++k) sp.pose[s][k] = poses[(size_t)s * 16 + k];
    for (int k = 0; k < 12; ++k) sp.cur_inv[k] = cur_pose_inv[k];
    const int n = sweep_offsets[n_sweeps];
    if (n == 0) return CPD_OK;
    if (!points || !out) return CPD_ERR_ARG;
    merge_sweeps_kernel<<<cpd_div_up(n, 256), 256, 0, cpd_s(st)>>>(points, n, c, sp, out);
    return cpd_check_launch();
}

// ---- prototype box crop (waymo_unsupervised_dataset.py:205-331 sample_prototype_cpu) ------------------------------------
// check_pt_in_box3d_cpu (roiaware_pool3d.cpp:128-140): |z - cz| > dz / 2.0 rejects; the rectangle test compares
// fabs(local) with d / 2.0 + MARGIN in DOUBLE (MARGIN = (float)1e-2 promoted), local coordinates from fp32
// lidar_to_local_coords_cpu (cos / sin of -rz, no contraction).
__device__ __forceinline__ bool pt_in_box_cpu(float x, float y, float z, const float *q, float ca, float sa) {
    // q = cx, cy, cz, dx, dy, dz; ca / sa = cos / sin(-rz)
    if ((double)fabsf(z - q[2]) > (double)q[5] / 2.0) return false;
    const float sx = x - q[0], sy = y - q[1];
    const float lx = __fadd_rn(__fmul_rn(sx, ca), __fmul_rn(sy, -sa));
    const float ly = __fadd_rn(__fmul_rn(sx, sa), __fmul_rn(sy, ca));
    const double margin = (double)1e-2f;
    return (double)fabsf(lx) < (double)q[3] / 2.0 + margin && (double)fabsf(ly) < (double)q[4] / 2.0 + margin;
}

// roiaware_pool3d_utils.points_in_boxes_cpu: out[box][point] = 1 / 0
__global__ void __launch_bounds__(256) points_in_boxes_mask_kernel(const float *__restrict__ boxes, int k, const float *__restrict__ pts,
                                                                   int n, int pts_ld, int32_t *__restrict__ out) {
    __shared__ float s_cs[2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const float *q = boxes + 7 * (size_t)b;
    // the host libm's cosf / sinf are correctly rounded in all but rare cases; so is the double routine rounded to float
    if (threadIdx.x == 0) { s_cs[0] = (float)cos((double)(-q[6])); s_cs[1] = (float)sin((double)(-q[6])); }
    __syncthreads();
    if (i >= n) return;
    const float *p = pts + (size_t)i * pts_ld;
    out[(size_t)b * n + i] = pt_in_box_cpu(p[0], p[1], p[2], q, s_cs[0], s_cs[1]) ? 1 : 0;
}

// The two retain masks of sample_prototype_cpu without the boxes x points matrix: bit 0 = the point lies in NO box
// (retain_mask_no_object), bit 1 = it lies in no box flagged `discard` (retain_mask_good_object). Boxes in LDS, 12 floats each.
__global__ void __launch_bounds__(256) crop_flags_kernel(const float *__restrict__ pts, int n, int c, const float *__restrict__ boxes,
                                                         const int32_t *__restrict__ discard, int k, uint8_t *__restrict__ flags) {
    extern __shared__ float sbox[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < n) { const float *p = pts + (size_t)i * c; x = p[0]; y = p[1]; z = p[2]; }
    bool any = false, any_discard = false;
    for (int k0 = 0; k0 < k; k0 += 512) {
        const int nk = min(512, k - k0);
        __syncthreads();
        for (int j = threadIdx.x; j < nk; j += blockDim.x) {
            const float *bq = boxes + 7 * (size_t)(k0 + j);
            for (int q = 0; q < 6; ++q) sbox[12 * j + q] = bq[q];
            sbox[12 * j + 6] = (float)cos((double)(-bq[6]));
            sbox[12 * j + 7] = (float)sin((double)(-bq[6]));
            sbox[12 * j + 8] = discard[k0 + j] ? 1.f : 0.f;
        }
        __syncthreads();
        if (i < n) {
            for (int j = 0; j < nk; ++j) {
                const float *bq = sbox + 12 * j;
                if (!pt_in_box_cpu(x, y, z, bq, bq[6], bq[7])) continue;
                any = true;
                if (bq[8] != 0.f) any_discard = true;
            }
        }
    }
    if (i < n) flags[i] = (any ? 0 : 1) | (any_discard ? 0 : 2);
}
struct ByteFlagFn {
    const uint8_t *flags;
    int bit;
    __device__ uint32_t operator()(long long i) const { return (flags[i] >> bit) & 1u; }
};

// Prototype placement (l.277-305): (x, y, z, 1) times A^T, then times B^T, both products in float64 without contraction (numpy
// matmul of a float64 cloud with float32 matrices promoted to float64); the result goes to columns 0-2 of a zeroed row, rounded
// to fp32 (the dtype the voxelizer consumes; the reference keeps float64 until its own cast).
struct TwoMats { double a[16], b[16]; };
__global__ void __launch_bounds__(256) transform_rows_kernel(const float *__restrict__ pts, int n, int ld, TwoMats m, int c_out,
                                                             float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pts + (size_t)i * ld;
    double v[4] = {(double)p[0], (double)p[1], (double)p[2], 1.0}, t[4], u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(v[0], m.a[4 * r]), __dmul_rn(v[1], m.a[4 * r + 1])), __dmul_rn(v[2], m.a[4 * r + 2])),
                         __dmul_rn(v[3], m.a[4 * r + 3]));
#pragma unroll
    for (int r = 0; r < 4; ++r)
        u[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(t[0], m.b[4 * r]), __dmul_rn(t[1], m.b[4 * r + 1])), __dmul_rn(t[2], m.b[4 * r + 2])),
                         __dmul_rn(t[3], m.b[4 * r + 3]));
    float *o = out + (size_t)i * c_out;
    o[0] = (float)u[0]; o[1] = (float)u[1]; o[2] = (float)u[2];
    for (int q = 3; q < c_out; ++q) o[q] = 0.f;
}

extern "C" int cpd_points_in_boxes_mask(const float *boxes, int k, const float *pts, int n, int pts_ld, int32_t *out,
                                        cpd_stream_t st) {
    if (k < 0 || n < 0 || pts_ld < 3 || ((size_t)k * n > 0 && (!boxes || !pts || !out))) return CPD_ERR_ARG;
    if (k == 0 || n == 0) return CPD_OK;
    points_in_boxes_mask_kernel<<<dim3(cpd_div_up(n, 256), k), 256, 0, cpd_s(st)>>>(boxes, k, pts, n, pts_ld, out);
    return cpd_check_launch();
}

extern "C" size_t cpd_crop_boxes_workspace_bytes(int n) {
    return cpd_align((size_t)(n > 0 ? n : 1)) + cpd_align((size_t)scan_num_blocks(n > 0 ? n : 1) * 4 + 16);
}

extern "C" int cpd_crop_boxes(const float *points, int n, int c, const float *boxes, const int32_t *discard, int k,
                              float *out_no_object, int32_t *n_no_object, float *out_good_object, int32_t *n_good_object,
                              void *workspace, size_t workspace_bytes, cpd_stream_t st) {
    if (n < 0 || c < 3 || k < 0 || !n_no_object || !n_good_object || !workspace || (k > 0 && (!boxes || !discard)) ||
        (n > 0 && (!points || !out_no_object || !out_good_object)))
        return CPD_ERR_ARG;
    if (workspace_bytes < cpd_crop_boxes_workspace_bytes(n)) return CPD_ERR_WORKSPACE;
    if (n == 0) {
        CPD_HIP_TRY(hipMemsetAsync(n_no_object, 0, 4, cpd_s(st)));
        CPD_HIP_TRY(hipMemsetAsync(n_good_object, 0, 4, cpd_s(st)));
        return CPD_OK;
    }
    uint8_t *flags = static_cast<uint8_t *>(workspace);
    uint32_t *scan_ws = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + cpd_align((size_t)n));
    crop_flags_kernel<<<cpd_div_up(n, 256), 256, 512 * 12 * sizeof(float), cpd_s(st)>>>(points, n, c, boxes, discard, k, flags);
    int rc = cpd_check_launch();
    if (rc != CPD_OK) return rc;
    rc = device_scan(n, ByteFlagFn{flags, 0}, CompactRowsFn{points, out_no_object, c}, scan_ws, n_no_object, -1, cpd_s(st));
    if (rc != CPD_OK) return rc;
    return device_scan(n, ByteFlagFn{flags, 1}, CompactRowsFn{points, out_good_object, c}, scan_ws, n_good_object, -1, cpd_s(st));
}

extern "C" int cpd_transform_points(const float *points, int n, int ld, const double a[16], const double b[16], int c_out,
                                    float *out, cpd_stream_t st) {
    if (n < 0 || ld < 3 || c_out < 3 || !a || !b || (n > 0 && (!points || !out))) return CPD_ERR_ARG;
    if (n == 0) return CPD_OK;
    TwoMats m;
    for (int q = 0; q < 16; ++q) { m.a[q] = a[q]; m.b[q] = b[q]; }
    transform_rows_kernel<<<cpd_div_up(n, 256), 256, 0, cpd_s(st)>>>(points, n, ld, m, c_out, out);
    return cpd_check_launch();
}

extern "C" int cpd_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts, int pts_ld,
                                   float margin, int32_t *box_idx_of_points, cpd_stream_t st) {
    if (batch < 0 || boxes_num < 0 || pts_num < 0 || pts_ld < 3 || !box_idx_of_points ||
        (batch * boxes_num > 0 && !boxes) || (batch * pts_num > 0 && !pts))
        return CPD_ERR_ARG;
    if (batch == 0 || pts_num == 0) return CPD_OK;
    points_in_boxes_kernel<<<dim3(cpd_div_up(pts_num, 256), batch), 256, 512 * 8 * sizeof(float), cpd_s(st)>>>(
        boxes_num, pts_num, boxes, pts, pts_ld, margin, box_idx_of_points);
    return cpd_check_launch();
}

extern "C" int cpd_voxel2pinds(const int32_t *indices, int n, int batch, const int32_t shape[3], int32_t *out,
                               cpd_stream_t st) {
    if (!out || n < 0 || batch <= 0 || !shape || (n > 0 && !indices)) return CPD_ERR_ARG;
    Grid g{batch, shape[0], shape[1], shape[2]};
    const long long cells = (long long)batch * shape[0] * shape[1] * shape[2];
    if (cells <= 0) return CPD_ERR_ARG;
    CPD_HIP_TRY(hipMemsetAsync(out, 0xff, (size_t)cells * 4, cpd_s(st)));
    if (n > 0) v2p_scatter_kernel<<<cpd_div_up(n, 256), 256, 0, cpd_s(st)>>>(indices, n, g, out);
    return cpd_check_launch();
}

static int query_args_ok(int m, int r1, int r2, int r3, int nsample, float radius, int zr, int yr, int xr, const void *a,
                         const void *b, const void *c, const void *d, const void *e) {
    return m >= 0 && r1 > 0 && r2 > 0 && r3 > 0 && nsample > 0 && radius >= 0.f && zr >= 0 && yr >= 0 && xr >= 0 &&
           (m == 0 || (a && b && c && d && e));
}

extern "C" int cpd_voxel_query(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range, int x_range,
                               const float *new_xyz, const float *xyz, const int32_t *new_coords,
                               const int32_t *point_indices, int32_t *idx, cpd_stream_t st) {
    if (!query_args_ok(m, r1, r2, r3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx))
        return CPD_ERR_ARG;
    if (m == 0) return CPD_OK;
    QueryParams p{m, r1, r2, r3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, idx};
    voxel_query_kernel<DenseLookup><<<cpd_div_up(m, 16), 256, 0, cpd_s(st)>>>(p, DenseLookup{point_indices});
    return cpd_check_launch();
}

extern "C" int cpd_voxel_query_index(int m, int batch, int r1, int r2, int r3, int nsample, float radius, int z_range,
                                     int y_range, int x_range, const float *new_xyz, const float *xyz,
                                     const int32_t *new_coords, const void *index, int use_perm, int n_sites, int32_t *idx,
                                     cpd_stream_t st) {
    if (!query_args_ok(m, r1, r2, r3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, index, idx) ||
        batch <= 0)
        return CPD_ERR_ARG;
    if (m == 0) return CPD_OK;
    const int32_t shape[3] = {r1, r2, r3};
    // canonical site lists (cpd_conv_outset / cpd_index_emit) have rank == row; lists in arbitrary order
    // (cpd_index_build) go through the permutation stored in the index. Which one this index is, is read from the
    // index itself on the device (flags[0], as rulebook_kernel does); `use_perm` is kept in the signature but ignored.
    (void)use_perm;
    IndexView v = index_carve(const_cast<void *>(index), batch, shape, n_sites);
    QueryParams p{m, r1, r2, r3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, idx};
    static const bool cellwise = getenv("CPD_QUERY_CELLWISE") != nullptr;       // diagnostic: round 3's cell-by-cell scan
    cpd_launch_log_note(2 * x_range + 1 <= 32 && !cellwise ? "voxel_query_rows_kernel" : "voxel_query_kernel<IndexLookup>");
    if (2 * x_range + 1 <= 32 && !cellwise)
        voxel_query_rows_kernel<<<cpd_div_up(m, 16), 256, 0, cpd_s(st)>>>(p, IndexLookup{v.bitmap, v.base, v.perm, v.flags});
    else
        voxel_query_kernel<IndexLookup><<<cpd_div_up(m, 16), 256, 0, cpd_s(st)>>>(p, IndexLookup{v.bitmap, v.base, v.perm, v.flags});
    return cpd_check_launch();
}

static int voxel_pool_max_mlp_impl(int m, int c, int nsample, const float *features_in, int features_ld, const float *xyz,
                                   const float *new_xyz, const int32_t *idx, const float *w_pos, const float *b_pos,
                                   const float *w_out, const float *t_out, int c_out, int relu, float *out, int out_ld,
                                   uint32_t *out_absmax, cpd_stream_t st) {
    if (m < 0 || c <= 0 || nsample <= 0 || c_out <= 0 || features_ld < c || out_ld < c_out || !w_pos || !b_pos || !w_out || !t_out ||
        (m > 0 && (!features_in || !xyz || !new_xyz || !idx || !out)))
        return CPD_ERR_ARG;
    if ((c != 16 && c != 32 && c != 64) || c_out > 2 * c) return CPD_ERR_UNSUPPORTED;   // (the thread keeps <= 2 columns of w_out in registers; other shapes: cpd_voxel_pool_max + a GEMM)
    if (m == 0) return CPD_OK;
    const dim3 grid((unsigned)cpd_div_up((long long)m, 8 * (256 / c)));     // ITERS x points per pass
    cpd_launch_log_note("voxel_pool_max_mlp_kernel");
#define CPD_POOL_MLP(C, N) voxel_pool_max_mlp_kernel<C, N><<<grid, 256, 0, cpd_s(st)>>>(m, nsample, features_in, features_ld, xyz, new_xyz, idx, w_pos, \
                                                                                         b_pos, w_out, t_out, c_out, relu, out, out_ld, out_absmax)
    if (c_out <= c) { if (c == 16) CPD_POOL_MLP(16, 1); else if (c == 32) CPD_POOL_MLP(32, 1); else CPD_POOL_MLP(64, 1); }
    else { if (c == 16) CPD_POOL_MLP(16, 2); else if (c == 32) CPD_POOL_MLP(32, 2); else CPD_POOL_MLP(64, 2); }
#undef CPD_POOL_MLP
    return cpd_check_launch();
}
extern "C" int cpd_voxel_pool_max_mlp(int m, int c, int nsample, const float *features_in, int features_ld, const float *xyz,
                                      const float *new_xyz, const int32_t *idx, const float *w_pos, const float *b_pos,
                                      const float *w_out, const float *t_out, int c_out, int relu, float *out, int out_ld,
                                      cpd_stream_t st) {
    return voxel_pool_max_mlp_impl(m, c, nsample, features_in, features_ld, xyz, new_xyz, idx, w_pos, b_pos, w_out, t_out, c_out, relu, out, out_ld,
                                   nullptr, st);
}
extern "C" int cpd_voxel_pool_max_mlp_ranged(int m, int c, int nsample, const float *features_in, int features_ld, const float *xyz,
                                             const float *new_xyz, const int32_t *idx, const float *w_pos, const float *b_pos,
                                             const float *w_out, const float *t_out, int c_out, int relu, float *out, int out_ld,
                                             uint32_t *out_absmax, cpd_stream_t st) {
    return voxel_pool_max_mlp_impl(m, c, nsample, features_in, features_ld, xyz, new_xyz, idx, w_pos, b_pos, w_out, t_out, c_out, relu, out, out_ld,
                                   out_absmax, st);
}

// ---- RoI grid points (round 5) ----------------------------------------------------------------------------------------------------
// VoxelRCNNHead.get_global_grid_points_of_roi + the cell coordinates roi_grid_pool derives from them (voxel_rcnn_head.py:186-273, 365-386;
// common_utils.rotate_points_along_z, common_utils.py:35-57) in ONE launch, thread = grid point: the reference's (and this library's
// earlier) torch sequence was ~25 elementwise launches over 1.7 M points per 16-frame step. fp32 operation for operation (this file is
// compiled with -ffp-contract=off):
//   local = (i + 0.5) / G * size - size / 2;  x = lx cos - ly sin, y = lx sin + ly cos, z = lz;  + centre
//   cell  = (point - range_lo) // voxel_size           (torch floor_divide on floats: c10::div_floor_floating)
//   level = cell // stride -> int32                    (the same floor division, then .int())
__device__ __forceinline__ float div_floor_f(float a, float b) {      // c10::div_floor_floating<float>
    if (b == 0.f) return __fdiv_rn(a, b);
    const float mod = fmodf(a, b);
    float div = __fdiv_rn(a - mod, b);
    if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div -= 1.f;
    float fl;
    if (div != 0.f) {
        fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.f;
    } else {
        fl = copysignf(0.f, __fdiv_rn(a, b));
    }
    return fl;
}
struct GridLevels {
    int n;
    int stride[4];
    int32_t *coords[4];           // [m][4] = (b, x, y, z) as roi_grid_pool's `cur_coords`, or (b, z, y, x) when bzyx
};
__global__ void __launch_bounds__(256) roi_grid_points_kernel(const float *__restrict__ rois, int roi_ld, int n_rois, int rois_per_frame, int g,
                                                              float vx, float vy, float vz, float lx0, float ly0, float lz0, GridLevels lv,
                                                              int bzyx, float *__restrict__ grid_xyz) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int g3 = g * g * g;
    if (tid >= (long long)n_rois * g3) return;
    const int roi = (int)(tid / g3), k = (int)(tid - (long long)roi * g3);
    const int ix = k / (g * g), iy = (k / g) % g, iz = k % g;              // new_ones(G, G, G).nonzero(): row-major triples
    const float *r = rois + (size_t)roi * roi_ld;
    const float gf = (float)g;
    const float sx = r[3], sy = r[4], sz = r[5];
    const float lx = __fdiv_rn((float)ix + 0.5f, gf) * sx - __fdiv_rn(sx, 2.f);
    const float ly = __fdiv_rn((float)iy + 0.5f, gf) * sy - __fdiv_rn(sy, 2.f);
    const float lz = __fdiv_rn((float)iz + 0.5f, gf) * sz - __fdiv_rn(sz, 2.f);
    const float ca = cosf(r[6]), sa = sinf(r[6]);
    const float px = (lx * ca - ly * sa) + r[0];
    const float py = (lx * sa + ly * ca) + r[1];
    const float pz = lz + r[2];
    float *o = grid_xyz + (size_t)tid * 3;
    o[0] = px; o[1] = py; o[2] = pz;
    const float cx = div_floor_f(px - lx0, vx), cy = div_floor_f(py - ly0, vy), cz = div_floor_f(pz - lz0, vz);
    const int b = roi / rois_per_frame;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l >= lv.n) break;
        const float st = (float)lv.stride[l];
        const int x = (int)div_floor_f(cx, st), y = (int)div_floor_f(cy, st), z = (int)div_floor_f(cz, st);
        int4 *dst = reinterpret_cast<int4 *>(lv.coords[l]) + tid;
        *dst = bzyx ? make_int4(b, z, y, x) : make_int4(b, x, y, z);
    }
}
extern "C" int cpd_roi_grid_points(const float *rois, int roi_ld, int n_rois, int rois_per_frame, int grid_size, const float voxel_size[3],
                                   const float range_lo[3], int n_levels, const int32_t *strides, int32_t *const *level_coords, int bzyx,
                                   float *grid_xyz, cpd_stream_t st) {
    if (n_rois < 0 || roi_ld < 7 || rois_per_frame <= 0 || grid_size <= 0 || grid_size > 32 || !voxel_size || !range_lo || n_levels < 0 ||
        n_levels > 4 || (n_levels > 0 && (!strides || !level_coords)) || (n_rois > 0 && (!rois || !grid_xyz)))
        return CPD_ERR_ARG;
    if (n_rois == 0) return CPD_OK;
    GridLevels lv;
    lv.n = n_levels;
    for (int l = 0; l < 4; ++l) {
        lv.stride[l] = l < n_levels ? strides[l] : 1;
        lv.coords[l] = l < n_levels ? level_coords[l] : nullptr;
        if (l < n_levels && (strides[l] <= 0 || !level_coords[l] || (((uintptr_t)level_coords[l]) & 15))) return CPD_ERR_ARG;
    }
    const long long m = (long long)n_rois * grid_size * grid_size * grid_size;
    cpd_launch_log_note("roi_grid_points_kernel");
    roi_grid_points_kernel<<<cpd_div_up(m, 256), 256, 0, cpd_s(st)>>>(rois, roi_ld, n_rois, rois_per_frame, grid_size, voxel_size[0], voxel_size[1],
                                                                       voxel_size[2], range_lo[0], range_lo[1], range_lo[2], lv, bzyx, grid_xyz);
    return cpd_check_launch();
}

extern "C" int cpd_voxel_query_index_grid(int m, int batch, int r1, int r2, int r3, int nsample, float radius, int z_range,
                                          int y_range, int x_range, const float *new_xyz, const int32_t *new_coords,
                                          const void *index, int n_sites, const float cell_xyz[3], const float origin_xyz[3],
                                          int32_t *idx, cpd_stream_t st) {
    if (!query_args_ok(m, r1, r2, r3, nsample, radius, z_range, y_range, x_range, new_xyz, new_xyz, new_coords, index, idx) ||
        batch <= 0 || !cell_xyz || !origin_xyz || !(cell_xyz[0] > 0.f) || !(cell_xyz[1] > 0.f) || !(cell_xyz[2] > 0.f))
        return CPD_ERR_ARG;
    if (2 * x_range + 1 > 32) return CPD_ERR_UNSUPPORTED;          // (a window row is one 32-bit mask; the xyz form takes any range)
    if (m == 0) return CPD_OK;
    const int32_t shape[3] = {r1, r2, r3};
    IndexView v = index_carve(const_cast<void *>(index), batch, shape, n_sites);
    QueryParams p{m, r1, r2, r3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, nullptr, new_coords, idx};
    cpd_launch_log_note("voxel_query_grid_kernel");
    voxel_query_grid_kernel<<<cpd_div_up(m, 16), 256, 0, cpd_s(st)>>>(p, IndexLookup{v.bitmap, v.base, v.perm, v.flags},
                                                                      CellGeom{cell_xyz[0], cell_xyz[1], cell_xyz[2], origin_xyz[0], origin_xyz[1], origin_xyz[2]}, radius);
    return cpd_check_launch();
}

extern "C" int cpd_group_points(int b, int m, int c, int nsample, const float *features, const int32_t *features_batch_cnt,
                                const int32_t *idx, const int32_t *idx_batch_cnt, float *out, cpd_stream_t st) {
    if (b <= 0 || m < 0 || c <= 0 || nsample <= 0 || !features_batch_cnt || !idx_batch_cnt || (m > 0 && (!features || !idx || !out)))
        return CPD_ERR_ARG;
    if (m == 0) return CPD_OK;
    const long long total = (long long)m * c * nsample;
    group_points_kernel<<<cpd_div_up(total, 256), 256, 0, cpd_s(st)>>>(b, m, c, nsample, features, features_batch_cnt, idx,
                                                                        idx_batch_cnt, out);
    return cpd_check_launch();
}

extern "C" int cpd_group_points_grad(int b, int m, int c, int nsample, int n, const float *grad_out, const int32_t *features_batch_cnt,
                                     const int32_t *idx, const int32_t *idx_batch_cnt, float *grad_features, cpd_stream_t st) {
    if (b <= 0 || m < 0 || c <= 0 || nsample <= 0 || n < 0 || !features_batch_cnt || !idx_batch_cnt || (n > 0 && !grad_features) ||
        (m > 0 && (!grad_out || !idx)))
        return CPD_ERR_ARG;
    if (n > 0) CPD_HIP_TRY(hipMemsetAsync(grad_features, 0, (size_t)n * c * sizeof(float), cpd_s(st)));
    if (m == 0 || n == 0) return CPD_OK;
    const long long total = (long long)m * c * nsample;
    group_points_grad_kernel<<<cpd_div_up(total, 256), 256, 0, cpd_s(st)>>>(b, m, c, nsample, grad_out, features_batch_cnt, idx,
                                                                             idx_batch_cnt, grad_features);
    return cpd_check_launch();
}

extern "C" int cpd_voxel_pool_max(int m, int c, int nsample, const float *features_in, int features_ld, const float *xyz,
                                  const float *new_xyz, const int32_t *idx, const float *w_pos, const float *b_pos, float *out,
                                  int out_ld, cpd_stream_t st) {
    if (m < 0 || c <= 0 || nsample <= 0 || features_ld < c || out_ld < c || !w_pos || !b_pos ||
        (m > 0 && (!features_in || !xyz || !new_xyz || !idx || !out)))
        return CPD_ERR_ARG;
    if (m == 0) return CPD_OK;
    voxel_pool_max_kernel<<<cpd_div_up((long long)m * c, 256), 256, 0, cpd_s(st)>>>(m, c, nsample, features_in, features_ld, xyz,
                                                                                   new_xyz, idx, w_pos, b_pos, out, out_ld);
    return cpd_check_launch();
}
