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
    __device__ __forceinline__ int32_t operator()(long long key) const { return site_lookup(bitmap, base, perm, key); }
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

}  // namespace

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
    // (cpd_index_build) go through the permutation stored in the index
    IndexView v = index_carve(const_cast<void *>(index), batch, shape, n_sites);
    QueryParams p{m, r1, r2, r3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, idx};
    voxel_query_kernel<IndexLookup><<<cpd_div_up(m, 16), 256, 0, cpd_s(st)>>>(p, IndexLookup{v.bitmap, v.base, use_perm ? v.perm : nullptr});
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
