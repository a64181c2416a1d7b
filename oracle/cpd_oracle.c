/*
 * cpd_oracle.c -- CPU restatement of the CPD detection hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for cpd_amd's HIP kernels. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / reported CPU baseline. The product path (cpd_amd/) never calls into it.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose behaviour it
 * restates. Functions marked [SPCONV] restate the algorithm of the third-party dependency
 * spconv-cu111==2.1.22 (pinned by the reference's README.md:22,31; source NOT under
 * /root/reference, not installed here): for those, PARITY IS UNPINNED by reference execution --
 * they are anchored on the reference's call sites and validated by definition against dense
 * torch.nn.functional.conv3d (tests/test_oracle_sparse.py). Everything else is pinned against
 * golden vectors generated from the reference's own Python files (tests/golden/make_golden.py)
 * and against the reference's iou3d_cpu.cpp compiled into oracle/_ref (oracle/Makefile).
 *
 * Plain C (gcc -O2 -fopenmp). All pointers are HOST pointers. Return 0 on success, <0 on error.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define CPD_OK 0
#define CPD_ERR_ARG (-1)
#define CPD_ERR_ALLOC (-2)
#define CPD_ERR_CAPACITY (-3)

/* ------------------------------------------------------------------------------------------
 * B1: voxelizer.  [SPCONV] Point2VoxelCPU3d.point_to_voxel, called from
 * cpd/datasets/processor/data_processor.py:35-41,53 (VoxelGeneratorWrapper.generate, l.43-59).
 * Semantics per SURVEY Appendix A.1.
 * ------------------------------------------------------------------------------------------ */

/* grid_size = round((hi - lo) / vsize) in fp32, returned z,y,x.
 * (data_processor.py:130-131 does the same in numpy; spconv does it in its ctor.) */
int cpd_ref_grid_size(const float vsize_xyz[3], const float range_xyz[6], int32_t grid_zyx[3]) {
    for (int a = 0; a < 3; ++a) {
        float g = (range_xyz[3 + a] - range_xyz[a]) / vsize_xyz[a];
        grid_zyx[2 - a] = (int32_t)roundf(g);
    }
    return CPD_OK;
}

/* Serial first-appearance voxelizer. points [n, c] (x,y,z,...). Outputs are caller-allocated for
 * max_voxels; slots beyond num_points[v] are zero. coords are (z,y,x) int32.               */
int cpd_ref_voxelize(const float *points, int n, int c, const float vsize_xyz[3],
                     const float range_xyz[6], int max_points, int max_voxels, float *voxels,
                     int32_t *coords_zyx, int32_t *num_points, int32_t *n_voxels) {
    if (n < 0 || c < 3 || max_points <= 0 || max_voxels <= 0) return CPD_ERR_ARG;
    int32_t grid[3];
    cpd_ref_grid_size(vsize_xyz, range_xyz, grid);
    const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
    /* like the reference generator object, the dense lookup volume persists across calls and only
     * the touched entries are reset afterwards (A.1) */
    static int32_t *lut = NULL;
    static size_t lut_cells = 0;
    if (lut_cells != cells) {
        free(lut);
        lut = (int32_t *)malloc(cells * sizeof(int32_t));
        if (!lut) { lut_cells = 0; return CPD_ERR_ALLOC; }
        memset(lut, 0xff, cells * sizeof(int32_t)); /* -1 */
        lut_cells = cells;
    }
    memset(voxels, 0, (size_t)max_voxels * max_points * c * sizeof(float));
    memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
    int nvox = 0;
    for (int i = 0; i < n; ++i) {
        const float *pt = points + (size_t)i * c;
        int32_t cz[3];
        int ok = 1;
        for (int j = 0; j < 3; ++j) { /* j over z,y,x ; pt index 2-j */
            float f = floorf((pt[2 - j] - range_xyz[2 - j]) / vsize_xyz[2 - j]);
            /* guard the int conversion against inf/nan/huge values */
            if (!(f >= 0.0f) || !(f < (float)grid[j])) { ok = 0; break; }
            cz[j] = (int32_t)f;
        }
        if (!ok) continue;
        size_t key = ((size_t)cz[0] * grid[1] + cz[1]) * grid[2] + cz[2];
        int32_t v = lut[key];
        if (v < 0) {
            if (nvox >= max_voxels) continue;
            v = nvox++;
            lut[key] = v;
            coords_zyx[3 * v + 0] = cz[0];
            coords_zyx[3 * v + 1] = cz[1];
            coords_zyx[3 * v + 2] = cz[2];
        }
        int32_t k = num_points[v];
        if (k < max_points) {
            memcpy(voxels + ((size_t)v * max_points + k) * c, pt, (size_t)c * sizeof(float));
            num_points[v] = k + 1;
        }
    }
    for (int v = 0; v < nvox; ++v) /* reset touched entries */
        lut[((size_t)coords_zyx[3 * v] * grid[1] + coords_zyx[3 * v + 1]) * grid[2] + coords_zyx[3 * v + 2]] = -1;
    *n_voxels = nvox;
    return CPD_OK;
}

/* MeanVFE: cpd/models/backbones_3d/vfe/mean_vfe.py:41-43
 *   points_mean = voxels.sum(dim=1) / clamp_min(num_points, 1)                               */
int cpd_ref_mean_vfe(const float *voxels, const int32_t *num_points, int m, int p, int c,
                     float *out) {
    for (int v = 0; v < m; ++v) {
        float norm = (float)(num_points[v] < 1 ? 1 : num_points[v]);
        for (int ch = 0; ch < c; ++ch) {
            float s = 0.f;
            for (int k = 0; k < p; ++k) s += voxels[((size_t)v * p + k) * c + ch];
            out[(size_t)v * c + ch] = s / norm;
        }
    }
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * B2: sparse convolution.  [SPCONV] SubMConv3d / SparseConv3d / SparseConvTensor.dense(),
 * call sites cpd/models/backbones_3d/spconv_backbone.py:17-21,108-115,414-455,524-529 and
 * cpd/models/backbones_2d/map_to_bev/height_compression.py:136-138. SURVEY Appendix A.2-A.5.
 *
 * The rulebook is an output-stationary neighbour table nbr[kv][n_out] (tap-major): the input
 * row feeding output row j through tap t, or -1. Tap index t = (tz*kH + ty)*kW + tx, matching
 * the spconv-2.x weight layout (Cout, kD, kH, kW, Cin).
 * ------------------------------------------------------------------------------------------ */

static inline int64_t lin_key(int b, int z, int y, int x, const int32_t shape[3]) {
    return (((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

/* dense lookup volume: index of the active site at each cell, -1 elsewhere */
static int32_t *build_lut(const int32_t *indices, int n, int batch, const int32_t shape[3]) {
    size_t cells = (size_t)batch * shape[0] * shape[1] * shape[2];
    int32_t *lut = (int32_t *)malloc(cells * sizeof(int32_t));
    if (!lut) return NULL;
    memset(lut, 0xff, cells * sizeof(int32_t));
    for (int i = 0; i < n; ++i) {
        const int32_t *q = indices + 4 * (size_t)i;
        lut[lin_key(q[0], q[1], q[2], q[3], shape)] = i;
    }
    return lut;
}

/* SubM rulebook (A.3): output set == input set, same order; tap t reads coord p + t - k/2. */
int cpd_ref_subm_rulebook(const int32_t *indices, int n, int batch, const int32_t shape[3],
                          const int32_t ksize[3], int32_t *nbr) {
    int32_t *lut = build_lut(indices, n, batch, shape);
    if (!lut) return CPD_ERR_ALLOC;
    const int kv = ksize[0] * ksize[1] * ksize[2];
    (void)kv;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
        const int32_t *q = indices + 4 * (size_t)j;
        int t = 0;
        for (int tz = 0; tz < ksize[0]; ++tz)
            for (int ty = 0; ty < ksize[1]; ++ty)
                for (int tx = 0; tx < ksize[2]; ++tx, ++t) {
                    int z = q[1] + tz - ksize[0] / 2, y = q[2] + ty - ksize[1] / 2,
                        x = q[3] + tx - ksize[2] / 2;
                    int32_t r = -1;
                    if (z >= 0 && z < shape[0] && y >= 0 && y < shape[1] && x >= 0 && x < shape[2])
                        r = lut[lin_key(q[0], z, y, x, shape)];
                    nbr[(size_t)t * n + j] = r;
                }
    }
    free(lut);
    return CPD_OK;
}

/* out_shape[d] = (in + 2*pad - k) / stride + 1   (A.4) */
int cpd_ref_conv_out_shape(const int32_t in_shape[3], const int32_t ksize[3],
                           const int32_t stride[3], const int32_t pad[3], int32_t out_shape[3]) {
    for (int d = 0; d < 3; ++d) {
        out_shape[d] = (in_shape[d] + 2 * pad[d] - ksize[d]) / stride[d] + 1;
        if (out_shape[d] <= 0) return CPD_ERR_ARG;
    }
    return CPD_OK;
}

/* Regular sparse conv output set (A.4): o active iff exists active p and tap t with
 * p + pad - t = o*stride, 0 <= o < out_shape. Emitted in canonical ascending (b,z,y,x) order
 * (the parity contract of SURVEY Appendix C). out_indices capacity = out_cap rows.            */
int cpd_ref_conv_outset(const int32_t *in_indices, int n_in, int batch, const int32_t in_shape[3],
                        const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                        int32_t *out_indices, int out_cap, int32_t *n_out) {
    int32_t os[3];
    int rc = cpd_ref_conv_out_shape(in_shape, ksize, stride, pad, os);
    if (rc) return rc;
    size_t cells = (size_t)batch * os[0] * os[1] * os[2];
    uint8_t *mark = (uint8_t *)calloc(cells, 1);
    if (!mark) return CPD_ERR_ALLOC;
    for (int i = 0; i < n_in; ++i) {
        const int32_t *q = in_indices + 4 * (size_t)i;
        for (int tz = 0; tz < ksize[0]; ++tz) {
            int nz = q[1] + pad[0] - tz;
            if (nz < 0 || nz % stride[0]) continue;
            nz /= stride[0];
            if (nz >= os[0]) continue;
            for (int ty = 0; ty < ksize[1]; ++ty) {
                int ny = q[2] + pad[1] - ty;
                if (ny < 0 || ny % stride[1]) continue;
                ny /= stride[1];
                if (ny >= os[1]) continue;
                for (int tx = 0; tx < ksize[2]; ++tx) {
                    int nx = q[3] + pad[2] - tx;
                    if (nx < 0 || nx % stride[2]) continue;
                    nx /= stride[2];
                    if (nx >= os[2]) continue;
                    mark[lin_key(q[0], nz, ny, nx, os)] = 1;
                }
            }
        }
    }
    int cnt = 0;
    for (int b = 0; b < batch; ++b)
        for (int z = 0; z < os[0]; ++z)
            for (int y = 0; y < os[1]; ++y)
                for (int x = 0; x < os[2]; ++x)
                    if (mark[lin_key(b, z, y, x, os)]) {
                        if (cnt >= out_cap) { free(mark); return CPD_ERR_CAPACITY; }
                        int32_t *o = out_indices + 4 * (size_t)cnt++;
                        o[0] = b; o[1] = z; o[2] = y; o[3] = x;
                    }
    free(mark);
    *n_out = cnt;
    return CPD_OK;
}

/* Regular conv rulebook: nbr[t][o] = input row at o*stride - pad + t, or -1. */
int cpd_ref_conv_rulebook(const int32_t *in_indices, int n_in, const int32_t *out_indices,
                          int n_out, int batch, const int32_t in_shape[3], const int32_t ksize[3],
                          const int32_t stride[3], const int32_t pad[3], int32_t *nbr) {
    int32_t *lut = build_lut(in_indices, n_in, batch, in_shape);
    if (!lut) return CPD_ERR_ALLOC;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n_out; ++j) {
        const int32_t *o = out_indices + 4 * (size_t)j;
        int t = 0;
        for (int tz = 0; tz < ksize[0]; ++tz)
            for (int ty = 0; ty < ksize[1]; ++ty)
                for (int tx = 0; tx < ksize[2]; ++tx, ++t) {
                    int z = o[1] * stride[0] - pad[0] + tz, y = o[2] * stride[1] - pad[1] + ty,
                        x = o[3] * stride[2] - pad[2] + tx;
                    int32_t r = -1;
                    if (z >= 0 && z < in_shape[0] && y >= 0 && y < in_shape[1] && x >= 0 &&
                        x < in_shape[2])
                        r = lut[lin_key(o[0], z, y, x, in_shape)];
                    nbr[(size_t)t * n_out + j] = r;
                }
    }
    free(lut);
    return CPD_OK;
}

/* out[j] = bias + sum_t W[:, t, :] . in[nbr[t][j]]   (A.3/A.4). weight is the reference layout
 * (Cout, kv, Cin); bias may be NULL. Accumulation order: taps ascending, cin ascending.       */
int cpd_ref_sparse_conv(const float *feat_in, int c_in, const float *weight, const float *bias,
                        const int32_t *nbr, int kv, int n_out, int c_out, float *feat_out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n_out; ++j) {
        float *o = feat_out + (size_t)j * c_out;
        for (int co = 0; co < c_out; ++co) o[co] = bias ? bias[co] : 0.f;
        for (int t = 0; t < kv; ++t) {
            int32_t i = nbr[(size_t)t * n_out + j];
            if (i < 0) continue;
            const float *x = feat_in + (size_t)i * c_in;
            for (int co = 0; co < c_out; ++co) {
                const float *w = weight + ((size_t)co * kv + t) * c_in;
                float s = 0.f;
                for (int ci = 0; ci < c_in; ++ci) s += w[ci] * x[ci];
                o[co] += s;
            }
        }
    }
    return CPD_OK;
}

/* Per-channel affine (+ optional residual) (+ optional ReLU) on [n, c] rows: the eval-mode
 * BatchNorm1d + ReLU of spconv_backbone.py:29-33 and the SparseBasicBlock tail (l.120-136):
 *   y = x*scale + shift ; y += residual ; y = max(y, 0).                                      */
int cpd_ref_affine_rows(float *x, int n, int c, const float *scale, const float *shift,
                        const float *residual, int relu) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j)
        for (int ch = 0; ch < c; ++ch) {
            size_t k = (size_t)j * c + ch;
            float y = x[k];
            if (scale) y = y * scale[ch];
            if (shift) y = y + shift[ch];
            if (residual) y += residual[k];
            if (relu && y < 0.f) y = 0.f;
            x[k] = y;
        }
    return CPD_OK;
}

/* SparseConvTensor.dense() + view(N, C*D, H, W) (A.2; height_compression.py:136-138):
 * out[b][c*D + z][y][x] = feat[n][c], zeros elsewhere. out is (B, C*D, H, W) f32.           */
int cpd_ref_densify(const float *feat, const int32_t *indices, int n, int c, int batch,
                    const int32_t shape[3], float *out) {
    size_t plane = (size_t)shape[1] * shape[2];
    memset(out, 0, (size_t)batch * c * shape[0] * plane * sizeof(float));
    for (int i = 0; i < n; ++i) {
        const int32_t *q = indices + 4 * (size_t)i;
        for (int ch = 0; ch < c; ++ch)
            out[(((size_t)q[0] * c + ch) * shape[0] + q[1]) * plane + (size_t)q[2] * shape[2] +
                q[3]] = feat[(size_t)i * c + ch];
    }
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Dense BEV convs: torch Conv2d / ConvTranspose2d / BatchNorm2d(eval) / ReLU as used by
 * cpd/models/backbones_2d/base_bev_backbone.py:31-59 and cpd/models/dense_heads/center_head.py:
 * 21-27,73-80.  NCHW f32, weights in torch layout.
 * ------------------------------------------------------------------------------------------ */

/* Conv2d: in (B,Cin,H,W), w (Cout,Cin,kh,kw), bias or NULL, stride s, zero pad p.
 * out (B,Cout,Ho,Wo), Ho = (H+2p-kh)/s+1. Accumulation order: cin, ky, kx ascending.        */
int cpd_ref_conv2d(const float *in, int b, int cin, int h, int w, const float *wt,
                   const float *bias, int cout, int kh, int kw, int stride, int pad, float *out) {
    int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return CPD_ERR_ARG;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < b; ++n)
        for (int co = 0; co < cout; ++co) {
            float *o = out + ((size_t)n * cout + co) * ho * wo;
            float bv = bias ? bias[co] : 0.f;
            for (int k = 0; k < ho * wo; ++k) o[k] = bv;
            for (int ci = 0; ci < cin; ++ci) {
                const float *ip = in + ((size_t)n * cin + ci) * h * w;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = 0; kx < kw; ++kx) {
                        float wv = wt[(((size_t)co * cin + ci) * kh + ky) * kw + kx];
                        for (int y = 0; y < ho; ++y) {
                            int iy = y * stride - pad + ky;
                            if (iy < 0 || iy >= h) continue;
                            /* x range with 0 <= x*stride - pad + kx < w */
                            int x0 = 0;
                            while (x0 < wo && x0 * stride - pad + kx < 0) ++x0;
                            int x1 = wo;
                            while (x1 > x0 && (x1 - 1) * stride - pad + kx >= w) --x1;
                            const float *irow = ip + (size_t)iy * w - pad + kx;
                            float *orow = o + (size_t)y * wo;
                            if (stride == 1)
                                for (int x = x0; x < x1; ++x) orow[x] += wv * irow[x];
                            else
                                for (int x = x0; x < x1; ++x) orow[x] += wv * irow[x * stride];
                        }
                    }
            }
        }
    return CPD_OK;
}

/* ConvTranspose2d with kernel == stride == k, no padding (base_bev_backbone.py:52-56):
 * in (B,Cin,H,W), w (Cin,Cout,k,k) -> out (B,Cout,H*k,W*k);
 * out[co][y*k+a][x*k+b] = sum_ci in[ci][y][x] * w[ci][co][a][b].                             */
int cpd_ref_deconv2d(const float *in, int b, int cin, int h, int w, const float *wt, int cout,
                     int k, float *out) {
    int ho = h * k, wo = w * k;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < b; ++n)
        for (int co = 0; co < cout; ++co) {
            float *o = out + ((size_t)n * cout + co) * ho * wo;
            for (int q = 0; q < ho * wo; ++q) o[q] = 0.f;
            for (int ci = 0; ci < cin; ++ci) {
                const float *ip = in + ((size_t)n * cin + ci) * h * w;
                for (int a = 0; a < k; ++a)
                    for (int bb = 0; bb < k; ++bb) {
                        float wv = wt[(((size_t)ci * cout + co) * k + a) * k + bb];
                        for (int y = 0; y < h; ++y)
                            for (int x = 0; x < w; ++x)
                                o[(size_t)(y * k + a) * wo + x * k + bb] += ip[(size_t)y * w + x] * wv;
                    }
            }
        }
    return CPD_OK;
}

/* Eval-mode BatchNorm2d (+ReLU) on NCHW: y = (x-mean)/sqrt(var+eps)*gamma+beta.              */
int cpd_ref_bn_relu_nchw(float *x, int b, int c, int hw, const float *gamma, const float *beta,
                         const float *mean, const float *var, float eps, int relu) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < b; ++n)
        for (int ch = 0; ch < c; ++ch) {
            float inv = 1.0f / sqrtf(var[ch] + eps);
            float *p = x + ((size_t)n * c + ch) * hw;
            for (int k = 0; k < hw; ++k) {
                float y = (p[k] - mean[ch]) * inv * gamma[ch] + beta[ch];
                if (relu && y < 0.f) y = 0.f;
                p[k] = y;
            }
        }
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * CenterHead decode: cpd/models/model_utils/centernet_utils.py:136-151 (_topk) and l.154-216
 * (decode_bbox_from_heatmap), driven by center_head.py:252-303. Single sample (B=1 slice).
 * Ties in top-K are broken by ascending flat index (torch leaves it unspecified; fixtures use
 * distinct scores).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float s; int32_t i; } sc_t;
static int sc_cmp(const void *a, const void *b) {
    const sc_t *x = (const sc_t *)a, *y = (const sc_t *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/* Descending top-k of v[n]: writes k scores and indices. */
int cpd_ref_topk(const float *v, int n, int k, float *out_s, int32_t *out_i) {
    if (k > n) return CPD_ERR_ARG;
    sc_t *a = (sc_t *)malloc((size_t)n * sizeof(sc_t));
    if (!a) return CPD_ERR_ALLOC;
    for (int i = 0; i < n; ++i) { a[i].s = v[i]; a[i].i = i; }
    qsort(a, n, sizeof(sc_t), sc_cmp);
    for (int i = 0; i < k; ++i) { out_s[i] = a[i].s; out_i[i] = a[i].i; }
    free(a);
    return CPD_OK;
}

/* hm (num_class,H,W) raw logits; center(2,H,W) center_z(1,H,W) dim(3,H,W) rot(2,H,W) raw heads
 * (rot[0] = cos, rot[1] = sin: center_head.py:266-267). Applies sigmoid to hm and exp to dim
 * (center_head.py:262,265), two-stage top-K, gather, atan2, scaling, range & score mask.
 * Outputs (capacity K): boxes [*,7], scores, labels (class id, 0-based), returns count in *n. */
int cpd_ref_center_decode(const float *hm, const float *center, const float *center_z,
                          const float *dim, const float *rot, int num_class, int h, int w, int K,
                          float stride, const float voxel_xy[2], const float range_lo_xy[2],
                          const float limit_range[6], float score_thresh, float *boxes,
                          float *scores, int32_t *labels, int32_t *n) {
    int hw = h * w;
    if (K > hw) return CPD_ERR_ARG;
    float *sig = (float *)malloc((size_t)num_class * hw * sizeof(float));
    float *s1 = (float *)malloc((size_t)num_class * K * sizeof(float));
    int32_t *i1 = (int32_t *)malloc((size_t)num_class * K * sizeof(int32_t));
    float *s2 = (float *)malloc((size_t)K * sizeof(float));
    int32_t *i2 = (int32_t *)malloc((size_t)K * sizeof(int32_t));
    if (!sig || !s1 || !i1 || !s2 || !i2) return CPD_ERR_ALLOC;
    for (size_t k = 0; k < (size_t)num_class * hw; ++k) sig[k] = 1.0f / (1.0f + expf(-hm[k]));
    for (int c = 0; c < num_class; ++c)
        cpd_ref_topk(sig + (size_t)c * hw, hw, K, s1 + (size_t)c * K, i1 + (size_t)c * K);
    cpd_ref_topk(s1, num_class * K, K, s2, i2);
    int cnt = 0;
    for (int k = 0; k < K; ++k) {
        int cls = i2[k] / K;
        int ind = i1[i2[k]];
        float ys = (float)(ind / w), xs = (float)(ind % w);
        float bx = (xs + center[ind]) * stride * voxel_xy[0] + range_lo_xy[0];
        float by = (ys + center[hw + ind]) * stride * voxel_xy[1] + range_lo_xy[1];
        float bz = center_z[ind];
        float d0 = expf(dim[ind]), d1 = expf(dim[hw + ind]), d2 = expf(dim[2 * hw + ind]);
        float ang = atan2f(rot[hw + ind], rot[ind]);
        float sc = s2[k];
        int keep = bx >= limit_range[0] && by >= limit_range[1] && bz >= limit_range[2] &&
                   bx <= limit_range[3] && by <= limit_range[4] && bz <= limit_range[5];
        if (keep && !(sc > score_thresh)) keep = 0;
        if (!keep) continue;
        float *o = boxes + 7 * (size_t)cnt;
        o[0] = bx; o[1] = by; o[2] = bz; o[3] = d0; o[4] = d1; o[5] = d2; o[6] = ang;
        scores[cnt] = sc;
        labels[cnt] = cls;
        ++cnt;
    }
    *n = cnt;
    free(sig); free(s1); free(i1); free(s2); free(i2);
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * B3: rotated BEV overlap / IoU / NMS. Restates cpd/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234
 * (== iou3d_cpu.cpp:59-229, same float structure) and the host greedy scan of
 * iou3d_nms.cpp:117-133. Boxes are [x,y,z,dx,dy,dz,heading] f32. Operation order is kept so the
 * degenerate cases (MARGIN corners, parallel edges) agree with the reference.
 * ------------------------------------------------------------------------------------------ */
#define IOU_EPS 1e-8f

static inline float cross3(const float *p1, const float *p2, const float *p0) {
    /* iou3d_nms_kernel.cu:39-41 */
    return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]);
}

static inline int rect_cross(const float *p1, const float *p2, const float *q1, const float *q2) {
    /* bounding-rectangle rejection, iou3d_nms_kernel.cu:43-49 */
    return fminf(p1[0], p2[0]) <= fmaxf(q1[0], q2[0]) && fminf(q1[0], q2[0]) <= fmaxf(p1[0], p2[0]) &&
           fminf(p1[1], p2[1]) <= fmaxf(q1[1], q2[1]) && fminf(q1[1], q2[1]) <= fmaxf(p1[1], p2[1]);
}

static inline int in_box2d(const float *box, const float *p) {
    /* iou3d_nms_kernel.cu:51-61; MARGIN 1e-2 */
    const float margin = 1e-2f;
    float ac = cosf(-box[6]), as = sinf(-box[6]);
    float rx = (p[0] - box[0]) * ac + (p[1] - box[1]) * (-as);
    float ry = (p[0] - box[0]) * as + (p[1] - box[1]) * ac;
    return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

static inline int seg_intersection(const float *p1, const float *p0, const float *q1,
                                   const float *q0, float *ans) {
    /* iou3d_nms_kernel.cu:63-92 */
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans[0] = (s5 * q0[0] - s1 * q1[0]) / (s5 - s1);
        ans[1] = (s5 * q0[1] - s1 * q1[1]) / (s5 - s1);
    } else {
        float a0 = p0[1] - p1[1], b0 = p1[0] - p0[0], c0 = p0[0] * p1[1] - p1[0] * p0[1];
        float a1 = q0[1] - q1[1], b1 = q1[0] - q0[0], c1 = q0[0] * q1[1] - q1[0] * q0[1];
        float D = a0 * b1 - a1 * b0;
        ans[0] = (b0 * c1 - b1 * c0) / D;
        ans[1] = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static inline void box_corners(const float *box, float c[5][2]) {
    /* iou3d_nms_kernel.cu:108-141: axis-aligned corners rotated about the centre */
    float hx = box[3] / 2, hy = box[4] / 2;
    float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    float ac = cosf(box[6]), as = sinf(box[6]);
    float raw[4][2] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; ++k) {
        float dx = raw[k][0] - box[0], dy = raw[k][1] - box[1];
        c[k][0] = dx * ac + dy * (-as) + box[0];
        c[k][1] = dx * as + dy * ac + box[1];
    }
    c[4][0] = c[0][0];
    c[4][1] = c[0][1];
}

float cpd_ref_box_overlap(const float *a, const float *b) {
    /* iou3d_nms_kernel.cu:104-225 */
    float ca[5][2], cb[5][2];
    box_corners(a, ca);
    box_corners(b, cb);
    float pts[16][2];
    float cx = 0.f, cy = 0.f;
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], pts[cnt])) {
                cx = cx + pts[cnt][0];
                cy = cy + pts[cnt][1];
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, cb[k])) {
            cx = cx + cb[k][0]; cy = cy + cb[k][1];
            pts[cnt][0] = cb[k][0]; pts[cnt][1] = cb[k][1];
            ++cnt;
        }
        if (in_box2d(b, ca[k])) {
            cx = cx + ca[k][0]; cy = cy + ca[k][1];
            pts[cnt][0] = ca[k][0]; pts[cnt][1] = ca[k][1];
            ++cnt;
        }
    }
    cx /= cnt; /* cnt == 0 gives nan/inf exactly as the reference; loops below are then empty */
    cy /= cnt;
    /* bubble sort by polar angle about the centroid (kernel.cu:100-102,200-211) */
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            float ai = atan2f(pts[i][1] - cy, pts[i][0] - cx);
            float an = atan2f(pts[i + 1][1] - cy, pts[i + 1][0] - cx);
            if (ai > an) {
                float tx = pts[i][0], ty = pts[i][1];
                pts[i][0] = pts[i + 1][0]; pts[i][1] = pts[i + 1][1];
                pts[i + 1][0] = tx; pts[i + 1][1] = ty;
            }
        }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        float ux = pts[k][0] - pts[0][0], uy = pts[k][1] - pts[0][1];
        float vx = pts[k + 1][0] - pts[0][0], vy = pts[k + 1][1] - pts[0][1];
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

float cpd_ref_iou_bev(const float *a, const float *b) {
    /* iou3d_nms_kernel.cu:227-234 */
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = cpd_ref_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, IOU_EPS);
}

static inline float iou_normal(const float *a, const float *b) {
    /* axis-aligned IoU ignoring heading, iou3d_nms_kernel.cu:314-326 */
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    float inter = width * height;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, IOU_EPS);
}

/* boxes_overlap_bev_gpu / boxes_iou_bev_gpu / boxes_iou_bev_cpu (iou3d_nms.cpp:49-88,
 * iou3d_cpu.cpp:232-252): full N x M matrices.                                                */
int cpd_ref_boxes_overlap_bev(const float *a, int n, const float *b, int m, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = cpd_ref_box_overlap(a + 7 * i, b + 7 * j);
    return CPD_OK;
}
int cpd_ref_boxes_iou_bev(const float *a, int n, const float *b, int m, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = cpd_ref_iou_bev(a + 7 * i, b + 7 * j);
    return CPD_OK;
}

/* 3D IoU composed as in cpd/ops/iou3d_nms/iou3d_nms_utils.py:67-100. */
int cpd_ref_boxes_iou3d(const float *a, int n, const float *b, int m, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            const float *p = a + 7 * i, *q = b + 7 * j;
            float amax = p[2] + p[5] / 2, amin = p[2] - p[5] / 2;
            float bmax = q[2] + q[5] / 2, bmin = q[2] - q[5] / 2;
            float oh = fminf(amax, bmax) - fmaxf(amin, bmin);
            if (oh < 0.f) oh = 0.f;
            float o3 = cpd_ref_box_overlap(p, q) * oh;
            float va = p[3] * p[4] * p[5], vb = q[3] * q[4] * q[5];
            out[(size_t)i * m + j] = o3 / fmaxf(va + vb - o3, 1e-6f);
        }
    return CPD_OK;
}

/* nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-186): boxes sorted by descending score on entry;
 * 64-wide bitmask rows (kernel.cu:267-311: bit i of word c set iff iou(row, 64c+i) > thr, only
 * columns after the row inside the diagonal block), then the serial greedy scan.
 * keep receives the kept row indices (int64), returns num_to_keep via *num_keep.             */
static int nms_impl(const float *boxes, int n, float thr, int64_t *keep, int32_t *num_keep,
                    int normal) {
    int cb = (n + 63) / 64;
    if (n == 0) { *num_keep = 0; return CPD_OK; }
    uint64_t *mask = (uint64_t *)calloc((size_t)n * cb, sizeof(uint64_t));
    uint64_t *remv = (uint64_t *)calloc((size_t)cb, sizeof(uint64_t));
    if (!mask || !remv) return CPD_ERR_ALLOC;
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < cb; ++c) {
            int csz = n - c * 64 < 64 ? n - c * 64 : 64;
            int start = (r / 64 == c) ? (r % 64) + 1 : 0;
            uint64_t t = 0;
            for (int i = start; i < csz; ++i) {
                const float *q = boxes + 7 * (size_t)(c * 64 + i);
                float v = normal ? iou_normal(boxes + 7 * (size_t)r, q)
                                 : cpd_ref_iou_bev(boxes + 7 * (size_t)r, q);
                if (v > thr) t |= 1ULL << i;
            }
            mask[(size_t)r * cb + c] = t;
        }
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[k++] = i;
            for (int j = nb; j < cb; ++j) remv[j] |= mask[(size_t)i * cb + j];
        }
    }
    *num_keep = k;
    free(mask);
    free(remv);
    return CPD_OK;
}
int cpd_ref_nms(const float *boxes, int n, float thr, int64_t *keep, int32_t *num_keep) {
    return nms_impl(boxes, n, thr, keep, num_keep, 0);
}
int cpd_ref_nms_normal(const float *boxes, int n, float thr, int64_t *keep, int32_t *num_keep) {
    return nms_impl(boxes, n, thr, keep, num_keep, 1);
}

/* ------------------------------------------------------------------------------------------
 * RoI-head feature pooling. The reference implements these as CUDA kernels, which
 * cannot be executed in a GPU-less container and cannot be compiled by gcc: PARITY OF THESE THREE
 * FUNCTIONS IS UNPINNED BY EXECUTION; they restate the kernels statement by statement, and the
 * Python layers around them (index shifting, empty-ball handling, MLPs, pooling) are pinned by
 * running the reference's own voxel_pool_modules.py / voxel_query_utils.py on top of these
 * functions (tests/golden/make_golden.py section 8).
 * ------------------------------------------------------------------------------------------ */

/* cpd/utils/spconv_utils.py:4-21 generate_voxel2pinds: dense (B,Z,Y,X) volume, -1 = empty,
 * otherwise the row of the voxel in the sparse tensor. */
int cpd_ref_voxel2pinds(const int32_t *indices, int n, int batch, const int32_t shape[3], int32_t *out) {
    if (!indices || !out || n < 0 || batch <= 0) return CPD_ERR_ARG;
    const long long cells = (long long)batch * shape[0] * shape[1] * shape[2];
    for (long long i = 0; i < cells; ++i) out[i] = -1;
    for (int i = 0; i < n; ++i) {
        const int32_t *q = indices + 4 * (size_t)i;
        out[(((long long)q[0] * shape[0] + q[1]) * shape[1] + q[2]) * shape[2] + q[3]] = i;
    }
    return CPD_OK;
}

/* cpd/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-87 voxel_query_kernel_stack, one
 * query point per iteration: scan the (2zr+1)(2yr+1)(2xr+1) neighbourhood of new_coords (b,z,y,x)
 * in dz, dy, dx order, keep the first nsample voxels whose centre is within radius; the first hit
 * pre-fills all slots; no hit -> idx[0] = -1 (the other slots keep the caller's initial zeros). */
int cpd_ref_voxel_query(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range,
                        int x_range, const float *new_xyz, const float *xyz, const int32_t *new_coords,
                        const int32_t *point_indices, int32_t *idx) {
    if (!new_xyz || !xyz || !new_coords || !point_indices || !idx || m < 0 || nsample <= 0) return CPD_ERR_ARG;
    const float radius2 = radius * radius;
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < m; ++pt) {
        const float *nx = new_xyz + 3 * (size_t)pt;
        const int32_t *nc = new_coords + 4 * (size_t)pt;
        int32_t *o = idx + (size_t)pt * nsample;
        const float new_x = nx[0], new_y = nx[1], new_z = nx[2];
        const int b = nc[0], cz = nc[1], cy = nc[2], cx = nc[3];
        int cnt = 0;
        for (int dz = -z_range; dz <= z_range; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= r1) continue;
            for (int dy = -y_range; dy <= y_range; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= r2) continue;
                for (int dx = -x_range; dx <= x_range; ++dx) {
                    const int x = cx + dx;
                    if (x < 0 || x >= r3) continue;
                    const long long cell = (((long long)b * r1 + z) * r2 + y) * r3 + x;
                    const int32_t nb = point_indices[cell];
                    if (nb < 0) continue;
                    const float xp = xyz[3 * (size_t)nb], yp = xyz[3 * (size_t)nb + 1], zp = xyz[3 * (size_t)nb + 2];
                    const float d2 = (xp - new_x) * (xp - new_x) + (yp - new_y) * (yp - new_y) + (zp - new_z) * (zp - new_z);
                    if (d2 > radius2) continue;
                    if (cnt < nsample) {
                        if (cnt == 0)
                            for (int l = 0; l < nsample; ++l) o[l] = nb;
                        o[cnt] = nb;
                        ++cnt;
                    }
                }
            }
        }
        if (cnt == 0) o[0] = -1;
    }
    return CPD_OK;
}

/* cpd/ops/pointnet2/pointnet2_stack/src/group_points_gpu.cu:69-99 group_points_kernel_stack:
 * out[pt][c][s] = features[start(batch of pt) + idx[pt][s]][c]. */
int cpd_ref_group_points(int b, int m, int c, int nsample, const float *features, const int32_t *features_batch_cnt,
                         const int32_t *idx, const int32_t *idx_batch_cnt, float *out) {
    if (!features || !features_batch_cnt || !idx || !idx_batch_cnt || !out || b <= 0) return CPD_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < m; ++pt) {
        int bs = 0, cnt = idx_batch_cnt[0];
        for (int k = 1; k < b; ++k) {
            if (pt < cnt) break;
            cnt += idx_batch_cnt[k];
            bs = k;
        }
        long long start = 0;
        for (int k = 0; k < bs; ++k) start += features_batch_cnt[k];
        for (int ci = 0; ci < c; ++ci)
            for (int s = 0; s < nsample; ++s)
                out[((size_t)pt * c + ci) * nsample + s] = features[(start + idx[(size_t)pt * nsample + s]) * c + ci];
    }
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Dataloader pre-filter.
 * ------------------------------------------------------------------------------------------ */

/* common_utils.mask_points_by_range (cpd/utils/common_utils.py:60-63) followed by the boolean
 * indexing of DataProcessor.mask_points_and_boxes_outside_range (data_processor.py:84-85): keeps,
 * in order, the points with range[0] <= x <= range[3] and range[1] <= y <= range[4]. */
int cpd_ref_mask_points_by_range(const float *points, int n, int c, const float range[6], float *out, int32_t *n_out) {
    if (!points || !out || !n_out || n < 0 || c < 2) return CPD_ERR_ARG;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float x = points[(size_t)i * c], y = points[(size_t)i * c + 1];
        if (x >= range[0] && x <= range[3] && y >= range[1] && y <= range[4]) {
            memcpy(out + (size_t)m * c, points + (size_t)i * c, (size_t)c * sizeof(float));
            ++m;
        }
    }
    *n_out = m;
    return CPD_OK;
}

/* roiaware_pool3d_kernel.cu:23-35 check_pt_in_box3d (MARGIN 1e-5) / roiaware_pool3d.cpp:121-140
 * check_pt_in_box3d_cpu (MARGIN 1e-2): point inside the z-extent and inside the rotated xy
 * rectangle grown by MARGIN. */
static int ref_pt_in_box(const float *pt, const float *b, float margin) {
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = b[0], cy = b[1], cz = b[2], dx = b[3], dy = b[4], dz = b[5], rz = b[6];
    if (fabsf(z - cz) > dz / 2.0) return 0;
    const float cosa = cosf(-rz), sina = sinf(-rz);
    const float sx = x - cx, sy = y - cy;
    const float lx = sx * cosa + sy * (-sina);
    const float ly = sx * sina + sy * cosa;
    return (fabs(lx) < dx / 2.0 + margin) & (fabs(ly) < dy / 2.0 + margin);
}

/* roiaware_pool3d_kernel.cu:313-336 points_in_boxes_kernel: index of the FIRST box (in box order)
 * that contains the point, -1 if none. boxes [B, N, 7], pts [B, M, 3], out [B, M]. */
int cpd_ref_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts,
                            float margin, int32_t *box_idx_of_points) {
    if (batch < 0 || boxes_num < 0 || pts_num < 0 || !box_idx_of_points) return CPD_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)batch * pts_num; ++i) {
        const int b = (int)(i / pts_num);
        const float *pt = pts + 3 * (size_t)i;
        int32_t hit = -1;
        for (int k = 0; k < boxes_num; ++k)
            if (ref_pt_in_box(pt, boxes + ((size_t)b * boxes_num + k) * 7, margin)) { hit = k; break; }
        box_idx_of_points[i] = hit;
    }
    return CPD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Anchor head: nearest-BEV IoU, ResidualCoder, target assignment, box decoding.
 * All of it is plain torch in the reference (importable by file path), so every function here is
 * pinned on goldens produced by the reference code itself (tests/golden/make_golden.py section 9).
 * Arithmetic is kept op for op in fp32 (no contraction) because the assigner compares IoUs for
 * exact equality (axis_aligned_target_assigner.py:178).
 * ------------------------------------------------------------------------------------------ */
static const float CPD_PI_F = 3.14159265358979323846f;

/* common_utils.limit_period (cpd/utils/common_utils.py:17-20) in fp32 tensor arithmetic */
static float ref_limit_period(float val, float offset, float period) {
    return val - floorf(val / period + offset) * period;
}

/* box_utils.boxes3d_lidar_to_aligned_bev_boxes (cpd/utils/box_utils.py:261-272) */
static void ref_aligned_bev(const float *b, float out[4]) {
    const float rot = fabsf(ref_limit_period(b[6], 0.5f, CPD_PI_F));
    const float quarter = (float)(3.14159265358979323846 / 4);
    const float d0 = rot < quarter ? b[3] : b[4], d1 = rot < quarter ? b[4] : b[3];
    out[0] = b[0] - d0 / 2; out[1] = b[1] - d1 / 2; out[2] = b[0] + d0 / 2; out[3] = b[1] + d1 / 2;
}

/* box_utils.boxes_iou_normal (l.238-258) on two aligned boxes */
static float ref_iou_normal(const float a[4], const float b[4]) {
    const float x_min = fmaxf(a[0], b[0]), x_max = fminf(a[2], b[2]);
    const float y_min = fmaxf(a[1], b[1]), y_max = fminf(a[3], b[3]);
    const float x_len = fmaxf(x_max - x_min, 0.f), y_len = fmaxf(y_max - y_min, 0.f);
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    const float inter = x_len * y_len;
    return inter / fmaxf(area_a + area_b - inter, 1e-6f);
}

/* box_utils.boxes3d_nearest_bev_iou (l.275-287): a [n,7], b [m,7] -> [n,m] */
int cpd_ref_nearest_bev_iou(const float *a, int n, const float *b, int m, float *out) {
    if (n < 0 || m < 0 || (n * m > 0 && (!a || !b || !out))) return CPD_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float ba[4];
        ref_aligned_bev(a + 7 * (size_t)i, ba);
        for (int j = 0; j < m; ++j) {
            float bb[4];
            ref_aligned_bev(b + 7 * (size_t)j, bb);
            out[(size_t)i * m + j] = ref_iou_normal(ba, bb);
        }
    }
    return CPD_OK;
}

/* ResidualCoder.encode_torch / decode_torch (cpd/utils/box_coder_utils.py:13-45, 47-81), code size 7 */
int cpd_ref_residual_encode(const float *boxes, const float *anchors, int n, float *out) {
    if (n < 0 || (n > 0 && (!boxes || !anchors || !out))) return CPD_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const float *g = boxes + 7 * (size_t)i, *a = anchors + 7 * (size_t)i;
        float *o = out + 7 * (size_t)i;
        const float dxa = fmaxf(a[3], 1e-5f), dya = fmaxf(a[4], 1e-5f), dza = fmaxf(a[5], 1e-5f);
        const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
        const float diag = sqrtf(dxa * dxa + dya * dya);
        o[0] = (g[0] - a[0]) / diag; o[1] = (g[1] - a[1]) / diag; o[2] = (g[2] - a[2]) / dza;
        o[3] = logf(dxg / dxa); o[4] = logf(dyg / dya); o[5] = logf(dzg / dza);
        o[6] = g[6] - a[6];
    }
    return CPD_OK;
}

/* AnchorHeadTemplate.generate_predicted_boxes (anchor_head_template.py:336-383): decode_torch against
 * the anchors, then the direction-classifier correction (l.365-376). box_preds [b,n,7], anchors [n,7],
 * dir_cls [b,n,nbins] or NULL. */
int cpd_ref_anchor_decode(const float *box_preds, const float *anchors, const float *dir_cls, int batch, int n, int nbins,
                          float dir_offset, float dir_limit_offset, float *out) {
    if (batch < 0 || n < 0 || (batch * n > 0 && (!box_preds || !anchors || !out)) || (dir_cls && nbins <= 0)) return CPD_ERR_ARG;
    const float period = (float)(2 * 3.14159265358979323846 / (nbins > 0 ? nbins : 1));
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)batch * n; ++i) {
        const float *t = box_preds + 7 * (size_t)i, *a = anchors + 7 * (size_t)(i % n);
        float *o = out + 7 * (size_t)i;
        const float diag = sqrtf(a[3] * a[3] + a[4] * a[4]);
        o[0] = t[0] * diag + a[0]; o[1] = t[1] * diag + a[1]; o[2] = t[2] * a[5] + a[2];
        o[3] = expf(t[3]) * a[3]; o[4] = expf(t[4]) * a[4]; o[5] = expf(t[5]) * a[5];
        float rg = t[6] + a[6];
        if (dir_cls) {
            const float *d = dir_cls + (size_t)i * nbins;
            int lab = 0;
            for (int k = 1; k < nbins; ++k)
                if (d[k] > d[lab]) lab = k;                       /* torch.max: first maximum */
            const float dir_rot = ref_limit_period(rg - dir_offset, dir_limit_offset, period);
            rg = dir_rot + dir_offset + period * (float)lab;
        }
        o[6] = rg;
    }
    return CPD_OK;
}

/* AxisAlignedTargetAssigner.assign_targets_single (axis_aligned_target_assigner.py:153-243) with
 * match_height = False (nearest-BEV IoU) and POS_FRACTION < 0 (no sampling, the shipped configs).
 * labels [n] i32 (-1 ignore, 0 background, class id), bbox_targets [n,7], reg_weights [n], gt_ious [n]. */
int cpd_ref_anchor_assign(const float *anchors, int n, const float *gt, int m, const int32_t *gt_classes,
                          float matched_thr, float unmatched_thr, int norm_by_num_examples, int32_t *labels,
                          float *bbox_targets, float *reg_weights, float *gt_ious) {
    if (n < 0 || m < 0 || !labels || !bbox_targets || !reg_weights || !gt_ious || (n > 0 && !anchors) ||
        (m > 0 && (!gt || !gt_classes)))
        return CPD_ERR_ARG;
    for (int i = 0; i < n; ++i) { labels[i] = -1; gt_ious[i] = 0.f; reg_weights[i] = 0.f; }
    memset(bbox_targets, 0, (size_t)n * 7 * sizeof(float));
    if (m == 0 || n == 0) {
        for (int i = 0; i < n; ++i) labels[i] = 0;
        return CPD_OK;
    }
    float *ov = (float *)malloc((size_t)n * m * sizeof(float));
    int32_t *amax = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    float *gmax = (float *)malloc((size_t)m * sizeof(float));
    char *forced = (char *)calloc((size_t)n, 1);
    if (!ov || !amax || !gmax || !forced) { free(ov); free(amax); free(gmax); free(forced); return CPD_ERR_ALLOC; }
    cpd_ref_nearest_bev_iou(anchors, n, gt, m, ov);
    for (int i = 0; i < n; ++i) {                                   /* argmax(axis=1): first maximum */
        int best = 0;
        for (int j = 1; j < m; ++j)
            if (ov[(size_t)i * m + j] > ov[(size_t)i * m + best]) best = j;
        amax[i] = best;
        gt_ious[i] = ov[(size_t)i * m + best];
    }
    for (int j = 0; j < m; ++j) {
        float mx = ov[j];
        for (int i = 1; i < n; ++i)
            if (ov[(size_t)i * m + j] > mx) mx = ov[(size_t)i * m + j];
        gmax[j] = mx == 0.f ? -1.f : mx;                            /* empty_gt_mask, l.175-176 */
    }
    for (int i = 0; i < n; ++i)                                     /* anchors_with_max_overlap, l.178 */
        for (int j = 0; j < m; ++j)
            if (ov[(size_t)i * m + j] == gmax[j]) { forced[i] = 1; break; }
    for (int i = 0; i < n; ++i) {
        if (forced[i]) labels[i] = gt_classes[amax[i]];
        if (gt_ious[i] >= matched_thr) labels[i] = gt_classes[amax[i]];
    }
    for (int i = 0; i < n; ++i)                                     /* POS_FRACTION None branch, l.214-216 */
        if (gt_ious[i] < unmatched_thr) labels[i] = 0;
    for (int i = 0; i < n; ++i)
        if (forced[i]) labels[i] = gt_classes[amax[i]];
    int n_examples = 0;
    for (int i = 0; i < n; ++i) n_examples += labels[i] >= 0;
    for (int i = 0; i < n; ++i) {
        if (labels[i] <= 0) continue;
        cpd_ref_residual_encode(gt + 7 * (size_t)amax[i], anchors + 7 * (size_t)i, 1, bbox_targets + 7 * (size_t)i);
        reg_weights[i] = norm_by_num_examples ? 1.0f / (float)(n_examples > 1 ? n_examples : 1) : 1.0f;
    }
    free(ov); free(amax); free(gmax); free(forced);
    return CPD_OK;
}
