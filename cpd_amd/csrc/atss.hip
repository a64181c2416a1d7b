// atss.hip -- ATSS target assignment for gfx950 without the (anchors x GT) matrices.
//
// Replaces ATSSTargetAssigner.assign_targets_single (cpd/models/dense_heads/target_assigner/atss_target_assigner.py:76-141),
// which builds an N x M IoU matrix (iou3d_nms_utils.boxes_iou_bev / boxes_iou3d_gpu over ALL anchors), an N x M distance
// matrix, a top-k over N per GT and a second N x M "ious_inf" matrix. For N = 212k anchors and M = 60 boxes that is
// 3 x 51 MB written and re-read per frame and class to find at most k*M + M positive anchors. Here:
//   atss_topk_kernel    one workgroup per GT: the k anchors nearest to the box centre (k rounds of a block-wide argmin over
//                       (distance, index) pairs -- ties go to the lower anchor index, the reference's topk leaves them
//                       unspecified), their IoUs with the GT, mean + unbiased std (Welford, like torch.std), the
//                       threshold test and the centre-inside-the-box test (l.94-112): <= k candidates per GT leave the chip
//   atss_colmax_kernel  ious.max(dim=0) (l.126): every (anchor, GT) pair whose circumscribed circles touch gets its rotated
//                       IoU evaluated in registers; one 64-bit atomicMax per overlapping pair keeps (IoU, lowest index)
//   atss_finalize_kernel one workgroup: per anchor the candidate GT of highest IoU (l.118-124, first GT on ties like
//                       torch.max on the CPU), then the forced matches in GT order (l.127-128, the last GT wins a shared
//                       anchor), labels / ResidualCoder.encode_torch targets / weights scattered into the zeroed outputs
// Compiled with -ffp-contract=off (Makefile): distances, thresholds and the encoding follow torch's fp32 operation order.
#include <math.h>

#include "box_geom.h"
#include "common.h"

namespace {

constexpr int ATSS_MAX_K = 64;
constexpr int ATSS_MAX_CAND = 4096;       // k * M candidates handled by the one-workgroup finalize pass

__device__ __forceinline__ float pair_iou(const BoxG &A, const BoxG &B, bool match_height) {
    if (!match_height) return iou_bev_g(A, B);
    // boxes_iou3d_gpu, iou3d_nms_utils.py:76-98
    const float amax = A.b[2] + A.b[5] / 2, amin = A.b[2] - A.b[5] / 2;
    const float bmax = B.b[2] + B.b[5] / 2, bmin = B.b[2] - B.b[5] / 2;
    float oh = fminf(amax, bmax) - fmaxf(amin, bmin);
    oh = oh < 0.f ? 0.f : oh;
    const float o3 = box_overlap_g(A, B) * oh;
    const float va = A.b[3] * A.b[4] * A.b[5], vb = B.b[3] * B.b[4] * B.b[5];
    return o3 / fmaxf(va + vb - o3, 1e-6f);
}

struct DistKey {
    float d;
    int i;
};
__device__ __forceinline__ bool key_less(DistKey a, DistKey b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }

// cand layout (per GT g, slot kk): cand_idx[kk * m + g] = anchor index, cand_iou = its IoU, cand_pos = is_pos & is_in_gt
__global__ void __launch_bounds__(256) atss_topk_kernel(const float *__restrict__ anchors, int n, const float *__restrict__ gt, int gt_ld,
                                                        int m, int k, int match_height, int32_t *__restrict__ cand_idx,
                                                        float *__restrict__ cand_iou, int32_t *__restrict__ cand_pos,
                                                        unsigned long long *__restrict__ colmax) {
    __shared__ DistKey s_red[4];
    __shared__ DistKey s_prev;
    __shared__ float s_iou[ATSS_MAX_K];
    __shared__ int s_idx[ATSS_MAX_K];
    __shared__ float s_thresh;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *gb = gt + (size_t)g * gt_ld;
    const float gx = gb[0], gy = gb[1], gz = gb[2];
    if (tid == 0) {
        s_prev = DistKey{-1.f, -1};
        colmax[g] = 0xffffffffull;          // IoU 0 at anchor 0 (see atss_colmax_kernel, which runs after this kernel)
    }
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        const DistKey prev = s_prev;
        DistKey best{INFINITY, 0x7fffffff};
        for (int i = tid; i < n; i += 256) {
            const float *a = anchors + 7 * (size_t)i;
            const float dx = a[0] - gx, dy = a[1] - gy, dz = a[2] - gz;
            const DistKey key{sqrtf(dx * dx + dy * dy + dz * dz), i};      // (anchors - gt).norm(dim=-1), l.93
            if (key_less(prev, key) && key_less(key, best)) best = key;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            DistKey o{__shfl_xor(best.d, off), __shfl_xor(best.i, off)};
            if (key_less(o, best)) best = o;
        }
        if (lane == 0) s_red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            DistKey b = s_red[0];
            for (int w = 1; w < 4; ++w)
                if (key_less(s_red[w], b)) b = s_red[w];
            s_prev = b;
            s_idx[round] = b.i;
        }
        __syncthreads();
    }
    BoxG G;
    box_setup(gb, G);
    if (tid < k) {
        BoxG A;
        box_setup(anchors + 7 * (size_t)s_idx[tid], A);
        s_iou[tid] = pair_iou(A, G, match_height != 0);
    }
    __syncthreads();
    if (tid == 0) {
        // torch.mean = sum / k; torch.std = sqrt(M2 / (k - 1)) with Welford's update (k = 1 -> nan -> nothing is positive)
        float sum = 0.f, mean = 0.f, m2 = 0.f;
        for (int j = 0; j < k; ++j) {
            const float x = s_iou[j];
            sum += x;
            const float delta = x - mean;
            mean += delta / (float)(j + 1);
            m2 += delta * (x - mean);
        }
        const float stdv = sqrtf(m2 / (float)(k - 1));
        s_thresh = sum / (float)k + stdv + 1e-6f;                           // l.98
    }
    __syncthreads();
    if (tid < k) {
        const float *a = anchors + 7 * (size_t)s_idx[tid];
        // centre of the anchor in the GT's frame (l.103-111): rotate_points_along_z(xyz_local, -heading) is the row vector
        // (x, y, z) times [[c, s, 0], [-s, c, 0], [0, 0, 1]] with c = cos(-heading), s = sin(-heading)
        const float lx = a[0] - gx, ly = a[1] - gy, lz = a[2] - gz;
        const float ang = -gb[6];
        const float c = cosf(ang), s = sinf(ang);
        const float xr = (lx * c + ly * (-s)) + lz * 0.f;
        const float yr = (lx * s + ly * c) + lz * 0.f;
        // "lw" = gt[3:5][:, [1, 0]]: x is tested against dy / 2, y against dx / 2 (l.108-109, the reference's own comment)
        const float hx = gb[4] / 2, hy = gb[3] / 2;
        const bool inside = xr <= hx && xr >= -hx && yr <= hy && yr >= -hy;
        const bool pos = s_iou[tid] >= s_thresh;
        cand_idx[tid * m + g] = s_idx[tid];
        cand_iou[tid * m + g] = s_iou[tid];
        cand_pos[tid * m + g] = (pos && inside) ? 1 : 0;
    }
}

// colmax[g] = (IoU bits << 32) | (0xffffffff - anchor index): max = highest IoU, then lowest index; initialised to IoU 0 at
// anchor 0, which is what ious.max(dim=0) returns for a GT nothing overlaps
__global__ void __launch_bounds__(256) atss_colmax_kernel(const float *__restrict__ anchors, int n, const float *__restrict__ gt, int gt_ld,
                                                          int m, int match_height, unsigned long long *__restrict__ colmax) {
    extern __shared__ float s_gt[];           // m x 8: box + circumradius
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const float *gb = gt + (size_t)j * gt_ld;
        for (int q = 0; q < 7; ++q) s_gt[8 * j + q] = gb[q];
        s_gt[8 * j + 7] = 0.5f * sqrtf(gb[3] * gb[3] + gb[4] * gb[4]);
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = anchors + 7 * (size_t)i;
    const float ax = a[0], ay = a[1];
    const float ar = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]);
    BoxG A;
    bool have = false;
    for (int j = 0; j < m; ++j) {
        const float dx = s_gt[8 * j] - ax, dy = s_gt[8 * j + 1] - ay;
        const float rr = (ar + s_gt[8 * j + 7]) * 1.0001f + 1e-4f;        // circles apart => the rectangles cannot intersect
        if (dx * dx + dy * dy > rr * rr) continue;
        if (!have) { box_setup(a, A); have = true; }
        BoxG G;
        box_setup(s_gt + 8 * j, G);
        const float v = pair_iou(A, G, match_height != 0);
        if (v > 0.f)
            atomicMax(&colmax[j], ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i));
    }
}

__device__ __forceinline__ void encode_target(const float *g, const float *an, float *o) {
    // ResidualCoder.encode_torch, box_coder_utils.py:13-44 (code_size 7)
    const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
    const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
    const float diag = sqrtf(dxa * dxa + dya * dya);
    o[0] = (g[0] - an[0]) / diag;
    o[1] = (g[1] - an[1]) / diag;
    o[2] = (g[2] - an[2]) / dza;
    o[3] = logf(dxg / dxa);
    o[4] = logf(dyg / dya);
    o[5] = logf(dzg / dza);
    o[6] = g[6] - an[6];
}

__global__ void __launch_bounds__(256) atss_finalize_kernel(const float *__restrict__ anchors, const float *__restrict__ gt, int gt_ld,
                                                            int m, int k, const int32_t *__restrict__ cand_idx,
                                                            const float *__restrict__ cand_iou, const int32_t *__restrict__ cand_pos,
                                                            const unsigned long long *__restrict__ colmax,
                                                            float *__restrict__ labels, float *__restrict__ targets,
                                                            float *__restrict__ weights) {
    const int nc = k * m;
    // a positive candidate (kk, g) owns its anchor unless another positive candidate of the same anchor has a higher IoU, or the
    // same IoU and a lower GT index (ious_inf.max(dim=1), l.124)
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        if (!cand_pos[c]) continue;
        const int g = c % m, a = cand_idx[c];
        const float v = cand_iou[c];
        bool wins = true;
        for (int o = 0; o < nc && wins; ++o) {
            if (o == c || !cand_pos[o] || cand_idx[o] != a) continue;
            const int og = o % m;
            const float ov = cand_iou[o];
            if (ov > v || (ov == v && og < g)) wins = false;
            if (ov == v && og == g && o < c) wins = false;          // the same (anchor, GT) pair cannot occur twice; guard anyway
        }
        if (!wins) continue;
        const float *gb = gt + (size_t)g * gt_ld;
        const float cls = gb[gt_ld - 1];
        labels[a] = cls;
        if (cls > 0.f) {
            encode_target(gb, anchors + 7 * (size_t)a, targets + 7 * (size_t)a);
            weights[a] = 1.f;
        }
    }
    __syncthreads();
    // anchors_to_gt_indexs[argmax_iou_of_each_gt] = arange(num_gt) (l.127-128): sequential, a later GT overrides an earlier one
    if (threadIdx.x == 0) {
        for (int g = 0; g < m; ++g) {
            const int a = (int)(0xffffffffu - (unsigned)(colmax[g] & 0xffffffffull));
            const float *gb = gt + (size_t)g * gt_ld;
            const float cls = gb[gt_ld - 1];
            labels[a] = cls;
            float *o = targets + 7 * (size_t)a;
            if (cls > 0.f) {
                encode_target(gb, anchors + 7 * (size_t)a, o);
                weights[a] = 1.f;
            } else {
                for (int q = 0; q < 7; ++q) o[q] = 0.f;
                weights[a] = 0.f;
            }
        }
    }
}

}  // namespace

extern "C" size_t cpd_atss_workspace_bytes(int m, int topk) {
    if (m <= 0 || topk <= 0) return 256;
    return cpd_align((size_t)m * 8) + 3 * cpd_align((size_t)m * topk * 4);
}

extern "C" int cpd_atss_assign(const float *anchors, int n_anchors, const float *gt_boxes, int gt_ld, int m, int topk,
                               int match_height, float *labels, float *reg_targets, float *reg_weights, void *workspace,
                               size_t workspace_bytes, cpd_stream_t stream) {
    if (!anchors || !gt_boxes || !labels || !reg_targets || !reg_weights || !workspace || n_anchors <= 0 || m <= 0 || gt_ld < 8 ||
        topk <= 0)
        return CPD_ERR_ARG;
    if (topk > ATSS_MAX_K || topk > n_anchors || (long long)topk * m > ATSS_MAX_CAND) return CPD_ERR_UNSUPPORTED;
    if (workspace_bytes < cpd_atss_workspace_bytes(m, topk)) return CPD_ERR_WORKSPACE;
    hipStream_t s = cpd_s(stream);
    char *w = static_cast<char *>(workspace);
    unsigned long long *colmax = reinterpret_cast<unsigned long long *>(w);
    w += cpd_align((size_t)m * 8);
    int32_t *cand_idx = reinterpret_cast<int32_t *>(w);
    w += cpd_align((size_t)m * topk * 4);
    float *cand_iou = reinterpret_cast<float *>(w);
    w += cpd_align((size_t)m * topk * 4);
    int32_t *cand_pos = reinterpret_cast<int32_t *>(w);
    CPD_HIP_TRY(hipMemsetAsync(labels, 0, (size_t)n_anchors * 4, s));
    CPD_HIP_TRY(hipMemsetAsync(reg_targets, 0, (size_t)n_anchors * 7 * 4, s));
    CPD_HIP_TRY(hipMemsetAsync(reg_weights, 0, (size_t)n_anchors * 4, s));
    atss_topk_kernel<<<m, 256, 0, s>>>(anchors, n_anchors, gt_boxes, gt_ld, m, topk, match_height, cand_idx, cand_iou, cand_pos, colmax);
    atss_colmax_kernel<<<cpd_div_up(n_anchors, 256), 256, (size_t)m * 8 * sizeof(float), s>>>(anchors, n_anchors, gt_boxes, gt_ld, m,
                                                                                              match_height, colmax);
    atss_finalize_kernel<<<1, 256, 0, s>>>(anchors, gt_boxes, gt_ld, m, topk, cand_idx, cand_iou, cand_pos, colmax, labels,
                                           reg_targets, reg_weights);
    return cpd_check_launch();
}
