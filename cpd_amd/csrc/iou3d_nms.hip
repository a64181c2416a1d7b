// iou3d_nms.hip -- rotated BEV overlap / IoU / 3D IoU / NMS for gfx950.
//
// Replaces the iou3d_nms_cuda extension: cpd/ops/iou3d_nms/src/iou3d_nms_kernel.cu (geometry
// l.35-234, pairwise kernels l.236-265, bitmask NMS l.267-372) and the host side of
// cpd/ops/iou3d_nms/src/iou3d_nms.cpp:90-186 (mask D2H + serial CPU greedy scan).
//
// Geometry: the reference's polygon-clipping arithmetic is restated with the same fp32 operation
// order (edge x edge crossings i=0..3 x j=0..3, then corners interleaved b-in-a / a-in-b, centroid,
// bubble sort by atan2, shoelace fan), so degenerate cases (MARGIN corners, parallel edges) agree.
// The functions are __host__ __device__: the one CPU entry point of the reference extension
// (boxes_iou_bev_cpu, iou3d_cpu.cpp:232-252) is served by the same code.
//
// NMS: the reference's 64-thread block IS a CDNA wavefront and its mask word IS a wave ballot. Here
// lane = column box, each wave walks its 64 rows, and `__ballot(iou > thr)` yields the mask word
// directly (all 64 lanes busy on every pair instead of one thread looping over 64 columns). The
// greedy scan runs on the device in one wave (removed-set kept as one 64-bit word per lane), so
// the reference's blocking D2H copy of the N x N/64 mask disappears.
#include <math.h>

#include "box_geom.h"
#include "common.h"

namespace {

enum { MODE_OVERLAP = 0, MODE_IOU_BEV = 1, MODE_IOU3D = 2 };

// Pairwise matrices: lane = column box (b side, kept in registers), each wave walks 16 rows.
template <int MODE>
__global__ void __launch_bounds__(256) pairwise_kernel(const float *__restrict__ a, int n, const float *__restrict__ b,
                                                       int m, float *__restrict__ out) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.y * 4 + wave) * 16;
    if (row0 >= n) return;
    BoxG B;
    const bool colok = col < m;
    box_setup(b + 7 * (size_t)(colok ? col : 0), B);
    const int row1 = row0 + 16 < n ? row0 + 16 : n;
    for (int row = row0; row < row1; ++row) {
        BoxG A;
        box_setup(a + 7 * (size_t)row, A);  // wave-uniform address: scalar/broadcast loads
        float v;
        if (MODE == MODE_OVERLAP) {
            v = box_overlap_g(A, B);
        } else if (MODE == MODE_IOU_BEV) {
            v = iou_bev_g(A, B);
        } else {
            // boxes_iou3d_gpu, iou3d_nms_utils.py:76-98
            const float amax = A.b[2] + A.b[5] / 2, amin = A.b[2] - A.b[5] / 2;
            const float bmax = B.b[2] + B.b[5] / 2, bmin = B.b[2] - B.b[5] / 2;
            float oh = fminf(amax, bmax) - fmaxf(amin, bmin);
            oh = oh < 0.f ? 0.f : oh;
            const float o3 = box_overlap_g(A, B) * oh;
            const float va = A.b[3] * A.b[4] * A.b[5], vb = B.b[3] * B.b[4] * B.b[5];
            v = o3 / fmaxf(va + vb - o3, 1e-6f);
        }
        if (colok) out[(size_t)row * m + col] = v;
    }
}

// mask[row][cb] bit i = iou(row, 64*cb + i) > thr, for columns after the row inside the diagonal
// block and all columns of later blocks (iou3d_nms_kernel.cu:267-311). Blocks below the diagonal
// are never read by the scan (it starts at j = nblock) and are not computed. One WAVE per
// (row, column block): lane = column box, the ballot is the mask word -- N * N/64 / 2 independent
// waves instead of the reference's 64 serial IoUs per thread. Batched over samples (blockIdx.z);
// per-sample box counts may live on the device (no host round trip between decode and NMS).
// (round 5) `box_cap` = boxes per sample in `boxes_all`, `cap` = rows / columns of the sample's MASK: the first-survivors entry point
// (cpd_nms_batch_first) builds the mask of a sample's first `cap` < box_cap boxes only.
template <bool NORMAL>
__global__ void __launch_bounds__(256) nms_mask_kernel(const float *__restrict__ boxes_all, const int32_t *__restrict__ counts,
                                                       int n_host, int cap, int box_cap, float thr,
                                                       unsigned long long *__restrict__ mask_all, const int32_t *__restrict__ only_if) {
    const int smp = blockIdx.z;
    if (only_if && !only_if[smp]) return;          // (cpd_nms_batch_where: samples whose flag is 0 keep what they have)
    const int n = counts ? min(counts[smp], cap) : n_host;
    const int cb = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.y * 4 + wave;
    if (row >= n || cb * 64 >= n || cb < (row >> 6)) return;
    const float *boxes = boxes_all + (size_t)smp * box_cap * 7;
    const int ncb_cap = (cap + 63) >> 6;
    unsigned long long *mask = mask_all + (size_t)smp * cap * ncb_cap;
    const int col = cb * 64 + lane;
    bool hit = false;
    if (col < n && col > row) {
        if (NORMAL) {
            hit = iou_normal_g(boxes + 7 * (size_t)row, boxes + 7 * (size_t)col) > thr;
        } else {
            BoxG A, B;
            box_setup(boxes + 7 * (size_t)row, A);
            box_setup(boxes + 7 * (size_t)col, B);
            hit = iou_bev_g(A, B) > thr;
        }
    }
    const unsigned long long word = __ballot(hit);
    if (lane == 0) mask[(size_t)row * ncb_cap + cb] = word;
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
    return v;
}

// Greedy scan (iou3d_nms.cpp:117-133) by ONE wave per sample. Lane w owns word w (+64, ...) of the
// removed set. Per 64-box block: the 64 diagonal words sit one per lane and the in-block greedy
// pass is register-only (shuffles); then lane i (= kept row i of the block) reads its later words
// and a butterfly OR folds them into the removed set. With LDS=true the sample's whole mask is
// first copied into LDS with coalesced, pipelined loads (fits up to 1024 boxes).
// (round 5) `keep_cap` = entries per sample in `keep_all` (the boxes' capacity), `cap` = the mask's; `max_keep`: the scan stops after
// the 64-box block in which the max_keep-th survivor was found (the survivors up to there are exactly the full scan's first ones);
// `incomplete` (may be NULL): set to 1 when the sample has more boxes than the mask covers and fewer than max_keep survived among those
// it covers -- the caller then needs the full scan.
// (round 5) launched with 256 threads when LDS: all four waves copy the mask (16-byte pieces), wave 0 scans -- one wave fetching 32 KB in
// 8-byte pieces was most of the kernel's 55 us at 500 boxes.
template <bool LDS>
__global__ void __launch_bounds__(LDS ? 256 : 64) nms_scan_kernel(const unsigned long long *__restrict__ mask_all,
                                                      const int32_t *__restrict__ counts, int n_host, int cap, int keep_cap, int max_keep,
                                                      long long *__restrict__ keep_all, int *__restrict__ num_keep,
                                                      int *__restrict__ incomplete, const int32_t *__restrict__ only_if) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_mask[];
    const int smp = blockIdx.x;
    if (only_if && !only_if[smp]) return;
    const int n = counts ? min(counts[smp], cap) : n_host;
    const int lane = threadIdx.x;
    const int ncb_cap = (cap + 63) >> 6;
    const int ncb = (n + 63) >> 6;
    const unsigned long long *gmask = mask_all + (size_t)smp * cap * ncb_cap;
    long long *keep = keep_all + (size_t)smp * keep_cap;
    if (LDS) {
        const int total = n * ncb_cap;                    // words; the sample's mask starts 16-byte aligned (cap * ncb_cap words per sample, a
        const int pairs = total >> 1;                     // 256-byte aligned workspace) and so does s_mask
        const uint4 *g4 = reinterpret_cast<const uint4 *>(gmask);
        uint4 *s4 = reinterpret_cast<uint4 *>(s_mask);
        if ((((size_t)cap * ncb_cap) & 1) == 0) {
            for (int e = threadIdx.x; e < pairs; e += blockDim.x) s4[e] = g4[e];
            if ((total & 1) && threadIdx.x == 0) s_mask[total - 1] = gmask[total - 1];
        } else {
            for (int e = threadIdx.x; e < total; e += blockDim.x) s_mask[e] = gmask[e];
        }
        __syncthreads();
        if (threadIdx.x >= 64) return;                    // (no barrier below this point)
    }
    const unsigned long long *mask = LDS ? s_mask : gmask;
    constexpr int MAXW = 8;  // up to 64*64*8 = 32768 boxes
    unsigned long long remv[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) remv[k] = 0ull;
    int kept = 0;
    for (int nb = 0; nb < ncb; ++nb) {
        unsigned long long cur = 0ull;
#pragma unroll
        for (int k = 0; k < MAXW; ++k)
            if (k == (nb >> 6)) cur = __shfl(remv[k], nb & 63, 64);
        const int lim = n - nb * 64 < 64 ? n - nb * 64 : 64;
        const int myrow = nb * 64 + lane;
        const unsigned long long diag = lane < lim ? mask[(size_t)myrow * ncb_cap + nb] : 0ull;
        unsigned long long keptmask = 0ull;
        for (int ib = 0; ib < lim; ++ib) {
            const unsigned long long d = __shfl(diag, ib, 64);
            if (!(cur & (1ull << ib))) {
                keptmask |= 1ull << ib;
                cur |= d;
            }
        }
        const bool mine = (keptmask >> lane) & 1ull;
        if (mine) keep[kept + __popcll(keptmask & ((1ull << lane) - 1ull))] = myrow;
        kept += __popcll(keptmask);
        if (kept >= max_keep) break;
        // later words: lane i holds row (nb*64+i); fold word w of all kept rows with a butterfly OR
        for (int w = nb + 1; w < ncb; ++w) {
            const unsigned long long v = mine ? mask[(size_t)myrow * ncb_cap + w] : 0ull;
            const unsigned long long all = wave_or64(v);
#pragma unroll
            for (int k = 0; k < MAXW; ++k)
                if (k == (w >> 6) && lane == (w & 63)) remv[k] |= all;
        }
    }
    if (lane == 0) {
        num_keep[smp] = kept;
        if (incomplete) incomplete[smp] = (kept < max_keep && counts && counts[smp] > cap) ? 1 : 0;
    }
}

// `lim` = boxes per sample the mask covers (cap: all of them), `max_keep` = survivors wanted (INT_MAX: all), see nms_scan_kernel
static int nms_impl(bool normal, const float *boxes, const int32_t *counts, int batch, int cap, float thr, int64_t *keep,
                    int32_t *num_keep, void *ws, size_t ws_bytes, hipStream_t s, int lim = -1, int max_keep = 0x7fffffff,
                    int32_t *incomplete = nullptr, const int32_t *only_if = nullptr) {
    if (cap < 0 || batch <= 0 || !keep || !num_keep || (cap > 0 && (!boxes || !ws))) return CPD_ERR_ARG;
    if (cap > 64 * 64 * 8) return CPD_ERR_UNSUPPORTED;
    if (cap == 0) {
        CPD_HIP_TRY(hipMemsetAsync(num_keep, 0, 4 * (size_t)batch, s));
        if (incomplete) CPD_HIP_TRY(hipMemsetAsync(incomplete, 0, 4 * (size_t)batch, s));
        return CPD_OK;
    }
    const int mcap = (lim > 0 && lim < cap) ? lim : cap;            // rows / columns of a sample's mask
    if (ws_bytes < (size_t)batch * cpd_nms_workspace_bytes(mcap)) return CPD_ERR_WORKSPACE;
    const int ncb = (mcap + 63) / 64;
    unsigned long long *mask = (unsigned long long *)ws;
    dim3 grid(ncb, (mcap + 3) / 4, batch);
    if (normal) nms_mask_kernel<true><<<grid, 256, 0, s>>>(boxes, counts, mcap, mcap, cap, thr, mask, only_if);
    else nms_mask_kernel<false><<<grid, 256, 0, s>>>(boxes, counts, mcap, mcap, cap, thr, mask, only_if);
    const size_t lds = (size_t)mcap * ncb * 8;
    if (lds <= 64 * 1024)
        nms_scan_kernel<true><<<batch, 256, lds, s>>>(mask, counts, mcap, mcap, cap, max_keep, (long long *)keep, num_keep, incomplete, only_if);
    else
        nms_scan_kernel<false><<<batch, 64, 0, s>>>(mask, counts, mcap, mcap, cap, max_keep, (long long *)keep, num_keep, incomplete, only_if);
    return cpd_check_launch();
}

// class_agnostic_nms tail (model_nms_utils.py:126-127): out[k] = in[keep[k]] for k < min(num_keep, post_max)
__global__ void __launch_bounds__(256) select_boxes_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                           const int32_t *__restrict__ labels,
                                                           const long long *__restrict__ keep,
                                                           const int32_t *__restrict__ num_keep, int cap, int post_max,
                                                           int label_offset, float *__restrict__ out_boxes,
                                                           float *__restrict__ out_scores, long long *__restrict__ out_labels,
                                                           int32_t *__restrict__ out_n) {
    const int smp = blockIdx.y;
    const int nk = min(num_keep[smp], post_max);
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) out_n[smp] = nk;
    if (k >= nk) return;
    const long long src = keep[(size_t)smp * cap + k];
    const float *b = boxes + ((size_t)smp * cap + src) * 7;
    float *o = out_boxes + ((size_t)smp * post_max + k) * 7;
#pragma unroll
    for (int c = 0; c < 7; ++c) o[c] = b[c];
    out_scores[(size_t)smp * post_max + k] = scores[(size_t)smp * cap + src];
    out_labels[(size_t)smp * post_max + k] = (long long)labels[(size_t)smp * cap + src] + label_offset;
}

// post_processing's score pipeline (detector3d_template.py:222-343 with MULTI_CLASSES_NMS False; model_nms_utils.class_agnostic_nms
// l.113-124) for all frames in one launch, one workgroup per frame: score = max_c sigmoid(cls[c]) (or cls as given when `normalized`),
// ok = score >= thresh, rows ranked by score descending with ties to the lower index (a stable descending sort), the not-ok rows last
// with score -1; boxes / labels gathered in that order; n_ok = min(#ok, pre_max). The rank of a row is COUNTED against all others
// (R^2 comparisons from LDS: 250 k at 500 RoIs) -- no sort passes, deterministic.
__global__ void __launch_bounds__(256) rank_scores_kernel(const float *__restrict__ cls, int n_cls, const float *__restrict__ boxes,
                                                          const long long *__restrict__ labels, int r, float thresh, int pre_max, int normalized,
                                                          float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                          int32_t *__restrict__ out_labels, int32_t *__restrict__ n_ok) {
    extern __shared__ float key[];                 // [r]
    const int smp = blockIdx.x;
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < r; i += blockDim.x) {
        const float *c = cls + ((size_t)smp * r + i) * n_cls;
        float sc = -INFINITY;
        for (int k = 0; k < n_cls; ++k) {
            const float v = normalized ? c[k] : __fdiv_rn(1.f, 1.f + expf(-c[k]));
            sc = v > sc ? v : sc;
        }
        const bool ok = sc >= thresh;
        key[i] = ok ? sc : -1.f;
        mine += ok ? 1 : 0;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) n_ok[smp] = cnt < pre_max ? cnt : pre_max;
    for (int i = threadIdx.x; i < r; i += blockDim.x) {
        const float ki = key[i];
        int rank = 0;
        for (int j = 0; j < r; ++j) {
            const float kj = key[j];
            rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
        }
        const float *b = boxes + ((size_t)smp * r + i) * 7;
        float *o = out_boxes + ((size_t)smp * r + rank) * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c) o[c] = b[c];
        out_scores[(size_t)smp * r + rank] = ki;
        out_labels[(size_t)smp * r + rank] = (int32_t)labels[(size_t)smp * r + i];
    }
}

template <int MODE>
static int pairwise_impl(const float *a, int n, const float *b, int m, float *out, hipStream_t s) {
    if (n < 0 || m < 0 || (n > 0 && m > 0 && (!a || !b || !out))) return CPD_ERR_ARG;
    if (n == 0 || m == 0) return CPD_OK;
    dim3 grid(cpd_div_up(m, 64), cpd_div_up(n, 64));
    pairwise_kernel<MODE><<<grid, 256, 0, s>>>(a, n, b, m, out);
    return cpd_check_launch();
}

}  // namespace

extern "C" int cpd_boxes_overlap_bev(const float *a, int n, const float *b, int m, float *out, cpd_stream_t stream) {
    return pairwise_impl<MODE_OVERLAP>(a, n, b, m, out, cpd_s(stream));
}
extern "C" int cpd_boxes_iou_bev(const float *a, int n, const float *b, int m, float *out, cpd_stream_t stream) {
    return pairwise_impl<MODE_IOU_BEV>(a, n, b, m, out, cpd_s(stream));
}
extern "C" int cpd_boxes_iou3d(const float *a, int n, const float *b, int m, float *out, cpd_stream_t stream) {
    return pairwise_impl<MODE_IOU3D>(a, n, b, m, out, cpd_s(stream));
}

extern "C" size_t cpd_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    return cpd_align((size_t)n * ((n + 63) / 64) * 8);
}
extern "C" int cpd_nms_rotated(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep, void *workspace,
                               size_t workspace_bytes, cpd_stream_t stream) {
    return nms_impl(false, boxes, nullptr, 1, n, thresh, keep, num_keep, workspace, workspace_bytes, cpd_s(stream));
}
extern "C" int cpd_nms_normal(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep, void *workspace,
                              size_t workspace_bytes, cpd_stream_t stream) {
    return nms_impl(true, boxes, nullptr, 1, n, thresh, keep, num_keep, workspace, workspace_bytes, cpd_s(stream));
}
extern "C" int cpd_nms_batch(const float *boxes, const int32_t *counts, int batch, int capacity, float thresh, int normal,
                             int64_t *keep, int32_t *num_keep, void *workspace, size_t workspace_bytes,
                             cpd_stream_t stream) {
    if (!counts) return CPD_ERR_ARG;
    return nms_impl(normal != 0, boxes, counts, batch, capacity, thresh, keep, num_keep, workspace, workspace_bytes,
                    cpd_s(stream));
}
extern "C" int cpd_nms_batch_first(const float *boxes, const int32_t *counts, int batch, int capacity, float thresh, int normal,
                                   int max_keep, int row_limit, int64_t *keep, int32_t *num_keep, int32_t *incomplete, void *workspace,
                                   size_t workspace_bytes, cpd_stream_t stream) {
    if (!counts || !incomplete || max_keep <= 0 || row_limit <= 0) return CPD_ERR_ARG;
    return nms_impl(normal != 0, boxes, counts, batch, capacity, thresh, keep, num_keep, workspace, workspace_bytes, cpd_s(stream), row_limit,
                    max_keep, incomplete);
}
extern "C" int cpd_nms_batch_where(const float *boxes, const int32_t *counts, const int32_t *where, int batch, int capacity, float thresh,
                                   int normal, int64_t *keep, int32_t *num_keep, void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    if (!counts || !where || capacity <= 0) return CPD_ERR_ARG;
    return nms_impl(normal != 0, boxes, counts, batch, capacity, thresh, keep, num_keep, workspace, workspace_bytes, cpd_s(stream), -1,
                    0x7fffffff, nullptr, where);
}
extern "C" int cpd_select_boxes(const float *boxes, const float *scores, const int32_t *labels, const int64_t *keep,
                                const int32_t *num_keep, int batch, int capacity, int post_max, int label_offset,
                                float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_n,
                                cpd_stream_t stream) {
    if (!boxes || !scores || !labels || !keep || !num_keep || !out_boxes || !out_scores || !out_labels || !out_n ||
        batch <= 0 || capacity <= 0 || post_max <= 0)
        return CPD_ERR_ARG;
    dim3 grid(cpd_div_up(post_max, 256), batch);
    select_boxes_kernel<<<grid, 256, 0, cpd_s(stream)>>>(boxes, scores, labels, (const long long *)keep, num_keep, capacity,
                                                         post_max, label_offset, out_boxes, out_scores,
                                                         (long long *)out_labels, out_n);
    return cpd_check_launch();
}

extern "C" int cpd_rank_scores(const float *cls, int n_cls, const float *boxes, const void *labels_i64, int batch, int r, float score_thresh,
                               int pre_max, int normalized, float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *n_ok,
                               cpd_stream_t stream) {
    if (batch <= 0 || r <= 0 || n_cls <= 0 || pre_max <= 0 || !cls || !boxes || !labels_i64 || !out_boxes || !out_scores || !out_labels || !n_ok)
        return CPD_ERR_ARG;
    if (r > 8192) return CPD_ERR_UNSUPPORTED;           // (the keys of a frame sit in LDS; the counting rank is quadratic)
    rank_scores_kernel<<<batch, 256, (size_t)r * sizeof(float), cpd_s(stream)>>>(cls, n_cls, boxes, (const long long *)labels_i64, r, score_thresh,
                                                                                pre_max, normalized, out_boxes, out_scores, out_labels, n_ok);
    return cpd_check_launch();
}

extern "C" int cpd_boxes_iou_bev_cpu(const float *a, int n, const float *b, int m, float *out) {
    if (n < 0 || m < 0 || (n > 0 && m > 0 && (!a || !b || !out))) return CPD_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        BoxG A;
        box_setup(a + 7 * (size_t)i, A);
        for (int j = 0; j < m; ++j) {
            BoxG B;
            box_setup(b + 7 * (size_t)j, B);
            out[(size_t)i * m + j] = iou_bev_g(A, B);
        }
    }
    return CPD_OK;
}
