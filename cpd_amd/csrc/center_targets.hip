// center_targets.hip -- CenterHead target assignment on the device, for gfx950: heat maps, regression targets, indices and masks of
// every sample of a batch in three launches, nothing read back.
//
// Replaces CenterHead.assign_targets / assign_target_of_single_head (cpd/models/dense_heads/center_head.py:103-219) with
// centernet_utils.gaussian_radius / draw_gaussian_to_heatmap (cpd/models/model_utils/centernet_utils.py:9-69). The reference walks the
// boxes of a sample one by one on the CPU (l.204); the torch restatement in cpd_amd/center_loss.py rasterises all patches at once --
// ~50 small launches and two host read-backs (is there a valid box? the largest radius), which in the train step sat between the
// forward and the backward pass. Here:
//   targets_clear_kernel   zeroes heat / target / inds / masks (grid-stride, 16-byte stores)
//   targets_boxes_kernel   one workgroup per sample: the head's boxes (class >= 1) compacted to the front in their order (l.180-196),
//                          the first K of that list (l.113) turned into centre pixel, gaussian radius, regression target, index, mask;
//                          the patch parameters of each slot go to the workspace
//   targets_draw_kernel    one workgroup per (slot, sample): exp(-(x^2 + y^2) / (2 sigma^2)), sigma = (2 r + 1) / 6, in double like numpy,
//                          rounded to float, max-merged into the class map with an atomic max on the bit pattern (values are >= 0:
//                          unsigned order = float order) -- order-independent, so the map is deterministic
// Compiled with -ffp-contract=off (Makefile): radii and centres follow the reference's fp32 operation order (a fused multiply-add in
// b^2 - 4ac moves a radius across an integer now and then).
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace {

struct TgParams {
    const float *gt;          // [batch][m][8]
    int batch, m, num_classes, h, w, k;
    float x0, y0, vx, vy, stride;
    double overlap;
    int min_radius;
    float *heat;              // [batch][num_classes][h][w]
    float *target;            // [batch][k][8]
    long long *inds, *masks;  // [batch][k]
    int4 *patch;              // [batch][k]: centre x, centre y, radius, class (-1: nothing to draw)
};

struct ClearJob {
    uint4 *p[4];
    size_t end16[4];          // running ends, in 16-byte pieces
    unsigned char *tail[4];
    int n_tail[4];
};

__global__ void __launch_bounds__(256) targets_clear_kernel(ClearJob j) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < j.end16[3]; i += stride) {
        const int q = (i >= j.end16[0]) + (i >= j.end16[1]) + (i >= j.end16[2]);
        j.p[q][i - (q ? j.end16[q - 1] : 0)] = uint4{0u, 0u, 0u, 0u};
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int q = threadIdx.x >> 4, e = threadIdx.x & 15;
        if (e < j.n_tail[q]) j.tail[q][e] = 0;
    }
}

// centernet_utils.py:9-31, height / width as the reference passes them (dx, dy in feature-map pixels), every operation in fp32 in the
// order python evaluates it: scalars folded in double first where python folds them (4 * a, -2 * min_overlap), then rounded to fp32
__device__ __forceinline__ float gaussian_radius(float height, float width, double mo) {
    const float b1 = height + width;
    const float c1 = width * height * (float)(1.0 - mo) / (float)(1.0 + mo);
    const float sq1 = sqrtf(b1 * b1 - 4.f * c1);
    const float r1 = (b1 + sq1) / 2.f;
    const float b2 = 2.f * (height + width);
    const float c2 = (float)(1.0 - mo) * width * height;
    const float sq2 = sqrtf(b2 * b2 - 16.f * c2);
    const float r2 = (b2 + sq2) / 2.f;
    const float a3 = (float)(4.0 * mo);
    const float b3 = (float)(-2.0 * mo) * (height + width);
    const float c3 = (float)(mo - 1.0) * width * height;
    const float sq3 = sqrtf(b3 * b3 - (float)(4.0 * (4.0 * mo)) * c3);
    const float r3 = (b3 + sq3) / 2.f;
    (void)a3;
    return fminf(fminf(r1, r2), r3);
}

__global__ void __launch_bounds__(256) targets_boxes_kernel(TgParams p) {
    extern __shared__ int s_pos[];              // slot of box j in the compacted list
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *gt = p.gt + (size_t)b * p.m * 8;
    if (tid == 0) {                              // stable partition: boxes of this head first, the rest after them (each in its order)
        int kept = 0;
        for (int j = 0; j < p.m; ++j) kept += !(gt[(size_t)j * 8 + 7] < 1.f);
        int a = 0, d = kept;
        for (int j = 0; j < p.m; ++j) s_pos[j] = !(gt[(size_t)j * 8 + 7] < 1.f) ? a++ : d++;
    }
    __syncthreads();
    for (int j = tid; j < p.m; j += blockDim.x) {
        const int slot = s_pos[j];
        if (slot >= p.k) continue;
        const float *g = gt + (size_t)j * 8;
        const float x = g[0], y = g[1], z = g[2], cls = g[7];
        // center_head.py:120-127
        float cx = (x - p.x0) / p.vx / p.stride, cy = (y - p.y0) / p.vy / p.stride;
        cx = fminf(fmaxf(cx, 0.f), (float)p.w - 0.5f);            // (a NaN coordinate: fmaxf gives 0 where torch.clamp keeps the NaN -- the
        cy = fminf(fmaxf(cy, 0.f), (float)p.h - 0.5f);            //  reference's int() of it is undefined either way)
        const int cxi = (int)cx, cyi = (int)cy;
        const float dx = g[3] / p.vx / p.stride, dy = g[4] / p.vy / p.stride;
        const bool valid = dx > 0.f && dy > 0.f && cls >= 1.f;
        int radius = 0;
        if (valid) {
            const int r = (int)gaussian_radius(fmaxf(dx, 1e-6f), fmaxf(dy, 1e-6f), p.overlap);
            radius = r > p.min_radius ? r : p.min_radius;
        }
        const float v = valid ? 1.f : 0.f;
        float *t = p.target + ((size_t)b * p.k + slot) * 8;
        t[0] = (cx - (float)cxi) * v;
        t[1] = (cy - (float)cyi) * v;
        t[2] = z * v;
        t[3] = logf(fmaxf(g[3], 1e-12f)) * v;
        t[4] = logf(fmaxf(g[4], 1e-12f)) * v;
        t[5] = logf(fmaxf(g[5], 1e-12f)) * v;
        t[6] = cosf(g[6]) * v;
        t[7] = sinf(g[6]) * v;
        p.inds[(size_t)b * p.k + slot] = valid ? (long long)cyi * p.w + cxi : 0;
        p.masks[(size_t)b * p.k + slot] = valid ? 1 : 0;
        int c = (int)cls - 1;
        c = c < 0 ? 0 : (c >= p.num_classes ? p.num_classes - 1 : c);
        p.patch[(size_t)b * p.k + slot] = valid ? int4{cxi, cyi, radius, c} : int4{0, 0, 0, -1};
    }
    // slots beyond the sample's boxes (m < k) stay as the clear kernel left them; their patch entries say "nothing"
    for (int s = p.m + tid; s < p.k; s += blockDim.x) p.patch[(size_t)b * p.k + s] = int4{0, 0, 0, -1};
}

__global__ void __launch_bounds__(64) targets_draw_kernel(TgParams p) {
    const int slot = blockIdx.x, b = blockIdx.y;
    const int4 q = p.patch[(size_t)b * p.k + slot];
    if (q.w < 0) return;
    const int r = q.z, d = 2 * r + 1;
    const double sigma = (double)d / 6.0;
    const double den = 2.0 * sigma * sigma;
    unsigned int *map = reinterpret_cast<unsigned int *>(p.heat + ((size_t)b * p.num_classes + q.w) * p.h * p.w);
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        const int oy = e / d - r, ox = e - (e / d) * d - r;
        const int px = q.x + ox, py = q.y + oy;
        if (px < 0 || px >= p.w || py < 0 || py >= p.h) continue;
        const float val = (float)exp(-(double)(ox * ox + oy * oy) / den);
        atomicMax(map + (size_t)py * p.w + px, __float_as_uint(val));
    }
}

}  // namespace

extern "C" size_t cpd_center_targets_workspace_bytes(int batch, int k) {
    if (batch <= 0 || k <= 0) return 0;
    return cpd_align((size_t)batch * k * sizeof(int4));
}

extern "C" int cpd_center_targets(const float *gt_boxes, int batch, int m, int num_classes, int h, int w, int k, const float pc_range_xy[2],
                                  const float voxel_xy[2], int feature_map_stride, double gaussian_overlap, int min_radius, float *heat,
                                  float *target, int64_t *inds, int64_t *masks, void *ws, size_t ws_bytes, cpd_stream_t st) {
    if (batch <= 0 || m < 0 || num_classes <= 0 || h <= 0 || w <= 0 || k <= 0 || feature_map_stride <= 0 || !heat || !target || !inds ||
        !masks || !pc_range_xy || !voxel_xy || (m > 0 && !gt_boxes))
        return CPD_ERR_ARG;
    if (m > 12000) return CPD_ERR_UNSUPPORTED;                   // the compaction's slot table lives in LDS
    if (ws_bytes < cpd_center_targets_workspace_bytes(batch, k) || !ws) return CPD_ERR_WORKSPACE;
    TgParams p;
    p.gt = gt_boxes; p.batch = batch; p.m = m; p.num_classes = num_classes; p.h = h; p.w = w; p.k = k;
    p.x0 = pc_range_xy[0]; p.y0 = pc_range_xy[1]; p.vx = voxel_xy[0]; p.vy = voxel_xy[1]; p.stride = (float)feature_map_stride;
    p.overlap = gaussian_overlap; p.min_radius = min_radius;
    p.heat = heat; p.target = target; p.inds = reinterpret_cast<long long *>(inds); p.masks = reinterpret_cast<long long *>(masks);
    p.patch = reinterpret_cast<int4 *>(ws);
    // the four outputs are cleared by one launch: whole 16-byte pieces of each, then the odd bytes
    struct { void *ptr; size_t bytes; } outs[4] = {{heat, (size_t)batch * num_classes * h * w * 4}, {target, (size_t)batch * k * 32},
                                                   {inds, (size_t)batch * k * 8}, {masks, (size_t)batch * k * 8}};
    ClearJob cj;
    size_t run = 0;
    for (int q = 0; q < 4; ++q) {
        if (((uintptr_t)outs[q].ptr & 15) != 0) return CPD_ERR_ARG;
        const size_t n16 = outs[q].bytes / 16;
        cj.p[q] = reinterpret_cast<uint4 *>(outs[q].ptr);
        run += n16;
        cj.end16[q] = run;
        cj.tail[q] = reinterpret_cast<unsigned char *>(outs[q].ptr) + n16 * 16;
        cj.n_tail[q] = (int)(outs[q].bytes - n16 * 16);
    }
    int blocks = (int)((run + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    targets_clear_kernel<<<blocks, 256, 0, cpd_s(st)>>>(cj);
    const int mk = m < k ? m : k;
    targets_boxes_kernel<<<batch, 256, (size_t)(m > 0 ? m : 1) * sizeof(int), cpd_s(st)>>>(p);
    if (mk > 0) targets_draw_kernel<<<dim3(mk, batch), 64, 0, cpd_s(st)>>>(p);
    return cpd_check_launch();
}
