// train_ops.hip -- kernels the TRAIN step (BASELINE config 3) needs on top of the forward path:
//   * column reductions over [n, c] row tensors: BatchNorm batch statistics (training-mode
//     nn.BatchNorm1d/2d, spconv_backbone.py:410, base_bev_backbone.py:38, center_head.py:24,78) and
//     the two sums of BatchNorm backward, deterministic two-stage (no atomics);
//   * fused row-wise affine (+residual)(+ReLU)  = BN apply in training mode;
//   * BatchNorm(+ReLU) backward apply;
//   * weight gradient of the rulebook convolution (cpd_gather_conv's adjoint w.r.t. W) on the
//     fp32 matrix pipe, 4-row skip granularity, deterministic two-stage reduction over row chunks;
//   * transposed rulebooks for the input gradient of strided convs (sparse and 2-D);
//   * Adam step on a flat parameter buffer (tools/train_utils/optimization: adam_onecycle without
//     the one-cycle schedule; plain decoupled weight decay).
// The input gradient itself is cpd_gather_conv again, on transposed (and, for SubM / stride-1
// convs, tap-flipped) weights with the same or the transposed rulebook.
#include "common.h"
#include "site_index_layout.h"

#ifndef CPD_WG_ABLATE
#define CPD_WG_ABLATE 0      // diagnostic builds of the split weight-gradient kernel (wrong results, timing only): 1 no row loads, 2 no MFMAs, 4 no split / LDS image writes
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

enum { COL_SUM = 0, COL_STATS = 1, COL_BNBWD = 2 };

struct ColParams {
    const float *a;      // COL_SUM/COL_STATS: x ; COL_BNBWD: dy
    const float *y;      // COL_BNBWD: BN(+ReLU) output, for the ReLU mask (may be null: no ReLU)
    const float *x;      // COL_BNBWD: BN input
    const float *mean, *invstd;
    int lda, ldy, ldx;
    int n, c, rows_per_chunk;
    int cq;              // column groups (of VEC columns) per block; 256 / cq row lanes
};

// Block (col tile, row chunk): thread = (column group q, row lane rl); a thread streams VEC
// consecutive columns (one 16-byte load for VEC = 4) of every (256/cq)-th row of the chunk, four
// rows in flight. Partials land in part[chunk][2][c]; col_final_kernel sums the chunks.
template <int MODE, int VEC>
__global__ void __launch_bounds__(256) col_partial_kernel(ColParams p, float *__restrict__ part) {
    __shared__ float sm[2 * VEC * 256];
    const int CQ = p.cq, RL = 256 / CQ;
    const int q = threadIdx.x % CQ, rl = threadIdx.x / CQ;
    const int col = (blockIdx.x * CQ + q) * VEC;
    const int r0 = blockIdx.y * p.rows_per_chunk;
    const int r1 = min(p.n, r0 + p.rows_per_chunk);
    float s1[VEC], s2[VEC], mu[VEC], is[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s1[v] = 0.f; s2[v] = 0.f; mu[v] = 0.f; is[v] = 1.f; }
    const bool live = col < p.c && rl < RL;
    if (live) {
        if (MODE == COL_BNBWD) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (col + v < p.c) { mu[v] = p.mean[col + v]; is[v] = p.invstd[col + v]; }
        }
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += RL) {
            float av[VEC], yv[VEC], xv[VEC];
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p.a + (size_t)r * p.lda + col);
                av[0] = t.x; av[1] = t.y; av[2] = t.z; av[VEC - 1] = t.w;
                if (MODE == COL_BNBWD) {
                    const float4 u = *reinterpret_cast<const float4 *>(p.x + (size_t)r * p.ldx + col);
                    xv[0] = u.x; xv[1] = u.y; xv[2] = u.z; xv[VEC - 1] = u.w;
                    if (p.y) {
                        const float4 w = *reinterpret_cast<const float4 *>(p.y + (size_t)r * p.ldy + col);
                        yv[0] = w.x; yv[1] = w.y; yv[2] = w.z; yv[VEC - 1] = w.w;
                    }
                }
            } else {
                av[0] = p.a[(size_t)r * p.lda + col];
                if (MODE == COL_BNBWD) {
                    xv[0] = p.x[(size_t)r * p.ldx + col];
                    if (p.y) yv[0] = p.y[(size_t)r * p.ldy + col];
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float g = av[v];
                if (MODE == COL_SUM) {
                    s1[v] += g;
                } else if (MODE == COL_STATS) {
                    s1[v] += g; s2[v] += g * g;
                } else {
                    if (p.y && !(yv[v] > 0.f)) g = 0.f;
                    s1[v] += g; s2[v] += g * ((xv[v] - mu[v]) * is[v]);
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { sm[(2 * v) * 256 + threadIdx.x] = s1[v]; sm[(2 * v + 1) * 256 + threadIdx.x] = s2[v]; }
    __syncthreads();
    if (live && rl == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float a = s1[v], b = s2[v];
            for (int k = 1; k < RL; ++k) { a += sm[(2 * v) * 256 + k * CQ + q]; b += sm[(2 * v + 1) * 256 + k * CQ + q]; }
            if (col + v < p.c) {
                part[((size_t)blockIdx.y * 2 + 0) * p.c + col + v] = a;
                part[((size_t)blockIdx.y * 2 + 1) * p.c + col + v] = b;
            }
        }
    }
}

struct BnFin {
    int n; float eps, momentum;
    const float *gamma, *beta;
    float *mean, *invstd, *scale, *shift, *running_mean, *running_var;
};
__device__ __forceinline__ void bn_finalize_col(const BnFin &f, int col, double sum, double sumsq) {
    const double mu = sum / f.n;
    double var = sumsq / f.n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)f.eps));
    f.mean[col] = (float)mu;
    f.invstd[col] = is;
    const float sc = f.gamma[col] * is;
    f.scale[col] = sc;
    f.shift[col] = f.beta[col] - (float)mu * sc;
    if (f.running_mean) {
        const double unbiased = f.n > 1 ? var * f.n / (f.n - 1.0) : var;
        f.running_mean[col] = (1.f - f.momentum) * f.running_mean[col] + f.momentum * (float)mu;
        f.running_var[col] = (1.f - f.momentum) * f.running_var[col] + f.momentum * (float)unbiased;
    }
}

// 16 columns x 16 lanes per block; lanes stride over the chunk partials, accumulate in double.
template <bool FIN>
__global__ void __launch_bounds__(256) col_final_kernel(const float *__restrict__ part, int nb, int c, float *__restrict__ s1,
                                                        float *__restrict__ s2, BnFin fin) {
    __shared__ double sa[256], sb[256];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (col < c)
        for (int k = lane; k < nb; k += 16) { a += part[((size_t)k * 2 + 0) * c + col]; b += part[((size_t)k * 2 + 1) * c + col]; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    if (lane == 0 && col < c) {
        for (int k = 1; k < 16; ++k) { a += sa[k * 16 + cl]; b += sb[k * 16 + cl]; }
        if (FIN) {
            bn_finalize_col(fin, col, a, b);
        } else {
            s1[col] = (float)a;
            if (s2) s2[col] = (float)b;
        }
    }
}

// V = 4: 16-byte accesses (c and the row pitches multiples of 4, 16-byte aligned bases), else V = 1
template <int V>
__global__ void __launch_bounds__(256) affine_rows_kernel(const float *__restrict__ x, int ldx, int n, int c,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          const float *__restrict__ res, int ldr, int relu,
                                                          float *__restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cv = c / V;
    const long long total = (long long)n * cv;
    if (i >= total) return;
    const int r = (int)(i / cv), col = (int)(i - (long long)r * cv) * V;
    float v[V], rv[V];
    if (V == 4) {
        *reinterpret_cast<f32x4 *>(v) = *reinterpret_cast<const f32x4 *>(x + (size_t)r * ldx + col);
        if (res) *reinterpret_cast<f32x4 *>(rv) = *reinterpret_cast<const f32x4 *>(res + (size_t)r * ldr + col);
    } else {
        v[0] = x[(size_t)r * ldx + col];
        if (res) rv[0] = res[(size_t)r * ldr + col];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        if (scale) v[e] *= scale[col + e];
        if (shift) v[e] += shift[col + e];
        if (res) v[e] += rv[e];
        if (relu && !(v[e] > 0.f)) v[e] = 0.f;
    }
    if (V == 4) *reinterpret_cast<f32x4 *>(out + (size_t)r * ldo + col) = *reinterpret_cast<const f32x4 *>(v);
    else out[(size_t)r * ldo + col] = v[0];
}

// dx = a * (dy_m - s1/n - xhat * s2/n), a = gamma*invstd ; dy_m = dy masked by (y > 0);
// dres (optional) = dy_m, the gradient flowing into a residual input added before the ReLU.
// V = 4: 16-byte accesses (c and every row pitch multiples of 4, 16-byte aligned bases), else V = 1. One item per thread (a
// grid-stride form with fewer, longer threads measured 2x slower: 29.8 vs 14.7 us on a 188 x 188 x 128 map); the max |dx| word
// costs one atomic per workgroup.
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ y,
                                                           int ldy, const float *__restrict__ x, int ldx, int n, int c,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ s1,
                                                           const float *__restrict__ s2, float *__restrict__ dx, int lddx,
                                                           float *__restrict__ dres, int lddres, uint32_t *__restrict__ dx_absmax,
                                                           const float *__restrict__ n_stat) {
    const int cv = c / V;
    const long long total = (long long)n * cv;
    // n_stat (SyncBN, cpd_bn_bwd_apply_sync): the row count the statistics -- and s1 / s2 -- were summed over (all ranks'), on the device
    const float inv_n = 1.0f / (n_stat ? *n_stat : (float)n);
    float vmax = 0.f;
    bool bad = false;                       // a NaN anywhere must reach the word (it switches the fp16 scaling off, loudly)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cv), col = (int)(i - (long long)r * cv) * V;
        float g[V], yv[V], xv[V], o[V];
        if (V == 4) {
            *reinterpret_cast<f32x4 *>(g) = *reinterpret_cast<const f32x4 *>(dy + (size_t)r * lddy + col);
            if (y) *reinterpret_cast<f32x4 *>(yv) = *reinterpret_cast<const f32x4 *>(y + (size_t)r * ldy + col);
            *reinterpret_cast<f32x4 *>(xv) = *reinterpret_cast<const f32x4 *>(x + (size_t)r * ldx + col);
        } else {
            g[0] = dy[(size_t)r * lddy + col];
            if (y) yv[0] = y[(size_t)r * ldy + col];
            xv[0] = x[(size_t)r * ldx + col];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (y && !(yv[e] > 0.f)) g[e] = 0.f;
            const float is = invstd[col + e];
            const float xh = (xv[e] - mean[col + e]) * is;
            o[e] = gamma[col + e] * is * (g[e] - s1[col + e] * inv_n - xh * s2[col + e] * inv_n);
            vmax = fmaxf(vmax, fabsf(o[e]));
            bad |= o[e] != o[e];
        }
        if (V == 4) {
            if (dres) *reinterpret_cast<f32x4 *>(dres + (size_t)r * lddres + col) = *reinterpret_cast<const f32x4 *>(g);
            *reinterpret_cast<f32x4 *>(dx + (size_t)r * lddx + col) = *reinterpret_cast<const f32x4 *>(o);
        } else {
            if (dres) dres[(size_t)r * lddres + col] = g[0];
            dx[(size_t)r * lddx + col] = o[0];
        }
    }
    // max |dx| of the whole tensor (bits of a non-negative float order like unsigned integers): the split-fp16 gradient
    // kernels that consume dx scale it by a power of two derived from this word (cpd_gather_conv_scaled, cpd_conv_wgrad_scaled)
    if (dx_absmax) {
        __shared__ uint32_t wave_max[4];
        uint32_t m = bad ? 0x7fc00000u : __float_as_uint(vmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
            m = t > m ? t : m;
        }
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            // one atomic per workgroup, spread over CPD_ABSMAX_SLOTS words in different 128-byte lines: atomics on one line
            // serialise at ~12 ns each (4418 workgroups: +48 us on one word, +1.7 us on 16 lines; a guarding read only adds
            // latency -- tools/atomic_probe.hip)
            m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
            atomicMax(dx_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE, m);
        }
    }
}

// ReLU backward for layers without BatchNorm in between: dx = dy * (y > 0)
__global__ void __launch_bounds__(256) relu_bwd_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ y,
                                                       int ldy, int n, int c, float *__restrict__ dx, int lddx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * c) return;
    const int r = (int)(i / c), col = (int)(i - (long long)r * c);
    dx[(size_t)r * lddx + col] = y[(size_t)r * ldy + col] > 0.f ? dy[(size_t)r * lddy + col] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[t][ci][co] = sum_j in[nbr[t][j]][ci] * dy[j][co].
// One wave = one (row chunk, tap, ci tile, co tile); MFMA 16x16x4 with K = 4 rows per step:
// lane (r, g) feeds A with VA consecutive channels of gathered row g and B with VB consecutive
// channels of dy row g, so one step is VA*VB MFMAs on a (16*VA) x (16*VB) tile. 4-row groups
// with no neighbour are skipped (wave-uniform). Row-chunk partials are summed by a second kernel.
// ---------------------------------------------------------------------------------------------
struct WgParams {
    const float *in;
    const float *dy;
    const int32_t *nbr;
    float *part;
    int in_ld, dy_ld, c_in, c_out, kv, n_out;
    int rows_per_chunk, n_chunks, ci_tiles, co_tiles;
    const uint32_t *in_absmax, *dy_absmax;   // split-fp16 kernel only (or NULL): bits of max |in| / max |dy|, see in_pow2_scale
};

template <int VA, int VB>
__global__ void __launch_bounds__(64) wgrad_kernel(WgParams p) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    int id = blockIdx.x;
    const int cot = id % p.co_tiles; id /= p.co_tiles;
    const int cit = id % p.ci_tiles; id /= p.ci_tiles;
    const int t = id % p.kv;
    const int chunk = id / p.kv;
    const int r0 = chunk * p.rows_per_chunk, r1 = min(p.n_out, r0 + p.rows_per_chunk);
    const int ci0 = cit * 16 * VA, co0 = cot * 16 * VB;

    f32x4 acc[VA][VB];
#pragma unroll
    for (int a = 0; a < VA; ++a)
#pragma unroll
        for (int b = 0; b < VB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j0 = r0; j0 < r1; j0 += 64) {
        // rulebook entries of 64 rows in one coalesced load, then 16 steps of 4 rows
        const int jr = j0 + lane;
        int idxv = -1;
        if (jr < r1) idxv = p.nbr ? p.nbr[(size_t)t * p.n_out + jr] : jr;
        if (!__any(idxv >= 0)) continue;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int idx = __shfl(idxv, 4 * s + g, 64);
            if (!__any(idx >= 0)) continue;
            const int row = j0 + 4 * s + g;
            float av[VA], bv[VB];
            const float *ap = p.in + (size_t)(idx < 0 ? 0 : idx) * p.in_ld + ci0 + VA * r;
            const float *bp = p.dy + (size_t)(row < r1 ? row : r0) * p.dy_ld + co0 + VB * r;
#pragma unroll
            for (int a = 0; a < VA; ++a) av[a] = (idx >= 0 && ci0 + VA * r + a < p.c_in) ? ap[a] : 0.f;
#pragma unroll
            for (int b = 0; b < VB; ++b) bv[b] = (row < r1 && co0 + VB * r + b < p.c_out) ? bp[b] : 0.f;
#pragma unroll
            for (int a = 0; a < VA; ++a)
#pragma unroll
                for (int b = 0; b < VB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // D[i][j]: i = 4g+e <-> channel ci0 + VA*i + a ; j = r <-> column co0 + VB*r + b
    float *out = p.part + ((size_t)chunk * p.kv + t) * p.c_in * p.c_out;
#pragma unroll
    for (int a = 0; a < VA; ++a)
#pragma unroll
        for (int b = 0; b < VB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + VA * (4 * g + e) + a, co = co0 + VB * r + b;
                if (ci < p.c_in && co < p.c_out) out[(size_t)ci * p.c_out + co] = acc[a][b][e];
            }
}

// Workgroup variant for channel counts that are multiples of 64: 4 waves (2 x 2) share a TM x TN
// (ci x co) tile of one tap over one row chunk. Per stage of 16 rows the gathered input rows and the
// dy rows go global -> registers -> LDS (double buffered, one barrier per stage, the next stage's
// loads in flight under the MFMAs); a wave feeds its (TM/2) x (TN/2) sub-tile with one 16-byte LDS
// read per operand side and 4-row step. Cuts the L2 traffic of the wave kernel by the tile's reuse
// (every row is fetched once per 128 instead of once per 64 channels of the other operand).
template <int TM, int TN>
__global__ void __launch_bounds__(256) wgrad_tile_kernel(WgParams p) {
    constexpr int KB = 16;                       // rows per stage
    constexpr int LDA = TM + 4, LDB = TN + 4;    // +16 B: the four k-groups of a read hit different banks
    constexpr int VA = TM / 32, VB = TN / 32;    // 16-wide MFMA tiles per wave and side
    constexpr int PA = KB * TM / 4 / 256, PB = KB * TN / 4 / 256;   // 16-byte pieces per thread and stage
    static_assert(PA >= 1 && PB >= 1, "tile too small for 256 threads");
    __shared__ float sA[2][KB * LDA];
    __shared__ float sB[2][KB * LDB];
    __shared__ __attribute__((aligned(16))) int sValid[2][KB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.co_tiles;
    const int ci0 = (blockIdx.x / tiles_n) * TM, co0 = (blockIdx.x % tiles_n) * TN;
    const int t = blockIdx.y;
    const int r0 = blockIdx.z * p.rows_per_chunk, r1 = min(p.n_out, r0 + p.rows_per_chunk);
    const int n_stages = (r1 - r0 + KB - 1) / KB;

    f32x4 acc[VA][VB];
#pragma unroll
    for (int a = 0; a < VA; ++a)
#pragma unroll
        for (int b = 0; b < VB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[PA], rb[PB];
    int rv[PA];
    auto load_stage = [&](int st) {
        const int j0 = r0 + st * KB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = tid + 256 * i, row = piece / (TM / 4), c4 = piece % (TM / 4);
            const int j = j0 + row;
            int idx = -1;
            if (j < r1) idx = p.nbr ? p.nbr[(size_t)t * p.n_out + j] : j;
            ra[i] = idx >= 0 ? *reinterpret_cast<const float4 *>(p.in + (size_t)idx * p.in_ld + ci0 + 4 * c4)
                             : float4{0.f, 0.f, 0.f, 0.f};
            rv[i] = idx >= 0;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int piece = tid + 256 * i, row = piece / (TN / 4), c4 = piece % (TN / 4);
            const int j = j0 + row;
            rb[i] = j < r1 ? *reinterpret_cast<const float4 *>(p.dy + (size_t)j * p.dy_ld + co0 + 4 * c4)
                           : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = tid + 256 * i, row = piece / (TM / 4), c4 = piece % (TM / 4);
            *reinterpret_cast<float4 *>(&sA[buf][row * LDA + 4 * c4]) = ra[i];
            if (c4 == 0) sValid[buf][row] = rv[i];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int piece = tid + 256 * i, row = piece / (TN / 4), c4 = piece % (TN / 4);
            *reinterpret_cast<float4 *>(&sB[buf][row * LDB + 4 * c4]) = rb[i];
        }
    };

    if (n_stages > 0) {
        load_stage(0);
        store_stage(0);
    }
    __syncthreads();
    for (int st = 0; st < n_stages; ++st) {
        const int buf = st & 1;
        if (st + 1 < n_stages) load_stage(st + 1);
#pragma unroll
        for (int ks = 0; ks < KB / 4; ++ks) {
            const int4 ok = reinterpret_cast<const int4 *>(sValid[buf])[ks];
            if (!(ok.x | ok.y | ok.z | ok.w)) continue;          // no row of this 4-row step has the tap
            const int row = 4 * ks + g;
            // VA consecutive channels per lane: channel = ci0 + wm*TM/2 + VA*r + a
            float a_[VA], b_[VB];
            const float *ap = &sA[buf][row * LDA + wm * (TM / 2) + VA * r];
            const float *bp = &sB[buf][row * LDB + wn * (TN / 2) + VB * r];
            if (VA == 4) { const f32x4 v = *reinterpret_cast<const f32x4 *>(ap); a_[0] = v[0]; a_[1] = v[1]; a_[VA - 2] = v[2]; a_[VA - 1] = v[3]; }
            else { const float2 v = *reinterpret_cast<const float2 *>(ap); a_[0] = v.x; a_[VA - 1] = v.y; }
            if (VB == 4) { const f32x4 v = *reinterpret_cast<const f32x4 *>(bp); b_[0] = v[0]; b_[1] = v[1]; b_[VB - 2] = v[2]; b_[VB - 1] = v[3]; }
            else { const float2 v = *reinterpret_cast<const float2 *>(bp); b_[0] = v.x; b_[VB - 1] = v.y; }
#pragma unroll
            for (int a = 0; a < VA; ++a)
#pragma unroll
                for (int b = 0; b < VB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[a], b_[b], acc[a][b], 0, 0, 0);
        }
        if (st + 1 < n_stages) store_stage(buf ^ 1);
        __syncthreads();
    }
    // D[i][j]: i = 4g+e <-> channel ci0 + wm*TM/2 + VA*i + a ; j = r <-> column co0 + wn*TN/2 + VB*r + b
    float *out = p.part + ((size_t)blockIdx.z * p.kv + t) * p.c_in * p.c_out;
#pragma unroll
    for (int a = 0; a < VA; ++a)
#pragma unroll
        for (int b = 0; b < VB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + wm * (TM / 2) + VA * (4 * g + e) + a, co = co0 + wn * (TN / 2) + VB * r + b;
                out[(size_t)ci * p.c_out + co] = acc[a][b][e];
            }
}

// Split-bf16 weight gradient (fp32-equivalent, see gather_conv.hip: x = h + m + l exactly, six bf16
// MFMA products per fp32 multiply-add). The contraction runs over ROWS, so both MFMA operands are
// "k-major": lane (i, kg) needs 8 consecutive rows of one channel. That transpose is free when a
// lane stages one CHANNEL: a wave reads 64 consecutive channels of one row per load instruction
// (256 B coalesced), a thread collects its channel's value from RA rows, splits them and writes
// them as one 16-byte k-group of the operand image [piece][k-group][channel][8 rows].
// Rows without a neighbour at the tap are dropped BEFORE staging: each 256-row block of the chunk
// is compacted (ballot + prefix) into a ring of (row, neighbour) pairs in LDS and the stages eat 32
// pairs at a time, so a sparse layer spends MFMAs on existing pairs only.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly (three bf16 terms), six products
struct WSplitBf16x3 {
    static constexpr int NP = 3;
    typedef wbf16x4 q4;
    typedef wbf16x8 q8;
    static __device__ __forceinline__ void split(const f32x4 &x, q4 (&p)[NP]) {
        p[0] = __builtin_convertvector(x, wbf16x4);
        const f32x4 r1 = x - __builtin_convertvector(p[0], f32x4);
        p[1] = __builtin_convertvector(r1, wbf16x4);
        const f32x4 r2 = r1 - __builtin_convertvector(p[1], f32x4);
        p[2] = __builtin_convertvector(r2, wbf16x4);
    }
    static __device__ __forceinline__ f32x4 mma(const q8 (&a)[NP], const q8 (&b)[NP], f32x4 c) {   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
// x = h + l (two fp16 terms: 2^-24 relative, 2^-25 absolute below 0.5), three products -- the operands are pre-scaled into
// fp16's range by a power of two (WgParams::in_absmax / dy_absmax), undone exactly on the partial sums
struct WSplitF16x2 {
    static constexpr int NP = 2;
    typedef wf16x4 q4;
    typedef wf16x8 q8;
    static __device__ __forceinline__ void split(const f32x4 &x, q4 (&p)[NP]) {
        p[0] = __builtin_convertvector(x, wf16x4);
        const f32x4 r1 = x - __builtin_convertvector(p[0], f32x4);
        p[1] = __builtin_convertvector(r1, wf16x4);
    }
    static __device__ __forceinline__ f32x4 mma(const q8 (&a)[NP], const q8 (&b)[NP], f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};

// Power-of-two pre-scale for the fp16 split (same rule as gather_conv.hip): s = 2^(14 - floor(log2 max|x|)), inv = 1 / s
__device__ __forceinline__ void in_pow2_scale(const uint32_t *absmax, float &s, float &inv) {
    s = 1.f; inv = 1.f;
    if (absmax) {
        // the maximum is kept as CPD_ABSMAX_SLOTS partial maxima, one per 128-byte line (same-line atomics serialise)
        uint32_t m = absmax[(threadIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE];
#pragma unroll
        for (int o = CPD_ABSMAX_SLOTS / 2; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
            m = t > m ? t : m;
        }
        m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        const int e = (int)((m >> 23) & 0xffu);                // biased exponent of max |in|
        if (e != 0 && e != 255) {                              // zero / denormal / inf / nan maximum: left alone
            int se = 268 - e;                                  // 127 + 14 - (e - 127)
            se = se < 1 ? 1 : (se > 253 ? 253 : se);
            s = __uint_as_float((uint32_t)se << 23);
            inv = __uint_as_float((uint32_t)(254 - se) << 23);
        }
    }
}

// R rows (4, 8 or 16) of one channel -> the piece images; `slot` = byte offset of k-group k0/8, channel ch
template <class S, int R, int IMG>
__device__ __forceinline__ void wgrad_put(char *img, int slot, int k0, const float (&v)[R], float scale) {
    if constexpr (R == 4) {
        typename S::q4 pc[S::NP];
        S::split(f32x4{v[0], v[1], v[2], v[3]} * scale, pc);
        char *dst = img + slot + ((k0 >> 2) & 1) * 8;
#pragma unroll
        for (int q = 0; q < S::NP; ++q) *reinterpret_cast<typename S::q4 *>(dst + q * IMG) = pc[q];
    } else {
#pragma unroll
        for (int q = 0; q < R / 8; ++q) {
            typename S::q4 p0[S::NP], p1[S::NP];
            S::split(f32x4{v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3]} * scale, p0);
            S::split(f32x4{v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]} * scale, p1);
            char *dst = img + slot + q * (IMG / 4);          // next k-group: T channels x 16 B further
#pragma unroll
            for (int w = 0; w < S::NP; ++w)
                *reinterpret_cast<typename S::q8 *>(dst + w * IMG) = __builtin_shufflevector(p0[w], p1[w], 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
}

template <class S, int TM, int TN>
__device__ __forceinline__ void wgrad_split_body(const WgParams &p, char *const sA, char *const sB, int *const ringJ, int *const ringI,
                                                 int *const wave_cnt) {
    constexpr int NP = S::NP;
    constexpr int KB = 32;                               // pairs per stage = K of one MFMA
    constexpr int RING = 512;                            // >= KB - 1 + 256 pairs
    constexpr int RA = TM / 8, RB = TN / 8;              // rows of a stage one thread stages per side
    constexpr int A_IMG = TM * 64, B_IMG = TN * 64;      // bytes of one piece image: 4 k-groups x T channels x 16 B
    constexpr int MS = TM / 32, NT = TN / 32;            // 2 x 2 waves, wave tile (TM/2) x (TN/2)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = (blockIdx.x / p.co_tiles) * TM, co0 = (blockIdx.x % p.co_tiles) * TN;
    const int t = blockIdx.y;
    const int r0 = blockIdx.z * p.rows_per_chunk, r1 = min(p.n_out, r0 + p.rows_per_chunk);
    const int a_ch = tid % TM, a_k0 = (tid / TM) * RA;
    const int b_ch = tid % TN, b_k0 = (tid / TN) * RB;
    const int a_slot = ((a_k0 >> 3) * TM + a_ch) << 4, b_slot = ((b_k0 >> 3) * TN + b_ch) << 4;
    const float *a_base = p.in + ci0 + a_ch;
    const float *b_base = p.dy + co0 + b_ch;
    const int32_t *nbr_t = p.nbr ? p.nbr + (size_t)t * p.n_out : nullptr;
    float sa_ = 1.f, ia_ = 1.f, sb_ = 1.f, ib_ = 1.f;
    if (NP == 2) { in_pow2_scale(p.in_absmax, sa_, ia_); in_pow2_scale(p.dy_absmax, sb_, ib_); }

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto nbr_of = [&](int j) { return j < r1 ? (nbr_t ? nbr_t[j] : j) : -1; };
    auto mma = [&]() {
        typename S::q8 a[MS][NP];
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const char *src = sA + ((g * TM + wm * (TM / 2) + 16 * s + r) << 4);
#pragma unroll
            for (int q = 0; q < NP; ++q) a[s][q] = *reinterpret_cast<const typename S::q8 *>(src + q * A_IMG);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const char *src = sB + ((g * TN + wn * (TN / 2) + 16 * nt + r) << 4);
            typename S::q8 b[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::q8 *>(src + q * B_IMG);
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                if (CPD_WG_ABLATE & 2) { asm volatile("" :: "v"(a[s][0]), "v"(a[s][1]), "v"(b[0]), "v"(b[1])); }
                else acc[s][nt] = S::mma(a[s], b, acc[s][nt]);
            }
        }
    };

    int head = 0, tail = 0, jb = r0;       // ring positions (head stays a multiple of KB) and next block of rows
    int idx_pref = nbr_of(jb + tid);
    // Two register sets: the rows of stage s + 2 are requested before stage s's MFMAs and used after stage s + 1's -- two
    // MFMA blocks of cover for the gather latency (with one set, 4 us per 32-pair stage went mostly to waiting).
    float va[2][RA], vb[2][RB];
    int cnts[2] = {0, 0};
    auto fill_and_load = [&](const int u) {
        while (tail - head < KB && jb < r1) {            // compact the next 256 rows into the ring
            const int j = jb + tid, idx = idx_pref;
            jb += 256;
            idx_pref = nbr_of(jb + tid);
            const unsigned long long bal = __ballot(idx >= 0);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int off = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c = wave_cnt[w];
                off += w < wave ? c : 0;
                total += c;
            }
            if (idx >= 0) {
                const int pos = (tail + off + __popcll(bal & ((1ull << lane) - 1ull))) & (RING - 1);
                ringJ[pos] = j;
                ringI[pos] = idx;
            }
            tail += total;
            __syncthreads();
        }
        const int cnt = min(KB, tail - head);
        cnts[u] = cnt;
        if (cnt <= 0) return;
        const int base = head & (RING - 1);
#pragma unroll
        for (int q = 0; q < RA / 4; ++q) {
            const int4 id = *reinterpret_cast<const int4 *>(&ringI[base + a_k0 + 4 * q]);
            const int ids[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                va[u][4 * q + e] = (CPD_WG_ABLATE & 1) ? (float)ids[e] : ((a_k0 + 4 * q + e < cnt) ? a_base[(size_t)ids[e] * p.in_ld] : 0.f);
        }
#pragma unroll
        for (int q = 0; q < RB / 4; ++q) {
            const int4 jd = *reinterpret_cast<const int4 *>(&ringJ[base + b_k0 + 4 * q]);
            const int js[4] = {jd.x, jd.y, jd.z, jd.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                vb[u][4 * q + e] = (CPD_WG_ABLATE & 1) ? (float)js[e] : ((b_k0 + 4 * q + e < cnt) ? b_base[(size_t)js[e] * p.dy_ld] : 0.f);
        }
        head += KB;
    };
    fill_and_load(0);
    fill_and_load(1);
    for (bool more = true; more;) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (cnts[u] <= 0) { more = false; break; }      // uniform: the chunk is used up
            if (CPD_WG_ABLATE & 4) {
#pragma unroll
                for (int e = 0; e < RA; ++e) asm volatile("" :: "v"(va[u][e]));
#pragma unroll
                for (int e = 0; e < RB; ++e) asm volatile("" :: "v"(vb[u][e]));
            } else {
                wgrad_put<S, RA, A_IMG>(sA, a_slot, a_k0, va[u], sa_);
                wgrad_put<S, RB, B_IMG>(sB, b_slot, b_k0, vb[u], sb_);
            }
            __syncthreads();
            fill_and_load(u);                                // stage s + 2 into the set just emptied
            mma();
            __syncthreads();                                 // everyone is done reading the images before they are overwritten
        }
    }

    // D[i][j]: i = 4g+e <-> channel ci0 + wm*TM/2 + 16s + i ; j = r <-> column co0 + wn*TN/2 + 16nt + j
    const float undo = ia_ * ib_;          // exact: powers of two
    float *out = p.part + ((size_t)blockIdx.z * p.kv + t) * p.c_in * p.c_out;
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + wm * (TM / 2) + 16 * s + 4 * g + e, co = co0 + wn * (TN / 2) + 16 * nt + r;
                out[(size_t)ci * p.c_out + co] = acc[s][nt][e] * undo;
            }
}

template <int TM, int TN>
__global__ void __launch_bounds__(256) wgrad_bf16_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) char sA[3 * TM * 64];
    __shared__ __attribute__((aligned(16))) char sB[3 * TN * 64];
    __shared__ __attribute__((aligned(16))) int ringJ[512];
    __shared__ __attribute__((aligned(16))) int ringI[512];
    __shared__ int wave_cnt[4];
    wgrad_split_body<WSplitBf16x3, TM, TN>(p, sA, sB, ringJ, ringI, wave_cnt);
}
template <int TM, int TN>
__global__ void __launch_bounds__(256) wgrad_f16_kernel(WgParams p) {
    __shared__ __attribute__((aligned(16))) char sA[2 * TM * 64];
    __shared__ __attribute__((aligned(16))) char sB[2 * TN * 64];
    __shared__ __attribute__((aligned(16))) int ringJ[512];
    __shared__ __attribute__((aligned(16))) int ringI[512];
    __shared__ int wave_cnt[4];
    wgrad_split_body<WSplitF16x2, TM, TN>(p, sA, sB, ringJ, ringI, wave_cnt);
}

// (same summation order per element in both forms: chunk 0, 1, 2, ...)
template <int V>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ part, int n_chunks, size_t elems,
                                                           float *__restrict__ dw, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= elems) return;
    if (V == 4) {
        f32x4 s = accumulate ? *reinterpret_cast<const f32x4 *>(dw + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < n_chunks; ++k) s += *reinterpret_cast<const f32x4 *>(part + (size_t)k * elems + i);
        *reinterpret_cast<f32x4 *>(dw + i) = s;
    } else {
        float s = accumulate ? dw[i] : 0.f;
        for (int k = 0; k < n_chunks; ++k) s += part[(size_t)k * elems + i];
        dw[i] = s;
    }
}
static void launch_wgrad_reduce(const float *part, int n_chunks, size_t elems, float *dw, int accumulate, hipStream_t s) {
    if (elems % 4 == 0 && (((uintptr_t)part | (uintptr_t)dw) & 15) == 0)
        wgrad_reduce_kernel<4><<<cpd_div_up((long long)(elems / 4), 256), 256, 0, s>>>(part, n_chunks, elems, dw, accumulate);
    else wgrad_reduce_kernel<1><<<cpd_div_up((long long)elems, 256), 256, 0, s>>>(part, n_chunks, elems, dw, accumulate);
}

typedef void (*wg_kernel_t)(WgParams);
static wg_kernel_t pick_wg(int va, int vb) {
    switch (va * 10 + vb) {
        case 11: return wgrad_kernel<1, 1>;
        case 12: return wgrad_kernel<1, 2>;
        case 14: return wgrad_kernel<1, 4>;
        case 21: return wgrad_kernel<2, 1>;
        case 22: return wgrad_kernel<2, 2>;
        case 24: return wgrad_kernel<2, 4>;
        case 41: return wgrad_kernel<4, 1>;
        case 42: return wgrad_kernel<4, 2>;
        case 44: return wgrad_kernel<4, 4>;
    }
    return nullptr;
}
static int lanes_per_16(int c) { return c >= 64 ? 4 : (c >= 32 ? 2 : 1); }

// Transposed rulebook of a strided sparse conv: nbrT[t][i] = output row fed by input i through tap t.
struct GridT { int32_t b, d, h, w; };
__device__ __forceinline__ int32_t lookup_canonical(const uint64_t *bitmap, const uint32_t *base, long long key) {
    const uint64_t w = bitmap[key >> 6];
    const uint64_t bit = 1ull << (key & 63);
    if (!(w & bit)) return -1;
    return (int32_t)(base[key >> 6] + __popcll(w & (bit - 1ull)));
}
__global__ void __launch_bounds__(256)
rulebook_transpose_kernel(const int32_t *__restrict__ in_idx, int n_in, GridT go, int kd, int kh, int kw, int sd, int sh, int sw,
                          int pd, int ph, int pw, const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                          const int32_t *__restrict__ perm_own, const int32_t *__restrict__ flags, int32_t *__restrict__ nbr_t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    // the output level's rows may have been re-ordered after its index was built (cpd_index_set_order, tap-pattern order): the
    // index then names the rank -> row map to go through, exactly as rulebook_kernel does
    const int32_t *perm = index_order(flags, perm_own);
    const int4 q = reinterpret_cast<const int4 *>(in_idx)[i];
    int t = 0;
    for (int tz = 0; tz < kd; ++tz)
        for (int ty = 0; ty < kh; ++ty)
            for (int tx = 0; tx < kw; ++tx, ++t) {
                int32_t rr = -1;
                const int nz = q.y + pd - tz, ny = q.z + ph - ty, nx = q.w + pw - tx;
                if (nz >= 0 && ny >= 0 && nx >= 0 && nz % sd == 0 && ny % sh == 0 && nx % sw == 0) {
                    const int oz = nz / sd, oy = ny / sh, ox = nx / sw;
                    if (oz < go.d && oy < go.h && ox < go.w)
                        rr = lookup_canonical(bitmap, base, (((long long)q.x * go.d + oz) * go.h + oy) * go.w + ox);
                    if (rr >= 0 && perm) rr = perm[rr];
                }
                nbr_t[(size_t)t * n_in + i] = rr;
            }
}

__global__ void __launch_bounds__(256) rulebook_conv2d_transpose_kernel(int batch, int h, int w, int ho, int wo, int kh, int kw,
                                                                        int stride, int pad, int32_t *__restrict__ nbr_t) {
    const long long n_in = (long long)batch * h * w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const int x = (int)(i % w);
    const long long t2 = i / w;
    const int y = (int)(t2 % h), b = (int)(t2 / h);
    int t = 0;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx, ++t) {
            int32_t rr = -1;
            const int ny = y + pad - ky, nx = x + pad - kx;
            if (ny >= 0 && nx >= 0 && ny % stride == 0 && nx % stride == 0) {
                const int oy = ny / stride, ox = nx / stride;
                if (oy < ho && ox < wo) rr = (int32_t)(((long long)b * ho + oy) * wo + ox);
            }
            nbr_t[(size_t)t * n_in + i] = rr;
        }
}

// Per-channel BatchNorm bookkeeping in one launch: batch mean / biased var -> invstd, the affine
// (scale, shift) that cpd_affine_rows applies, and the running-stat update with the unbiased
// variance (torch.nn.BatchNorm semantics; momentum 0.01 / eps 1e-3 in the backbones).
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float *__restrict__ sum, const float *__restrict__ sumsq, int n_rows,
                                                          int c, float eps, float momentum, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ mean,
                                                          float *__restrict__ invstd, float *__restrict__ scale,
                                                          float *__restrict__ shift, float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, const float *__restrict__ n_stat) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    const double n = n_stat ? (double)*n_stat : (double)n_rows;        // (SyncBN: the all-reduced row count, a device value)
    const double mu = (double)sum[col] / n;
    double var = (double)sumsq[col] / n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[col] = (float)mu;
    invstd[col] = is;
    const float sc = gamma[col] * is;
    scale[col] = sc;
    shift[col] = beta[col] - (float)mu * sc;
    if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * (float)mu;
        running_var[col] = (1.f - momentum) * running_var[col] + momentum * (float)unbiased;
    }
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2, float gscale,
                                                   const float *__restrict__ gscale_dev) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (gscale_dev) gscale *= *gscale_dev;              // e.g. the clip factor, computed on the device: no host read-back
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float w = p[i];
    w -= lr * wd * w;                                   // decoupled weight decay (true_wd, fastai_optim.py:132-150)
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
}

// ---------------------------------------------------------------------------------------------
// CenterHead.get_loss (center_head.py:225-250) fused with its gradient: CornerNet focal loss on the
// clamped sigmoid of the heatmap logits (loss_utils.py:265-300) + masked L1 on the regression maps
// gathered at the object pixels (loss_utils.py:315-386). Three launches, deterministic:
//   focal_partial: one pass over the B*H*W*C logits -> un-normalised d(loss)/d(logit) in d_rows and
//                  per-block partial sums (pos loss, neg loss, positives);
//   focal_final:   one block sums the partials in a fixed order (double) -> hm loss and 1/num_pos;
//   loss_finish:   scales the heatmap gradient columns, adds the regression gradients (one thread per
//                  (sample, box dimension) walks the objects in order: no atomics) and the loss parts.
// ---------------------------------------------------------------------------------------------
struct ClParams {
    const float *rows;      // head outputs [B*HW][ld]
    const float *heat;      // targets [B][C][HW]
    const float *target;    // [B][K][8]
    const long long *inds, *masks;   // [B][K]
    float *d_rows;          // [B*HW][ld]
    float *losses;          // [3] total, hm, loc
    double *part;           // [blocks][3]
    double *fin;            // [4] pos loss, neg loss, positives, scale
    int ld, batch, hw, num_classes, hm_col, k;
    float cw[8];
    float loc_weight, cls_weight;
};

__global__ void __launch_bounds__(256) focal_partial_kernel(ClParams p) {
    __shared__ double sm[3][256];
    const long long total = (long long)p.batch * p.hw * p.ld;      // one thread per (row, column) of d_rows
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double pos = 0.0, neg = 0.0, cnt = 0.0;
    if (i < total) {
        const long long row = i / p.ld;
        const int col = (int)(i - row * p.ld);
        float grad = 0.f;
        const int c = col - p.hm_col;
        if (c >= 0 && c < p.num_classes) {
            const long long b = row / p.hw, pix = row - b * p.hw;
            const float x = p.rows[i];
            const float sg = 1.0f / (1.0f + expf(-x));
            const float lo = 1e-4f, hi = 1.0f - 1e-4f;
            const float pr = fminf(fmaxf(sg, lo), hi);              // torch.clamp(sigmoid, 1e-4, 1 - 1e-4)
            const float dpdx = (sg >= lo && sg <= hi) ? sg * (1.0f - sg) : 0.f;
            const float g = p.heat[((size_t)b * p.num_classes + c) * p.hw + pix];
            if (g == 1.0f) {
                const float om = 1.0f - pr;
                pos = (double)(logf(pr) * om * om);
                cnt = 1.0;
                grad = -(om * om / pr - 2.0f * logf(pr) * om) * dpdx;           // d(-log(p)(1-p)^2)/dx
            } else if (g < 1.0f) {
                const float w1 = 1.0f - g, w = (w1 * w1) * (w1 * w1);
                const float l1 = logf(1.0f - pr);
                neg = (double)(l1 * pr * pr * w);
                grad = w * (pr * pr / (1.0f - pr) - 2.0f * pr * l1) * dpdx;     // d(-log(1-p) p^2 w)/dx
            }
        }
        p.d_rows[i] = grad;
    }
    sm[0][threadIdx.x] = pos; sm[1][threadIdx.x] = neg; sm[2][threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.part[(size_t)blockIdx.x * 3 + 0] = sm[0][0];
        p.part[(size_t)blockIdx.x * 3 + 1] = sm[1][0];
        p.part[(size_t)blockIdx.x * 3 + 2] = sm[2][0];
    }
}

__global__ void __launch_bounds__(256) focal_final_kernel(ClParams p, int n_blocks) {
    __shared__ double sm[3][256];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = threadIdx.x; k < n_blocks; k += 256) { a += p.part[(size_t)k * 3]; b += p.part[(size_t)k * 3 + 1]; c += p.part[(size_t)k * 3 + 2]; }
    sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double pos = sm[0][0], neg = sm[1][0], np = sm[2][0];
        const double scale = np > 0.0 ? 1.0 / np : 1.0;            // num_pos == 0: loss = -neg_loss
        p.fin[0] = pos; p.fin[1] = neg; p.fin[2] = np; p.fin[3] = scale;
        p.losses[1] = (float)(-(pos + neg) * scale) * p.cls_weight;
    }
}

// blocks [0, scale_blocks): scale the heatmap gradient columns; last block: regression loss + gradient
__global__ void __launch_bounds__(256) loss_finish_kernel(ClParams p, int scale_blocks) {
    if ((int)blockIdx.x < scale_blocks) {
        const long long total = (long long)p.batch * p.hw * p.num_classes;
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= total) return;
        const long long row = i / p.num_classes;
        const int c = (int)(i - row * p.num_classes);
        float *g = p.d_rows + (size_t)row * p.ld + p.hm_col + c;
        *g = *g * (float)p.fin[3] * p.cls_weight;
        return;
    }
    __shared__ double snum[256];
    __shared__ double sl1[256];
    const long long bk = (long long)p.batch * p.k;
    double num = 0.0;
    for (long long e = threadIdx.x; e < bk; e += 256) num += p.masks[e] != 0 ? 1.0 : 0.0;
    snum[threadIdx.x] = num;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) snum[threadIdx.x] += snum[threadIdx.x + s];
        __syncthreads();
    }
    const double n_obj = snum[0] < 1.0 ? 1.0 : snum[0];              // clamp_min(mask.sum(), 1)
    double l1 = 0.0;                                                 // this thread's (sample, dimension) pairs, weighted
    for (int pair = threadIdx.x; pair < p.batch * 8; pair += 256) {
        const int b = pair >> 3, d = pair & 7;
        const float gscale = p.cw[d] * p.loc_weight / (float)n_obj;
        double acc = 0.0;
        for (int k = 0; k < p.k; ++k) {
            const size_t e = (size_t)b * p.k + k;
            const float t = p.target[e * 8 + d];
            const bool t_nan = t != t;
            const float m = (p.masks[e] != 0 && !t_nan) ? 1.f : 0.f;         // mask * !isnan(target)
            if (m == 0.f && !t_nan) continue;                                // |pred*0 - t*0| = 0, gradient 0
            // as in the reference, a NaN target is NOT neutralised by its zero mask (NaN * 0 = NaN): NaN in, NaN out
            const size_t row = (size_t)b * p.hw + (size_t)p.inds[e];
            const float diff = p.rows[row * p.ld + d] * m - t * m;
            acc += (double)fabsf(diff);
            p.d_rows[row * p.ld + d] += t_nan ? diff : (diff > 0.f ? gscale : (diff < 0.f ? -gscale : 0.f));
        }
        l1 += acc * (double)p.cw[d];
    }
    sl1[threadIdx.x] = l1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sl1[threadIdx.x] += sl1[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float loc = (float)(sl1[0] / n_obj) * p.loc_weight;
        p.losses[2] = loc;
        p.losses[0] = p.losses[1] + loc;
    }
}

// ---------------------------------------------------------------------------------------------
// AnchorHeadTemplate.get_loss (anchor_head_template.py:179-334) fused with its gradient, one thread
// per (sample, anchor): sigmoid focal classification loss (alpha 0.25, gamma 2; loss_utils.py:10-75),
// smooth-L1 (beta 1/9) on the residual-coded box with the sin-difference heading encoding
// (l.219-226, 243-271; loss_utils.py:77-150) and cross entropy on the direction bins (l.228-241,
// 273-293; loss_utils.py:182-206). Normalisers are per sample: 1 / max(#positives, 1).
//   anchor_count: positives per sample (integer atomics: exact);
//   anchor_loss:  losses + gradients, per-block partial sums (double);
//   anchor_final: one block sums the partials in a fixed order -> losses[4] = total, cls, loc, dir.
// ---------------------------------------------------------------------------------------------
struct AlParams {
    const float *cls, *box, *dir;       // [B][A][C], [B][A][7], [B][A][NB] (dir may be null)
    const int32_t *labels;              // [B][A]: -1 don't care, 0 background, k > 0 class k
    const float *reg;                   // [B][A][7]
    const float *anchors;               // [A][7]
    float *d_cls, *d_box, *d_dir;
    float *losses;
    int *pos;                           // [B]
    double *part;                       // [blocks][3]
    int batch, n_anchors, num_class, num_bins;
    float dir_offset, cw[7], cls_weight, loc_weight, dir_weight;
};

__global__ void __launch_bounds__(256) anchor_count_kernel(AlParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.batch * p.n_anchors;
    const bool is_pos = i < total && p.labels[i] > 0;
    // blocks never straddle samples when n_anchors % 256 == 0; in general count per lane's own sample
    if (is_pos) atomicAdd(&p.pos[i / p.n_anchors], 1);
}

__global__ void __launch_bounds__(256) anchor_loss_kernel(AlParams p) {
    __shared__ double sm[3][256];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.batch * p.n_anchors;
    double l_cls = 0.0, l_loc = 0.0, l_dir = 0.0;
    if (i < total) {
        const int b = (int)(i / p.n_anchors);
        const int a = (int)(i - (long long)b * p.n_anchors);
        const int label = p.labels[i];
        const float norm = 1.0f / fmaxf((float)p.pos[b], 1.0f);
        const float inv_b = 1.0f / (float)p.batch;
        // ---- classification: weight 1/norm for background and positives, 0 for don't-care
        const float w_cls = label >= 0 ? norm : 0.f;
        for (int c = 0; c < p.num_class; ++c) {
            const float x = p.cls[i * p.num_class + c];
            const float t = (label == c + 1) ? 1.f : 0.f;
            const float pr = 1.0f / (1.0f + expf(-x));
            const float alpha = t * 0.25f + (1.f - t) * 0.75f;
            const float pt = t * (1.f - pr) + (1.f - t) * pr;
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            const float fw = alpha * pt * pt;
            l_cls += (double)(fw * bce * w_cls);
            const float dpt = (t > 0.5f ? -1.f : 1.f) * pr * (1.f - pr);
            const float g = alpha * 2.f * pt * dpt * bce + fw * (pr - t);
            p.d_cls[i * p.num_class + c] = g * w_cls * p.cls_weight * inv_b;
        }
        // ---- box regression + direction: positives only
        const float w_reg = label > 0 ? norm : 0.f;
        float gb[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float beta = 1.0f / 9.0f;
        if (w_reg > 0.f) {
            const float *bp = p.box + i * 7, *tp = p.reg + i * 7;
            for (int k = 0; k < 7; ++k) {
                float pv = bp[k], tv = tp[k], dd = 1.f;             // dd = d(encoded diff)/d(pred)
                if (k == 6) {
                    const float sp = sinf(pv), cp = cosf(pv), st = sinf(tv), ct = cosf(tv);
                    pv = sp * ct; tv = cp * st;                         // sin(a - b) = sin a cos b - cos a sin b
                    dd = cp * ct + sp * st;
                }
                float diff = (tv != tv) ? 0.f : (pv - tv);              // NaN target -> replaced by the prediction
                if (tv != tv) dd = 0.f;
                diff *= p.cw[k];
                const float n = fabsf(diff);
                l_loc += (double)((n < beta ? 0.5f * n * n / beta : n - 0.5f * beta) * w_reg);
                const float ds = n < beta ? diff / beta : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
                gb[k] = ds * p.cw[k] * dd * w_reg * p.loc_weight * inv_b;
            }
        }
        for (int k = 0; k < 7; ++k) p.d_box[i * 7 + k] = gb[k];
        if (p.dir) {
            const int nb = p.num_bins;
            const float *dp = p.dir + i * nb;
            float *gd = p.d_dir + i * nb;
            if (w_reg > 0.f) {
                const float two_pi = 6.283185307179586f;
                const float rot = p.reg[i * 7 + 6] + p.anchors[(size_t)a * 7 + 6];
                const float v = rot - p.dir_offset;
                const float off = v - floorf(v / two_pi + 0.f) * two_pi;       // limit_period(v, 0, 2 pi)
                int bin = (int)floorf(off / (two_pi / (float)nb));
                bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
                float mx = dp[0];
                for (int k = 1; k < nb; ++k) mx = fmaxf(mx, dp[k]);
                float se = 0.f;
                for (int k = 0; k < nb; ++k) se += expf(dp[k] - mx);
                const float lse = logf(se) + mx;
                l_dir += (double)((lse - dp[bin]) * w_reg);
                for (int k = 0; k < nb; ++k)
                    gd[k] = (expf(dp[k] - lse) - (k == bin ? 1.f : 0.f)) * w_reg * p.dir_weight * inv_b;
            } else {
                for (int k = 0; k < nb; ++k) gd[k] = 0.f;
            }
        }
    }
    sm[0][threadIdx.x] = l_cls; sm[1][threadIdx.x] = l_loc; sm[2][threadIdx.x] = l_dir;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.part[(size_t)blockIdx.x * 3 + 0] = sm[0][0];
        p.part[(size_t)blockIdx.x * 3 + 1] = sm[1][0];
        p.part[(size_t)blockIdx.x * 3 + 2] = sm[2][0];
    }
}

__global__ void __launch_bounds__(256) anchor_final_kernel(AlParams p, int n_blocks) {
    __shared__ double sm[3][256];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = threadIdx.x; k < n_blocks; k += 256) { a += p.part[(size_t)k * 3]; b += p.part[(size_t)k * 3 + 1]; c += p.part[(size_t)k * 3 + 2]; }
    sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b; sm[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float lc = (float)(sm[0][0] / p.batch) * p.cls_weight;
        const float ll = (float)(sm[1][0] / p.batch) * p.loc_weight;
        const float ld = p.dir ? (float)(sm[2][0] / p.batch) * p.dir_weight : 0.f;
        p.losses[1] = lc; p.losses[2] = ll; p.losses[3] = ld;
        p.losses[0] = lc + ll + ld;
    }
}

}  // namespace

enum { COL_MAX_CHUNKS = 512 };

static int col_reduce(int mode, const float *a, int lda, const float *y, int ldy, const float *x, int ldx, const float *mean,
                      const float *invstd, int n, int c, float *s1, float *s2, void *ws, size_t ws_bytes, hipStream_t s,
                      const BnFin *fin = nullptr) {
    if (!a || (!s1 && !fin) || n <= 0 || c <= 0 || !ws) return CPD_ERR_ARG;
    auto al16 = [](const void *q, int ld) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && (ld & 3) == 0); };
    const bool vec = (c % 4 == 0) && al16(a, lda) && al16(y, ldy) && al16(x, ldx);
    const int V = vec ? 4 : 1;
    const int cq_total = (c + V - 1) / V;
    const int cq = cq_total < 16 ? cq_total : 16;
    const int rl = 256 / cq;
    const int col_tiles = (cq_total + cq - 1) / cq;
    // partial workgroups to aim for: one per CU. Measured on the train step (one frame per GPU, 97 reductions per step):
    // 2048 workgroups 10.85 ms/step, 1024 10.82, 512 10.77, 256 10.66, 128 10.58, 64 11.07, 32 12.2 -- the final sum over the
    // chunks (a dependent second launch) is the part that shrinks. (Folding that sum into this kernel -- last-workgroup-done
    // tickets -- was measured twice, at 512 and at 128 chunks per tile: 16.9 / 12.5 ms per step. Not done.)
    int col_blocks = 256;
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_COL_BLOCKS")) col_blocks = atoi(e);
    int chunks = col_blocks / col_tiles;
    const int max_chunks = (n + 4 * rl - 1) / (4 * rl);           // at least four rows per thread
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > COL_MAX_CHUNKS) chunks = COL_MAX_CHUNKS;
    if (chunks < 1) chunks = 1;
    int rpc = (n + chunks - 1) / chunks;
    rpc = (rpc + rl - 1) / rl * rl;
    chunks = (n + rpc - 1) / rpc;
    if (ws_bytes < (size_t)chunks * 2 * c * sizeof(float)) return CPD_ERR_WORKSPACE;
    ColParams p{a, y, x, mean, invstd, lda, ldy, ldx, n, c, rpc, cq};
    float *part = (float *)ws;
    const dim3 grid(col_tiles, chunks);
    if (vec) {
        if (mode == COL_SUM) col_partial_kernel<COL_SUM, 4><<<grid, 256, 0, s>>>(p, part);
        else if (mode == COL_STATS) col_partial_kernel<COL_STATS, 4><<<grid, 256, 0, s>>>(p, part);
        else col_partial_kernel<COL_BNBWD, 4><<<grid, 256, 0, s>>>(p, part);
    } else {
        if (mode == COL_SUM) col_partial_kernel<COL_SUM, 1><<<grid, 256, 0, s>>>(p, part);
        else if (mode == COL_STATS) col_partial_kernel<COL_STATS, 1><<<grid, 256, 0, s>>>(p, part);
        else col_partial_kernel<COL_BNBWD, 1><<<grid, 256, 0, s>>>(p, part);
    }
    if (fin) col_final_kernel<true><<<cpd_div_up(c, 16), 256, 0, s>>>(part, chunks, c, nullptr, nullptr, *fin);
    else col_final_kernel<false><<<cpd_div_up(c, 16), 256, 0, s>>>(part, chunks, c, s1, s2, BnFin{});
    return cpd_check_launch();
}

extern "C" size_t cpd_col_reduce_workspace_bytes(int n, int c) {
    if (n <= 0 || c <= 0) return 0;
    return cpd_align((size_t)COL_MAX_CHUNKS * 2 * c * sizeof(float));
}
extern "C" int cpd_col_sum(const float *x, int ldx, int n, int c, float *sum, void *ws, size_t ws_bytes, cpd_stream_t st) {
    return col_reduce(COL_SUM, x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, n, c, sum, nullptr, ws, ws_bytes, cpd_s(st));
}
extern "C" int cpd_bn_stats(const float *x, int ldx, int n, int c, float *sum, float *sumsq, void *ws, size_t ws_bytes,
                            cpd_stream_t st) {
    if (!sumsq) return CPD_ERR_ARG;
    return col_reduce(COL_STATS, x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, n, c, sum, sumsq, ws, ws_bytes, cpd_s(st));
}
extern "C" int cpd_bn_bwd_reduce(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, const float *mean,
                                 const float *invstd, int n, int c, float *dbeta, float *dgamma, void *ws, size_t ws_bytes,
                                 cpd_stream_t st) {
    if (!x || !mean || !invstd || !dgamma) return CPD_ERR_ARG;
    return col_reduce(COL_BNBWD, dy, lddy, y, ldy, x, ldx, mean, invstd, n, c, dbeta, dgamma, ws, ws_bytes, cpd_s(st));
}
extern "C" int cpd_bn_stats_finalize(const float *x, int ldx, int n, int c, float eps, float momentum, const float *gamma,
                                     const float *beta, float *mean, float *invstd, float *scale, float *shift,
                                     float *running_mean, float *running_var, void *ws, size_t ws_bytes, cpd_stream_t st) {
    if (!gamma || !beta || !mean || !invstd || !scale || !shift || (running_mean && !running_var)) return CPD_ERR_ARG;
    BnFin f{n, eps, momentum, gamma, beta, mean, invstd, scale, shift, running_mean, running_var};
    return col_reduce(COL_STATS, x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, n, c, nullptr, nullptr, ws, ws_bytes, cpd_s(st),
                      &f);
}
extern "C" int cpd_bn_finalize(const float *sum, const float *sumsq, int n, int c, float eps, float momentum, const float *gamma,
                               const float *beta, float *mean, float *invstd, float *scale, float *shift,
                               float *running_mean, float *running_var, cpd_stream_t st) {
    if (!sum || !sumsq || !gamma || !beta || !mean || !invstd || !scale || !shift || n <= 0 || c <= 0 ||
        (running_mean && !running_var))
        return CPD_ERR_ARG;
    bn_finalize_kernel<<<cpd_div_up(c, 256), 256, 0, cpd_s(st)>>>(sum, sumsq, n, c, eps, momentum, gamma, beta, mean, invstd, scale,
                                                                 shift, running_mean, running_var, nullptr);
    return cpd_check_launch();
}
extern "C" int cpd_bn_finalize_sync(const float *sum, const float *sumsq, const float *n_total, int c, float eps, float momentum,
                                    const float *gamma, const float *beta, float *mean, float *invstd, float *scale, float *shift,
                                    float *running_mean, float *running_var, cpd_stream_t st) {
    if (!sum || !sumsq || !n_total || !gamma || !beta || !mean || !invstd || !scale || !shift || c <= 0 || (running_mean && !running_var))
        return CPD_ERR_ARG;
    bn_finalize_kernel<<<cpd_div_up(c, 256), 256, 0, cpd_s(st)>>>(sum, sumsq, 0, c, eps, momentum, gamma, beta, mean, invstd, scale,
                                                                 shift, running_mean, running_var, n_total);
    return cpd_check_launch();
}
extern "C" int cpd_affine_rows(const float *x, int ldx, int n, int c, const float *scale, const float *shift,
                               const float *residual, int ldr, int relu, float *out, int ldo, cpd_stream_t st) {
    if (!x || !out || n < 0 || c <= 0) return CPD_ERR_ARG;
    if (n == 0) return CPD_OK;
    const bool vec = c % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!residual || ldr % 4 == 0) &&
                     ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)residual)) & 15) == 0;
    if (vec) affine_rows_kernel<4><<<cpd_div_up((long long)n * (c / 4), 256), 256, 0, cpd_s(st)>>>(x, ldx, n, c, scale, shift, residual, ldr,
                                                                                                    relu, out, ldo);
    else affine_rows_kernel<1><<<cpd_div_up((long long)n * c, 256), 256, 0, cpd_s(st)>>>(x, ldx, n, c, scale, shift, residual, ldr, relu,
                                                                                        out, ldo);
    return cpd_check_launch();
}
static int bn_bwd_apply_impl(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, int n, int c,
                             const float *mean, const float *invstd, const float *gamma, const float *dbeta,
                             const float *dgamma, float *dx, int lddx, float *dres, int lddres, uint32_t *dx_absmax,
                             const float *n_stat, cpd_stream_t st);
extern "C" int cpd_bn_bwd_apply(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, int n, int c,
                                const float *mean, const float *invstd, const float *gamma, const float *dbeta,
                                const float *dgamma, float *dx, int lddx, float *dres, int lddres, uint32_t *dx_absmax,
                                cpd_stream_t st) {
    return bn_bwd_apply_impl(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, gamma, dbeta, dgamma, dx, lddx, dres, lddres, dx_absmax, nullptr, st);
}
extern "C" int cpd_bn_bwd_apply_sync(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, int n, int c,
                                     const float *mean, const float *invstd, const float *gamma, const float *sum_dy,
                                     const float *sum_dy_xhat, const float *n_total, float *dx, int lddx, float *dres, int lddres,
                                     uint32_t *dx_absmax, cpd_stream_t st) {
    if (!n_total) return CPD_ERR_ARG;
    return bn_bwd_apply_impl(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, gamma, sum_dy, sum_dy_xhat, dx, lddx, dres, lddres, dx_absmax, n_total, st);
}
static int bn_bwd_apply_impl(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, int n, int c,
                             const float *mean, const float *invstd, const float *gamma, const float *dbeta,
                             const float *dgamma, float *dx, int lddx, float *dres, int lddres, uint32_t *dx_absmax,
                             const float *n_stat, cpd_stream_t st) {
    if (!dy || !x || !mean || !invstd || !gamma || !dbeta || !dgamma || !dx || n <= 0 || c <= 0) return CPD_ERR_ARG;
    const bool vec = c % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!y || ldy % 4 == 0) && (!dres || lddres % 4 == 0) &&
                     ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)dres)) & 15) == 0;
    const long long items = (long long)n * (vec ? c / 4 : c);
    const long long nblk = (items + 255) / 256;
    if (nblk >= (1ll << 31)) return CPD_ERR_UNSUPPORTED;
    const unsigned blocks = (unsigned)nblk;
    if (vec) bn_bwd_apply_kernel<4><<<blocks, 256, 0, cpd_s(st)>>>(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, gamma, dbeta, dgamma, dx, lddx,
                                                                  dres, lddres, dx_absmax, n_stat);
    else bn_bwd_apply_kernel<1><<<blocks, 256, 0, cpd_s(st)>>>(dy, lddy, y, ldy, x, ldx, n, c, mean, invstd, gamma, dbeta, dgamma, dx, lddx,
                                                              dres, lddres, dx_absmax, n_stat);
    return cpd_check_launch();
}
extern "C" int cpd_relu_bwd(const float *dy, int lddy, const float *y, int ldy, int n, int c, float *dx, int lddx,
                            cpd_stream_t st) {
    if (!dy || !y || !dx || n < 0 || c <= 0) return CPD_ERR_ARG;
    if (n == 0) return CPD_OK;
    relu_bwd_kernel<<<cpd_div_up((long long)n * c, 256), 256, 0, cpd_s(st)>>>(dy, lddy, y, ldy, n, c, dx, lddx);
    return cpd_check_launch();
}

// Tile kernel plan: (TM, TN) in {64,128}^2 when both channel counts are multiples of 64.
static void wgrad_chunks(int n_out, long long per_chunk, int min_rows, int round, WgParams *p, long long target = 512) {
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_WGRAD_WGS")) target = atoll(e);
    int chunks = (int)((target + per_chunk - 1) / per_chunk);
    const int max_chunks = (n_out + min_rows - 1) / min_rows;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    p->rows_per_chunk = ((n_out + chunks - 1) / chunks + round - 1) / round * round;
    p->n_chunks = (n_out + p->rows_per_chunk - 1) / p->rows_per_chunk;
}

static bool wgrad_tile_plan(int n_out, int c_in, int c_out, int kv, WgParams *p, int *tm, int *tn) {
    if (c_in % 64 || c_out % 64) return false;
    *tm = c_in % 128 ? 64 : 128;
    *tn = c_out % 128 ? 64 : 128;
    p->ci_tiles = c_in / *tm;
    p->co_tiles = c_out / *tn;
    wgrad_chunks(n_out, (long long)kv * p->ci_tiles * p->co_tiles, 128, 16, p);
    return true;
}

// split-bf16 kernel: whole 32-channel tiles on both sides
static int wgrad_bf16_tile(int c) { return c % 128 == 0 ? 128 : (c % 64 == 0 ? 64 : 32); }
static bool wgrad_bf16_plan(int n_out, int c_in, int c_out, int kv, int flags, WgParams *p, int *tm, int *tn) {
    int on = (flags & (2 | 4)) != 0;
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_WGRAD_BF16X3")) on = atoi(e);
    if (!on || c_in % 32 || c_out % 32) return false;
    *tm = wgrad_bf16_tile(c_in);
    *tn = wgrad_bf16_tile(c_out);
    p->ci_tiles = c_in / *tm;
    p->co_tiles = c_out / *tn;
    // workgroups to aim for: 2 per CU for the 128 x 128 tile (its partial sums are the largest), 4 per CU for the smaller tiles
    wgrad_chunks(n_out, (long long)kv * p->ci_tiles * p->co_tiles, 256, 256, p, (*tm) * (*tn) >= 128 * 128 ? 512 : 1024);
    return true;
}

static void wgrad_plan(int n_out, int c_in, int c_out, int kv, WgParams *p, int *va, int *vb) {
    int tm, tn;
    *va = *vb = 0;
    if (wgrad_tile_plan(n_out, c_in, c_out, kv, p, &tm, &tn)) return;
    *va = lanes_per_16(c_in); *vb = lanes_per_16(c_out);
    p->ci_tiles = (c_in + 16 * *va - 1) / (16 * *va);
    p->co_tiles = (c_out + 16 * *vb - 1) / (16 * *vb);
    // enough waves to fill the chip, few enough chunks to keep the partial buffer small
    long long per_chunk = (long long)kv * p->ci_tiles * p->co_tiles;
    int chunks = (int)((8192 + per_chunk - 1) / per_chunk);
    int max_chunks = (n_out + 255) / 256;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    p->rows_per_chunk = ((n_out + chunks - 1) / chunks + 63) / 64 * 64;
    p->n_chunks = (n_out + p->rows_per_chunk - 1) / p->rows_per_chunk;
}
// the workspace holds the row-chunk partials of whichever kernel the flags select: size it for the larger plan
extern "C" size_t cpd_conv_wgrad_workspace_bytes(int n_out, int c_in, int c_out, int kv) {
    if (n_out <= 0 || c_in <= 0 || c_out <= 0 || kv <= 0) return 0;
    WgParams p; int va, vb, tm, tn;
    wgrad_plan(n_out, c_in, c_out, kv, &p, &va, &vb);
    int chunks = p.n_chunks;
    if (wgrad_bf16_plan(n_out, c_in, c_out, kv, 2, &p, &tm, &tn) && p.n_chunks > chunks) chunks = p.n_chunks;
    return cpd_align((size_t)chunks * kv * c_in * c_out * sizeof(float));
}

template <int TM>
static void launch_wgrad_bf16(int tn, dim3 grid, hipStream_t s, const WgParams &p) {
    if (tn == 128) wgrad_bf16_kernel<TM, 128><<<grid, 256, 0, s>>>(p);
    else if (tn == 64) wgrad_bf16_kernel<TM, 64><<<grid, 256, 0, s>>>(p);
    else wgrad_bf16_kernel<TM, 32><<<grid, 256, 0, s>>>(p);
}
template <int TM>
static void launch_wgrad_f16(int tn, dim3 grid, hipStream_t s, const WgParams &p) {
    if (tn == 128) wgrad_f16_kernel<TM, 128><<<grid, 256, 0, s>>>(p);
    else if (tn == 64) wgrad_f16_kernel<TM, 64><<<grid, 256, 0, s>>>(p);
    else wgrad_f16_kernel<TM, 32><<<grid, 256, 0, s>>>(p);
}

static int conv_wgrad_impl(const float *in, int in_ld, int c_in, const float *dy, int dy_ld, int c_out, const int32_t *nbr,
                           int kv, int n_out, float *dw_kio, int flags, const uint32_t *in_absmax, const uint32_t *dy_absmax,
                           void *ws, size_t ws_bytes, cpd_stream_t st) {
    if (!in || !dy || !dw_kio || !ws || n_out <= 0 || c_in <= 0 || c_out <= 0 || kv <= 0 || (!nbr && kv != 1)) return CPD_ERR_ARG;
    const int accumulate = flags & 1;
    WgParams p;
    int va, vb, tm, tn;
    p.in = in; p.dy = dy; p.nbr = nbr; p.part = (float *)ws; p.in_absmax = in_absmax; p.dy_absmax = dy_absmax;
    p.in_ld = in_ld; p.dy_ld = dy_ld; p.c_in = c_in; p.c_out = c_out; p.kv = kv; p.n_out = n_out;
    const size_t elems = (size_t)kv * c_in * c_out;
    if (wgrad_bf16_plan(n_out, c_in, c_out, kv, flags, &p, &tm, &tn)) {
        if (ws_bytes < (size_t)p.n_chunks * elems * sizeof(float)) return CPD_ERR_WORKSPACE;
        if (p.n_chunks >= 65536 || kv >= 65536) return CPD_ERR_UNSUPPORTED;
        const dim3 grid(p.ci_tiles * p.co_tiles, kv, p.n_chunks);
        {
            char nm[64];
            snprintf(nm, sizeof nm, "wgrad_%s_kernel<%d,%d>", (flags & 4) ? "f16" : "bf16", tm, tn);
            cpd_launch_log_note(nm);
        }
        if (flags & 4) {
            if (tm == 128) launch_wgrad_f16<128>(tn, grid, cpd_s(st), p);
            else if (tm == 64) launch_wgrad_f16<64>(tn, grid, cpd_s(st), p);
            else launch_wgrad_f16<32>(tn, grid, cpd_s(st), p);
        } else if (tm == 128) launch_wgrad_bf16<128>(tn, grid, cpd_s(st), p);
        else if (tm == 64) launch_wgrad_bf16<64>(tn, grid, cpd_s(st), p);
        else launch_wgrad_bf16<32>(tn, grid, cpd_s(st), p);
        launch_wgrad_reduce(p.part, p.n_chunks, elems, dw_kio, accumulate, cpd_s(st));
        return cpd_check_launch();
    }
    wgrad_plan(n_out, c_in, c_out, kv, &p, &va, &vb);
    if (ws_bytes < (size_t)p.n_chunks * elems * sizeof(float)) return CPD_ERR_WORKSPACE;
    const bool aligned = ((((uintptr_t)in) | ((uintptr_t)dy)) & 15) == 0 && in_ld % 4 == 0 && dy_ld % 4 == 0;
    if (va == 0 && !aligned) {                        // tile kernel needs 16-byte rows: fall back to the wave kernel
        va = lanes_per_16(c_in); vb = lanes_per_16(c_out);
        p.ci_tiles = (c_in + 16 * va - 1) / (16 * va);
        p.co_tiles = (c_out + 16 * vb - 1) / (16 * vb);
    }
    if (va == 0) {
        wgrad_tile_plan(n_out, c_in, c_out, kv, &p, &tm, &tn);
        if (p.n_chunks >= 65536 || kv >= 65536) return CPD_ERR_UNSUPPORTED;
        const dim3 grid(p.ci_tiles * p.co_tiles, kv, p.n_chunks);
        { char nm[64]; snprintf(nm, sizeof nm, "wgrad_tile_kernel<%d,%d>", tm, tn); cpd_launch_log_note(nm); }
        if (tm == 128 && tn == 128) wgrad_tile_kernel<128, 128><<<grid, 256, 0, cpd_s(st)>>>(p);
        else if (tm == 128) wgrad_tile_kernel<128, 64><<<grid, 256, 0, cpd_s(st)>>>(p);
        else if (tn == 128) wgrad_tile_kernel<64, 128><<<grid, 256, 0, cpd_s(st)>>>(p);
        else wgrad_tile_kernel<64, 64><<<grid, 256, 0, cpd_s(st)>>>(p);
    } else {
        wg_kernel_t k = pick_wg(va, vb);
        if (!k) return CPD_ERR_UNSUPPORTED;
        { char nm[64]; snprintf(nm, sizeof nm, "wgrad_kernel<%d,%d>", va, vb); cpd_launch_log_note(nm); }
        const long long blocks = (long long)p.n_chunks * kv * p.ci_tiles * p.co_tiles;
        if (blocks >= (1ll << 31)) return CPD_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), 0, cpd_s(st), p);
    }
    launch_wgrad_reduce(p.part, p.n_chunks, elems, dw_kio, accumulate, cpd_s(st));
    return cpd_check_launch();
}

extern "C" int cpd_conv_wgrad(const float *in, int in_ld, int c_in, const float *dy, int dy_ld, int c_out, const int32_t *nbr,
                              int kv, int n_out, float *dw_kio, int flags, void *ws, size_t ws_bytes, cpd_stream_t st) {
    return conv_wgrad_impl(in, in_ld, c_in, dy, dy_ld, c_out, nbr, kv, n_out, dw_kio, flags, nullptr, nullptr, ws, ws_bytes, st);
}
extern "C" int cpd_conv_wgrad_scaled(const float *in, int in_ld, int c_in, const float *dy, int dy_ld, int c_out, const int32_t *nbr,
                                     int kv, int n_out, float *dw_kio, int flags, const uint32_t *in_absmax, const uint32_t *dy_absmax,
                                     void *ws, size_t ws_bytes, cpd_stream_t st) {
    return conv_wgrad_impl(in, in_ld, c_in, dy, dy_ld, c_out, nbr, kv, n_out, dw_kio, flags, in_absmax, dy_absmax, ws, ws_bytes, st);
}

extern "C" int cpd_rulebook_conv_transpose(const int32_t *in_indices, int n_in, int batch, const int32_t in_shape[3],
                                           const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                                           const void *out_index, int32_t *nbr_t, cpd_stream_t st) {
    int32_t os[3];
    if (!in_indices || !out_index || !nbr_t || n_in <= 0 || batch <= 0) return CPD_ERR_ARG;
    int rc = cpd_conv_out_shape(in_shape, ksize, stride, pad, os);
    if (rc) return rc;
    if (ksize[2] > 32) return CPD_ERR_UNSUPPORTED;
    // the output index (cpd_conv_outset / cpd_index_build): flags | bitmap | base | spine | own rank -> row map; which map applies
    // (none, its own, a caller-owned one) is read from the flags on the device
    IndexView v = index_carve(const_cast<void *>(out_index), batch, os, 1);
    const uint64_t *bitmap = v.bitmap;
    const uint32_t *base = v.base;
    GridT go{batch, os[0], os[1], os[2]};
    rulebook_transpose_kernel<<<cpd_div_up(n_in, 256), 256, 0, cpd_s(st)>>>(in_indices, n_in, go, ksize[0], ksize[1], ksize[2],
                                                                           stride[0], stride[1], stride[2], pad[0], pad[1],
                                                                           pad[2], bitmap, base, v.perm, v.flags, nbr_t);
    return cpd_check_launch();
}

extern "C" int cpd_rulebook_conv2d_transpose(int batch, int h, int w, int kh, int kw, int stride, int pad, int32_t *nbr_t,
                                             cpd_stream_t st) {
    if (batch <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || !nbr_t) return CPD_ERR_ARG;
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    const long long n_in = (long long)batch * h * w;
    rulebook_conv2d_transpose_kernel<<<cpd_div_up(n_in, 256), 256, 0, cpd_s(st)>>>(batch, h, w, ho, wo, kh, kw, stride, pad, nbr_t);
    return cpd_check_launch();
}

extern "C" int cpd_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, const float *grad_scale_dev,
                             cpd_stream_t st) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step <= 0) return CPD_ERR_ARG;
    if (n == 0) return CPD_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    adam_kernel<<<cpd_div_up((long long)n, 256), 256, 0, cpd_s(st)>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                                     weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    return cpd_check_launch();
}

// ---- fused CenterHead loss + gradient ----
static int center_loss_blocks(long long n_rows, int ld) { return (int)cpd_div_up(n_rows * ld, 256); }
extern "C" size_t cpd_center_loss_workspace_bytes(int batch, int hw, int ld) {
    if (batch <= 0 || hw <= 0 || ld <= 0) return 0;
    return cpd_align(((size_t)center_loss_blocks((long long)batch * hw, ld) * 3 + 4) * sizeof(double));
}
extern "C" int cpd_center_loss(const float *rows, int ld, int batch, int hw, int num_classes, int hm_col, const float *heat,
                               const float *target, const int64_t *inds, const int64_t *masks, int k, const float code_weights[8],
                               float loc_weight, float cls_weight, float *d_rows, float *losses, void *ws, size_t ws_bytes,
                               cpd_stream_t st) {
    if (!rows || !heat || !target || !inds || !masks || !code_weights || !d_rows || !losses || !ws || batch <= 0 || hw <= 0 ||
        num_classes <= 0 || hm_col < 8 || hm_col + num_classes > ld || k < 0)
        return CPD_ERR_ARG;
    const long long n_rows = (long long)batch * hw;
    if (n_rows * ld >= (1ll << 40)) return CPD_ERR_UNSUPPORTED;
    const int blocks = center_loss_blocks(n_rows, ld);
    if (ws_bytes < ((size_t)blocks * 3 + 4) * sizeof(double)) return CPD_ERR_WORKSPACE;
    ClParams p;
    p.rows = rows; p.heat = heat; p.target = target; p.inds = (const long long *)inds; p.masks = (const long long *)masks;
    p.d_rows = d_rows; p.losses = losses; p.part = (double *)ws; p.fin = (double *)ws + (size_t)blocks * 3;
    p.ld = ld; p.batch = batch; p.hw = hw; p.num_classes = num_classes; p.hm_col = hm_col; p.k = k;
    for (int d = 0; d < 8; ++d) p.cw[d] = code_weights[d];
    p.loc_weight = loc_weight; p.cls_weight = cls_weight;
    focal_partial_kernel<<<blocks, 256, 0, cpd_s(st)>>>(p);
    focal_final_kernel<<<1, 256, 0, cpd_s(st)>>>(p, blocks);
    const int scale_blocks = (int)cpd_div_up(n_rows * num_classes, 256);
    loss_finish_kernel<<<scale_blocks + 1, 256, 0, cpd_s(st)>>>(p, scale_blocks);
    return cpd_check_launch();
}

// ---- fused anchor-head loss + gradient ----
extern "C" size_t cpd_anchor_loss_workspace_bytes(int batch, int n_anchors) {
    if (batch <= 0 || n_anchors <= 0) return 0;
    const size_t blocks = (size_t)cpd_div_up((long long)batch * n_anchors, 256);
    return cpd_align(blocks * 3 * sizeof(double)) + cpd_align((size_t)batch * sizeof(int));
}
extern "C" int cpd_anchor_loss(const float *cls_preds, const float *box_preds, const float *dir_preds, const int32_t *labels,
                               const float *reg_targets, const float *anchors, int batch, int n_anchors, int num_class, int num_dir_bins,
                               float dir_offset, const float code_weights[7], float cls_weight, float loc_weight, float dir_weight,
                               float *d_cls, float *d_box, float *d_dir, float *losses, void *ws, size_t ws_bytes, cpd_stream_t st) {
    if (!cls_preds || !box_preds || !labels || !reg_targets || !anchors || !code_weights || !d_cls || !d_box || !losses || !ws ||
        batch <= 0 || n_anchors <= 0 || num_class <= 0 || (dir_preds && (!d_dir || num_dir_bins < 2 || num_dir_bins > 16)))
        return CPD_ERR_ARG;
    if (ws_bytes < cpd_anchor_loss_workspace_bytes(batch, n_anchors)) return CPD_ERR_WORKSPACE;
    const long long total = (long long)batch * n_anchors;
    const int blocks = (int)cpd_div_up(total, 256);
    AlParams p;
    p.cls = cls_preds; p.box = box_preds; p.dir = dir_preds; p.labels = labels; p.reg = reg_targets; p.anchors = anchors;
    p.d_cls = d_cls; p.d_box = d_box; p.d_dir = d_dir; p.losses = losses;
    p.part = (double *)ws;
    p.pos = (int *)((char *)ws + cpd_align((size_t)blocks * 3 * sizeof(double)));
    p.batch = batch; p.n_anchors = n_anchors; p.num_class = num_class; p.num_bins = num_dir_bins;
    p.dir_offset = dir_offset;
    for (int k = 0; k < 7; ++k) p.cw[k] = code_weights[k];
    p.cls_weight = cls_weight; p.loc_weight = loc_weight; p.dir_weight = dir_weight;
    if (hipMemsetAsync(p.pos, 0, (size_t)batch * sizeof(int), cpd_s(st)) != hipSuccess) return CPD_ERR_LAUNCH;
    anchor_count_kernel<<<blocks, 256, 0, cpd_s(st)>>>(p);
    anchor_loss_kernel<<<blocks, 256, 0, cpd_s(st)>>>(p);
    anchor_final_kernel<<<1, 256, 0, cpd_s(st)>>>(p, blocks);
    return cpd_check_launch();
}
