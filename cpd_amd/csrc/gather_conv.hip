// gather_conv.hip -- the one GEMM-shaped kernel of the path: rulebook-driven implicit GEMM on the
// gfx950 fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, 157 TFLOP/s peak).
//
//   out[j, :] = act( (sum_t in[nbr[t][j], :] . W[t]) * scale + shift + residual[j, :] )
//
// It serves both halves of the reference's compute:
//   * [SPCONV] SubMConv3d / SparseConv3d forward (spconv_backbone.py:17-21,108-115,414-455) with the
//     rulebook of site_index.hip, bias + eval BatchNorm1d + ReLU + SparseBasicBlock residual fused
//     into the epilogue (spconv_backbone.py:29-33,120-136);
//   * every Conv2d / ConvTranspose2d (+BatchNorm2d+ReLU) of BaseBEVBackbone
//     (base_bev_backbone.py:31-59) and CenterHead (center_head.py:21-27,73-80) on channels-last
//     maps, with a dense pixel rulebook (cpd_rulebook_conv2d) -- cuDNN's role in the reference.
//
// Wavefront-segmented: one wave64 owns a (16*MS rows) x (16*NT cols) output tile, keeps it in
// accumulator registers across all taps and all input channels, and writes it once. No LDS, no
// barriers: A rows are gathered straight from L2 as 16-byte pieces in MFMA operand order (lane
// (r,g) holds channels 4g..4g+3 of row r), B fragments are pre-packed so each MFMA's B operand is
// one fully coalesced 256-byte read shared by all MS row sub-tiles. Row sub-tiles (16 rows) with no
// active neighbour at a tap skip that tap's MFMAs (wave-uniform branch on a ballot) -- that is what
// "sparse" buys on the matrix pipe. Work items are laid out so each XCD's L2 sees a contiguous band
// of output rows (A halo reuse) and all XCDs stream the same (L2-resident) weights.
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/probe): bit 0 drops the B-fragment loads, bit 1 the A gathers, so
// the MFMA loop can be timed without its memory traffic. The shipped library is built with 0.
#ifndef CPD_GC_ABLATE
#define CPD_GC_ABLATE 0
#endif

namespace {

struct GcParams {
    const float *in;
    const float *w;
    const int32_t *nbr;
    const float *scale, *shift, *residual;
    float *out;
    const int32_t *out_row_map;
    int in_ld, c_in, kc;  // kc = 16-channel chunks
    int kv, n_out, c_out, ntot;
    int res_ld, relu, out_ld, col_group;
    int n_rb, n_cb, items;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Bijective "contiguous chunk per XCD" remap (workgroup b runs on XCD b % 8; speed only).
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// A piece for lane (r,g): channels kc*16+4g .. +3 of input row idx. Rows without a neighbour
// (idx < 0) read row 0 and are zeroed by a select afterwards, so the load itself is unconditional
// (no exec-mask branches between the loads and the MFMAs they feed).
template <bool VEC>
__device__ __forceinline__ f32x4 load_a(const GcParams &p, int idx, int kc, int g) {
    const float *row = p.in + (size_t)(idx < 0 ? 0 : idx) * p.in_ld + kc * 16 + 4 * g;
    f32x4 a;
    if (VEC) {
        a = *reinterpret_cast<const f32x4 *>(row);
    } else {
        const int ch = kc * 16 + 4 * g;
        a[0] = ch + 0 < p.c_in ? row[0] : 0.f;
        a[1] = ch + 1 < p.c_in ? row[1] : 0.f;
        a[2] = ch + 2 < p.c_in ? row[2] : 0.f;
        a[3] = ch + 3 < p.c_in ? row[3] : 0.f;
    }
    return a;
}
__device__ __forceinline__ f32x4 zero_if(f32x4 a, bool z) {
    a[0] = z ? 0.f : a[0]; a[1] = z ? 0.f : a[1]; a[2] = z ? 0.f : a[2]; a[3] = z ? 0.f : a[3];
    return a;
}

template <int MS, int NT, bool VEC, unsigned MASK>
struct TapRegs {
    f32x4 a[MS];
    float b[4][NT];
    __device__ __forceinline__ void load(const GcParams &p, const int (&idx)[MS], const float *wk, int kc, int g) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (MASK & (1u << s)) {
                if (CPD_GC_ABLATE & 2) a[s] = f32x4{(float)g, 1.f, (float)kc, 2.f};
                else a[s] = load_a<VEC>(p, idx[s], kc, g);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (CPD_GC_ABLATE & 1) b[q][nt] = (float)(q + nt + g);
                else b[q][nt] = wk[((size_t)q * p.ntot + nt) * 64];
            }
    }
    __device__ __forceinline__ void mma(const int (&idx)[MS], f32x4 (&acc)[MS][NT]) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (MASK & (1u << s)) a[s] = zero_if(a[s], idx[s] < 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < MS; ++s)
                    if (MASK & (1u << s))
                        acc[s][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][q], b[q][nt], acc[s][nt], 0, 0, 0);
    }
};

// One tap for the row sub-tiles in MASK. Two register sets ping-pong over the 16-channel chunks
// (no copies), so the next chunk's A pieces and B fragments stay in flight under the current
// chunk's MFMAs behind a counted vmcnt.
template <int MS, int NT, bool VEC, unsigned MASK>
__device__ __forceinline__ void tap_compute(const GcParams &p, const int (&idx)[MS], const float *wt, int g,
                                            f32x4 (&acc)[MS][NT]) {
    TapRegs<MS, NT, VEC, MASK> r0, r1;
    const size_t kstride = (size_t)4 * p.ntot * 64;
    r0.load(p, idx, wt, 0, g);
    int kc = 0;
    for (; kc + 1 < p.kc; kc += 2) {
        r1.load(p, idx, wt + (size_t)(kc + 1) * kstride, kc + 1, g);
        r0.mma(idx, acc);
        if (kc + 2 < p.kc) r0.load(p, idx, wt + (size_t)(kc + 2) * kstride, kc + 2, g);
        r1.mma(idx, acc);
    }
    if (kc < p.kc) r0.mma(idx, acc);
}

template <int MS, int NT, bool VEC>
__global__ void __launch_bounds__(256) gather_conv_kernel(GcParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= p.items) return;
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = rb * (16 * MS);

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const size_t w_tap = (size_t)p.kc * 4 * p.ntot * 64;
    // Rulebook column of this wave's rows, fetched one tap ahead (unconditional, clamped loads).
    int rowc[MS];
    bool rowok[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        const int row = row0 + 16 * s + r;
        rowok[s] = row < p.n_out;
        rowc[s] = rowok[s] ? row : p.n_out - 1;
    }
    int idx_nxt[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) idx_nxt[s] = p.nbr ? p.nbr[rowc[s]] : rowc[s];
    for (int t = 0; t < p.kv; ++t) {
        int idx[MS];
        unsigned active = 0;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int v = rowok[s] ? idx_nxt[s] : -1;
            idx[s] = v;
            if (__ballot(v >= 0)) active |= 1u << s;
        }
        if (t + 1 < p.kv) {
#pragma unroll
            for (int s = 0; s < MS; ++s) idx_nxt[s] = p.nbr[(size_t)(t + 1) * p.n_out + rowc[s]];
        }
        if (!active) continue;  // no row of this tile has a neighbour at tap t
        const float *wt = p.w + (size_t)t * w_tap + (size_t)(cb * NT) * 64 + lane;
        if constexpr (MS == 1) {
            tap_compute<MS, NT, VEC, 1u>(p, idx, wt, g, acc);
        } else if constexpr (MS == 2) {
            // exact 16-row skipping: only the sub-tiles that have a neighbour issue MFMAs
            if (active == 3u) tap_compute<MS, NT, VEC, 3u>(p, idx, wt, g, acc);
            else if (active == 1u) tap_compute<MS, NT, VEC, 1u>(p, idx, wt, g, acc);
            else tap_compute<MS, NT, VEC, 2u>(p, idx, wt, g, acc);
        } else {
            tap_compute<MS, NT, VEC, (1u << MS) - 1u>(p, idx, wt, g, acc);
        }
    }

    // Epilogue: C/D layout of 16x16x4: col = lane & 15, row = 4*(lane >> 4) + i.
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (cb * NT + nt) * 16 + r;
        sc[nt] = (p.scale && col < p.c_out) ? p.scale[col] : 1.f;
        sh[nt] = (p.shift && col < p.c_out) ? p.shift[col] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < MS; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 16 * s + 4 * g + i;
            if (row >= p.n_out) continue;
            size_t orow = (size_t)row;
            if (p.out_row_map && !p.col_group) orow = (size_t)p.out_row_map[row];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = (cb * NT + nt) * 16 + r;
                if (col >= p.c_out) continue;
                float v = acc[s][nt][i];
                v = v * sc[nt] + sh[nt];
                if (p.residual) v += p.residual[(size_t)row * p.res_ld + col];
                if (p.relu) v = v > 0.f ? v : 0.f;
                if (p.col_group) {
                    const int grp = col / p.col_group;
                    const size_t drow = (size_t)p.out_row_map[(size_t)grp * p.n_out + row];
                    p.out[drow * p.out_ld + (col - grp * p.col_group)] = v;
                } else {
                    p.out[orow * p.out_ld + col] = v;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float *__restrict__ w, int kv, int c_in, int c_out, int kc,
                                                          int ntot, float *__restrict__ packed) {
    // packed[(((t*KC + kc)*4 + q)*NTOT + nt)*64 + lane] = W[t][kc*16 + 4*(lane>>4) + q][nt*16 + (lane&15)]
    size_t total = (size_t)kv * kc * 4 * ntot * 64;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lane = (int)(i & 63);
    size_t rest = i >> 6;
    int nt = (int)(rest % ntot);
    rest /= ntot;
    int q = (int)(rest & 3);
    rest >>= 2;
    int k = (int)(rest % kc);
    int t = (int)(rest / kc);
    int ch = k * 16 + 4 * (lane >> 4) + q, col = nt * 16 + (lane & 15);
    float v = 0.f;
    if (ch < c_in && col < c_out) v = w[((size_t)t * c_in + ch) * c_out + col];
    packed[i] = v;
}

typedef void (*gc_kernel_t)(GcParams);

template <int MS, int NT>
static gc_kernel_t pick_vec(bool vec) {
    return vec ? gather_conv_kernel<MS, NT, true> : gather_conv_kernel<MS, NT, false>;
}
template <int MS>
static gc_kernel_t pick_nt(int nt, bool vec) {
    switch (nt) {
        case 1: return pick_vec<MS, 1>(vec);
        case 2: return pick_vec<MS, 2>(vec);
        case 4: return pick_vec<MS, 4>(vec);
        case 5: return pick_vec<MS, 5>(vec);
        case 8: return pick_vec<MS, 8>(vec);
    }
    return nullptr;
}
static gc_kernel_t pick(int ms, int nt, bool vec) {
    switch (ms) {
        case 1: return pick_nt<1>(nt, vec);
        case 2: return pick_nt<2>(nt, vec);
        case 4: return pick_nt<4>(nt, vec);
    }
    return nullptr;
}

// Tile choice (measured, tools/sweep_tiles.py on MI355X): per-wave efficiency grows with the tile
// (B fragments reused across MS row sub-tiles, A pieces across NT column tiles) but fp32 MFMA only
// needs one wave per SIMD, so what matters first is having >= ~4 wave tiles per SIMD (4096 items)
// to keep 256 CUs x 4 SIMDs evenly loaded; take the largest tile that still gives that many.
static void choose_tile(int n_out, int ntot, int *ms_out, int *nt_out) {
    static const int cand[][2] = {{2, 8}, {4, 4}, {2, 5}, {2, 4}, {2, 2}, {1, 4}, {1, 5}, {1, 2}, {2, 1}, {1, 1}};
    const long long want = 4096;
    int best_ms = 1, best_nt = 1;
    long long best_items = -1;
    for (auto &c : cand) {
        int ms = c[0], nt = c[1];
        if (ntot % nt) continue;
        long long items = (long long)((n_out + 16 * ms - 1) / (16 * ms)) * (ntot / nt);
        if (items >= want) { *ms_out = ms; *nt_out = nt; return; }
        if (items > best_items) { best_items = items; best_ms = ms; best_nt = nt; }
    }
    *ms_out = best_ms;
    *nt_out = best_nt;
}

}  // namespace

extern "C" size_t cpd_packed_weight_floats(int kv, int c_in, int c_out) {
    if (kv <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return (size_t)kv * ((c_in + 15) / 16) * 4 * ((c_out + 15) / 16) * 64;
}

extern "C" int cpd_pack_weight(const float *w_kio, int kv, int c_in, int c_out, float *packed, cpd_stream_t stream) {
    if (!w_kio || !packed || kv <= 0 || c_in <= 0 || c_out <= 0) return CPD_ERR_ARG;
    int kc = (c_in + 15) / 16, ntot = (c_out + 15) / 16;
    size_t total = cpd_packed_weight_floats(kv, c_in, c_out);
    pack_weight_kernel<<<cpd_div_up((long long)total, 256), 256, 0, cpd_s(stream)>>>(w_kio, kv, c_in, c_out, kc, ntot, packed);
    return cpd_check_launch();
}

static void plan_tile(int n_out, int c_in, int c_out, int in_ld, const void *in, int *ms, int *nt, int *vec) {
    const int ntot = (c_out + 15) / 16;
    choose_tile(n_out, ntot, ms, nt);
    if (const char *e = getenv("CPD_GC_MS")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) *ms = v; }
    if (const char *e = getenv("CPD_GC_NT")) { int v = atoi(e); if ((v == 1 || v == 2 || v == 4 || v == 5 || v == 8) && ntot % v == 0) *nt = v; }
    *vec = (c_in % 16 == 0) && (in_ld % 4 == 0) && (((uintptr_t)in & 15) == 0);
}

extern "C" int cpd_gather_conv_tile(int n_out, int c_in, int c_out, int in_ld, int *ms, int *nt, int *vec) {
    if (n_out <= 0 || c_in <= 0 || c_out <= 0 || !ms || !nt || !vec) return CPD_ERR_ARG;
    plan_tile(n_out, c_in, c_out, in_ld, nullptr, ms, nt, vec);
    return CPD_OK;
}

extern "C" int cpd_gather_conv(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                               int kv, int n_out, int c_out, const float *scale, const float *shift,
                               const float *residual, int res_ld, int relu, float *out, int out_ld,
                               const int32_t *out_row_map, int out_col_group, cpd_stream_t stream) {
    if (!in || !packed_w || !out || n_in < 0 || n_out < 0 || c_in <= 0 || c_out <= 0 || kv <= 0 || in_ld < c_in ||
        (residual && res_ld < c_out) || (!nbr && kv != 1) || out_col_group < 0 || (out_col_group > 0 && !out_row_map) ||
        out_ld < (out_col_group > 0 ? (out_col_group < c_out ? out_col_group : c_out) : c_out))
        return CPD_ERR_ARG;
    if (n_out == 0) return CPD_OK;
    GcParams p;
    p.in = in; p.w = packed_w; p.nbr = nbr; p.scale = scale; p.shift = shift; p.residual = residual;
    p.out = out; p.out_row_map = out_row_map;
    p.in_ld = in_ld; p.c_in = c_in; p.kc = (c_in + 15) / 16;
    p.kv = kv; p.n_out = n_out; p.c_out = c_out; p.ntot = (c_out + 15) / 16;
    p.res_ld = res_ld; p.relu = relu; p.out_ld = out_ld; p.col_group = out_col_group;
    int ms, nt, veci;
    plan_tile(n_out, c_in, c_out, in_ld, in, &ms, &nt, &veci);
    const bool vec = veci != 0;
    gc_kernel_t k = pick(ms, nt, vec);
    if (!k) return CPD_ERR_UNSUPPORTED;
    p.n_rb = (n_out + 16 * ms - 1) / (16 * ms);
    p.n_cb = p.ntot / nt;
    p.items = p.n_rb * p.n_cb;
    int blocks = (p.items + 3) / 4;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, cpd_s(stream), p);
    return cpd_check_launch();
}
