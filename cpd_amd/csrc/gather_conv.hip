// gather_conv.hip -- the one GEMM-shaped op of the path: rulebook-driven implicit GEMM on the
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
// Two kernels share the operand mapping (MFMA k-slot g of 16-channel chunk kc, step q <-> channel
// 16*kc + 4*g + q, so a lane's four steps are one 16-byte piece of a feature row) and the packed
// weight image P[t][kc][g][n][q] (a lane's four steps of one column are one 16-byte piece too):
//
//   wave kernel (gather_conv_kernel<MS,NT>): one wave64 owns a (16*MS) x (16*NT) output tile in
//     accumulator registers across all taps/channels; no LDS, no barriers; A pieces gathered
//     straight from L2, B pieces as coalesced 16-byte loads; 16-row sub-tiles with no neighbour at
//     a tap skip that tap's MFMAs (wave-uniform branch on a ballot) -- what "sparse" buys on the
//     matrix pipe. Used for the sparse backbone and small layers.
//   workgroup kernel (tile_conv_kernel<BM,BN>): 4 waves share a BM x BN tile; per 32-channel stage
//     the gathered A rows and the B panel go global -> registers -> LDS (double buffered, one
//     barrier per stage, next stage's loads in flight under the current MFMAs), every wave then
//     feeds its MFMAs with conflict-free ds_read_b128. Cuts L2->CU traffic by the tile's reuse
//     factor; used for the dense BEV / head convolutions (CPD_GC_DENSE flag).
// Work items are laid out so each XCD's L2 sees a contiguous band of output rows (A halo reuse)
// while all XCDs stream the same L2-resident weights.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/probe): bit 0 drops the B loads, bit 1 the A gathers, so the MFMA
// loop can be timed without its memory traffic. The shipped library is built with 0.
#ifndef CPD_GC_ABLATE
#define CPD_GC_ABLATE 0
#endif

namespace {

struct GcParams {
    const float *in;
    const float *w;
    const void *wb;           // split image of the weights (bf16x3 or f16x2, after the fp32 image) or NULL
    const uint32_t *in_absmax; // f16x2 only, or NULL: device word holding the bits of max |in| -- the kernel scales `in` by the power of two
                              // that puts it at [2^14, 2^15) before the split and undoes it in the epilogue (gradients: fp16's range)
    uint32_t *out_absmax;     // or NULL: absmax block (CPD_ABSMAX_SLOTS words, one per 128-byte line) raised to the bits of max |out| by the
                              // epilogue -- the `in_absmax` of the layers that read `out` (the range guard of the f16x2 inference path)
    const float *dsc;         // f16x2 only: per output column, the power of two that undoes the weights' pre-scale (or NULL)
    const int32_t *nbr;
    const uint32_t *tapmask;  // per 16-row sub-tile: bit t = some row has a neighbour at tap t (or NULL)
    const float *scale, *shift, *residual;
    float *out;
    const int32_t *out_row_map;
    int in_ld, c_in, kc;  // kc = 16-channel chunks
    int n_in_rows;        // rows of `in` (bounds of the row-wave kernels' buffer loads)
    int kv, n_out, c_out, ntot, np;  // np = padded columns (16*ntot)
    int res_ld, relu, out_ld, col_group;
    int n_rb, n_cb, items, n_sub;
    int img_h, img_w;         // window kernel only: the rows are frames x img_h x img_w pixels
    int taps_inner;           // row-wave kernel: stage order (32-channel block outer, tap inner)
    // fp16-pair rows (CPD_GC_*_PAIRS): a row keeps its 4 * C bytes, but every 32-channel block holds the fp16 HIGH terms of its
    // channels (64 B) followed by the fp16 LOW terms (64 B) -- the split x = h + l of SplitF16x2, made ONCE by the epilogue that
    // produced the row instead of by every gather of it (27 taps x ...); the row-wave kernel then takes gathered bits as fragments
    int in_pairs, out_pairs, res_pairs;
    // tap split of the row-wave kernels (small launches): blockIdx.y = z takes every split-th active tap of its row tile and
    // writes its RAW accumulators to part[z][n_out][c_out]; split_finish_kernel sums the parts in order and runs the epilogue
    int split;
    int epi_lds;              // window kernels: the output tile goes through LDS and leaves as whole 16-byte row pieces (set by the launcher when the tile shape allows)
    float *part;
    // row plan of a sub-manifold rulebook (cpd_rulebook_plan; the staged row-wave kernel): per 128-row tile and dz group of 9 taps
    // the sorted list of DISTINCT input rows the group touches and, per (tap, row), the 16-bit position in that list
    const uint16_t *plan_slots;   // [tiles][27][128], 0xffff = no neighbour
    const int32_t *plan_ulist;    // [tiles][3][CPD_PLAN_LIST]
    const int32_t *plan_count;    // [tiles][4]: list lengths of the three groups
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Bijective "contiguous chunk per XCD" remap (workgroup b runs on XCD b % 8; speed only).
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// A piece: 4 consecutive channels starting at `ch` of input row idx. Rows without a neighbour
// (idx < 0) read row 0 and are zeroed by a select afterwards, so the load itself is unconditional
// (no exec-mask branches between the loads and the MFMAs they feed).
template <bool VEC>
__device__ __forceinline__ f32x4 load_a(const GcParams &p, int idx, int ch) {
    const float *row = p.in + (size_t)(idx < 0 ? 0 : idx) * p.in_ld + ch;
    f32x4 a;
    if (VEC) {
        a = *reinterpret_cast<const f32x4 *>(row);
    } else {
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

// Shared epilogue. C/D layout of 16x16x4: col = lane & 15, row = 4*(lane >> 4) + i.
// Power-of-two pre-scale of the input for the fp16 split: s = 2^(14 - floor(log2 max|in|)), inv = 1 / s (both exact)
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

// PAIRS: the instantiation honours p.out_pairs / p.res_pairs (the sparse kernels; elsewhere the flags are refused by the launcher)
template <int MS, int NT, bool PAIRS = false>
__device__ __forceinline__ void epilogue(const GcParams &p, f32x4 (&acc)[MS][NT], int row0, int col0, int r, int g, float acc_scale = 1.f) {
    float sc[NT], sh[NT];
    int grp[NT], cloc[NT];            // column-group scatter (ConvTranspose as one GEMM): group and column inside it, per column tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = col0 + nt * 16 + r;
        sc[nt] = (p.scale && col < p.c_out) ? p.scale[col] : 1.f;
        if (p.dsc && col < p.np) sc[nt] *= p.dsc[col];       // exact (a power of two): (acc * 2^-e) * scale, bit for bit
        sc[nt] *= acc_scale;                                 // the input's pre-scale (a power of two; 1 when unused)
        sh[nt] = (p.shift && col < p.c_out) ? p.shift[col] : 0.f;
        grp[nt] = p.col_group ? col / p.col_group : 0;
        cloc[nt] = col - grp[nt] * p.col_group;
    }
    uint32_t vmax = 0;                // bits of the largest |v| this lane stores (non-negative floats order like their bits; NaN on top)
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        // Residual first, for the whole sub-tile: the loads of all 4 * NT elements are issued back to back and their round trips overlap.
        // (Loaded inside the element loop below, each load sat between the previous element's store and its own use: the compiler
        // cannot move a load above a store through a pointer it cannot tell apart, so an epilogue was 4 * MS * NT SERIAL global round
        // trips -- 16 ... 64 per workgroup; tools/rowplan_bench.py with the epilogue ablated: 354 of 821 us on the 32-channel level.)
        // (column tiles in chunks of NC: 4 * NC live values -- all of a sub-tile at once costs the 128-column kernels their registers)
        constexpr int NC = NT > 4 ? 4 : NT;
#pragma unroll
        for (int c0 = 0; c0 < NT; c0 += NC) {
        float resv[4][NC];
        if (p.residual) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 16 * s + 4 * g + i;
                const int rowc = row < p.n_out ? row : p.n_out - 1;
#pragma unroll
                for (int nc = 0; nc < NC; ++nc) {
                    const int nt = c0 + nc;
                    int col = col0 + nt * 16 + r;
                    col = col < p.c_out ? col : p.c_out - 1;
                    if (PAIRS && p.res_pairs == 2) {  // 16-channel pair rows: k-group (col >> 2) = [hi 4 | lo 4]
                        const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + ((col >> 2) << 4) + ((col & 3) << 1);
                        resv[i][nc] = (float)*reinterpret_cast<const _Float16 *>(rp) + (float)*reinterpret_cast<const _Float16 *>(rp + 8);
                    } else if (PAIRS && p.res_pairs) {       // h + l is exact in fp32
                        const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + ((col >> 5) << 7) + ((col & 31) << 1);
                        resv[i][nc] = (float)*reinterpret_cast<const _Float16 *>(rp) + (float)*reinterpret_cast<const _Float16 *>(rp + 64);
                    } else {
                        resv[i][nc] = p.residual[(size_t)rowc * p.res_ld + col];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 16 * s + 4 * g + i;
            if (row >= p.n_out) continue;
            size_t orow = (size_t)row;
            if (p.out_row_map && !p.col_group) orow = (size_t)p.out_row_map[row];
            int grp_have = -1;
            size_t drow = 0;
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) {
                const int nt = c0 + nc;
                const int col = col0 + nt * 16 + r;
                if (col >= p.c_out) continue;
                float v = acc[s][nt][i];
                v = v * sc[nt] + sh[nt];
                if (p.residual) v += resv[i][nc];
                if (p.relu) v = v > 0.f ? v : 0.f;
                const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                vmax = vb > vmax ? vb : vmax;
                if (p.col_group) {
                    if (grp[nt] != grp_have) {                 // one map entry per (row, group): a column tile rarely spans two
                        grp_have = grp[nt];
                        drow = (size_t)p.out_row_map[(size_t)grp_have * p.n_out + row];
                    }
                    p.out[drow * p.out_ld + cloc[nt]] = v;
                } else if (PAIRS && p.out_pairs) {
                    // fp16-pair row: lanes r, r ^ 1 (columns col, col ^ 1; same row) trade their (h, l): the even lane stores the two
                    // high terms, the odd lane the two low terms -- one 4-byte store per lane, as for an fp32 row
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    const uint32_t mine = (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
                    const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                    const bool odd = (r & 1) != 0;
                    const uint32_t word = odd ? ((theirs >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (theirs << 16));
                    char *op = reinterpret_cast<char *>(p.out + orow * p.out_ld);
                    if (p.out_pairs == 2) op += ((col >> 2) << 4) + (odd ? 8 : 0) + ((col & 2) << 1);          // 16-channel pair rows
                    else op += ((col >> 5) << 7) + (odd ? 64 : 0) + ((col & 30) << 1);
                    *reinterpret_cast<uint32_t *>(op) = word;
                } else {
                    p.out[orow * p.out_ld + col] = v;
                }
            }
        }
        }
    }
    if (p.out_absmax) {
        // one (rarely issued) atomic per wave: the slot is read past the L1 first and raised only when this wave holds a larger
        // value -- after the first workgroups almost nobody does. Slots sit on CPD_ABSMAX_SLOTS different 128-byte lines.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
            vmax = t > vmax ? t : vmax;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
            if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
        }
    }
}

// ================================== wave kernel ==============================================
// One pipeline step = (tap t, 16-channel chunk kc) of this wave's tile. Two register sets ping-pong
// over the FLAT sequence of steps of all ACTIVE taps (no copies, no per-tap bubble): while one
// set's MFMAs issue, the next step's A pieces, B pieces and -- a tap ahead -- its rulebook column
// are in flight behind a counted vmcnt. Which taps are active for which 16-row sub-tile comes from
// the rulebook's tap mask (one word per sub-tile), so inactive (sub-tile, tap) pairs cost nothing.
template <int MS, int NT>
struct StepRegs {
    f32x4 a[MS];
    f32x4 b[NT];  // b[nt][q]
    int idx[MS];
    unsigned act;  // sub-tiles with at least one neighbour at this step's tap
};

// QUAD: the A pieces were gathered quad-shaped (lane -> row lane >> 2, piece lane & 3: a quad reads the 64 contiguous bytes of
// one row's 16-channel chunk -- one L1 access instead of four, see the row-wave kernel) and become fragments (lane -> row
// lane & 15, piece lane >> 4) through a 4 x 16 lane transpose here; `tsrc` = byte address of the source lane for ds_bpermute.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// H16 (16-channel fp16-pair rows, K = 16 MFMA): a gathered 16-byte piece is [hi 4 | lo 4] of channels 4g..4g+3 of its row, a weight
// piece the same of one column: three products (hi*hi, hi*lo, lo*hi) per (sub-tile, column tile), 48 matrix cycles instead of the
// 128 of four fp32 MFMAs -- and no split: the producing epilogue made it
template <int MS, int NT, unsigned MASK, bool QUAD = false, bool H16 = false>
__device__ __forceinline__ void step_mma(StepRegs<MS, NT> &R, f32x4 (&acc)[MS][NT], int tsrc = 0) {
#pragma unroll
    for (int s = 0; s < MS; ++s)
        if (MASK & (1u << s)) {
            R.a[s] = zero_if(R.a[s], R.idx[s] < 0);
            if (QUAD) {
#pragma unroll
                for (int k = 0; k < 4; ++k) R.a[s][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(tsrc, __float_as_int(R.a[s][k])));
            }
        }
    if (H16) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const h16x4 bh = __builtin_bit_cast(h16x4, f32x2{R.b[nt][0], R.b[nt][1]}), bl = __builtin_bit_cast(h16x4, f32x2{R.b[nt][2], R.b[nt][3]});
#pragma unroll
            for (int s = 0; s < MS; ++s)
                if (MASK & (1u << s)) {
                    const h16x4 ah = __builtin_bit_cast(h16x4, f32x2{R.a[s][0], R.a[s][1]}), al = __builtin_bit_cast(h16x4, f32x2{R.a[s][2], R.a[s][3]});
                    f32x4 c = acc[s][nt];
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
                    acc[s][nt] = c;
                }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < MS; ++s)
                if (MASK & (1u << s))
                    acc[s][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.a[s][q], R.b[nt][q], acc[s][nt], 0, 0, 0);
}

template <int MS, int NT, bool VEC, bool H16 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(H16 && MS * NT == 4 ? 4 : 1, 8))) gather_conv_kernel(GcParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: item, row0, col0 and the weight / mask addresses stay scalar
    const int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= p.items) return;
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int r = lane & 15, g = lane >> 4;
    // gather coordinates of the 16-byte-piece (VEC) path: quad-shaped, transposed into fragments in step_mma
    const int lr = VEC ? (lane >> 2) : r, lp = VEC ? (lane & 3) : g;
    const int tsrc = (4 * r + g) << 2;
    const int row0 = rb * (16 * MS);
    const int col0 = cb * NT * 16;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int rowc[MS];
    bool rowok[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        const int row = row0 + 16 * s + r;
        rowok[s] = row < p.n_out;
        rowc[s] = rowok[s] ? row : p.n_out - 1;
    }
    // tap activity of each 16-row sub-tile (wave-uniform); without a mask every tap is active
    const unsigned all_taps = p.kv >= 32 ? 0xffffffffu : ((1u << p.kv) - 1u);
    unsigned tm[MS];
    unsigned any = 0;
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        unsigned m = all_taps;
        if (p.tapmask) {
            const int st = (row0 >> 4) + s;
            m = st < p.n_sub ? p.tapmask[st] : 0u;
        } else if (row0 + 16 * s >= p.n_out) {
            m = 0u;
        }
        tm[s] = __builtin_amdgcn_readfirstlane(m);
        any |= tm[s];
    }
    auto next_tap = [&](int t) -> int {  // next active tap after t, or -1
        const unsigned m = t >= 31 ? 0u : (any & ~((2u << t) - 1u));
        return m ? __builtin_ctz(m) : -1;
    };
    auto act_of = [&](int t) -> unsigned {
        unsigned a = 0;
#pragma unroll
        for (int s = 0; s < MS; ++s) a |= ((tm[s] >> t) & 1u) << s;
        return a;
    };
    // The rulebook columns of this wave's rows for all taps go to LDS once (one round of
    // independent loads); the pipeline then reads them with ds_read, which keeps every global load
    // inside it unconditional and counted on vmcnt alone.
    __shared__ int s_idx[4][32][16 * MS];
    {
        // (round 4: UB columns' loads in flight together, then their LDS writes -- element by element this was one global round trip per
        // 64 entries, 14 in a row for a 27-tap kernel, before the first step: a third of a level-1 wave's life)
        const int total = (p.kv < 32 ? p.kv : 32) * 16 * MS;
        constexpr int UB = 8;
        int *const flat = &s_idx[wave][0][0];
        for (int e0 = 0; e0 < total; e0 += 64 * UB) {
            int v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = e0 + u * 64 + lane, t = e / (16 * MS), row = row0 + (e & (16 * MS - 1));
                v[u] = -1;
                if (e < total && ((any >> t) & 1u) && row < p.n_out) v[u] = p.nbr ? p.nbr[(size_t)t * p.n_out + row] : row;   // a tap no sub-tile has: never read
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = e0 + u * 64 + lane;
                if (e < total && ((any >> (e / (16 * MS))) & 1u)) flat[e] = v[u];
            }
        }
    }
    (void)rowc; (void)rowok;

    int t_cur = any ? __builtin_ctz(any) : -1;
    if (t_cur >= 0) {
        const size_t w_chunk = (size_t)4 * p.np * 4;  // floats per 16-channel chunk of P (H16: per tap of Ph16 -- the same 16 bytes per (g, n))
        const float *wl = (H16 ? reinterpret_cast<const float *>(p.wb) : p.w) + ((size_t)g * p.np + col0 + r) * 4;
        int kc_cur = 0;
        int t_ld = t_cur;  // tap the load cursor points at (stays valid after the last step)

        StepRegs<MS, NT> R0, R1;
        // H16: rows through a buffer resource (the launcher checks < 4 GB): a row without a neighbour is out of range and comes back as
        // zeros -- no select afterwards, 32-bit offsets
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                                 (int)(uint32_t)((size_t)p.n_in_rows * p.in_ld * sizeof(float)), 0x00020000);
        const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
        auto load = [&](StepRegs<MS, NT> &R) {
            R.act = act_of(t_ld);
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                R.idx[s] = s_idx[wave][t_ld][16 * s + lr];
                if (CPD_GC_ABLATE & 2) R.a[s] = f32x4{(float)g, 1.f, (float)kc_cur, 2.f};
                else if (H16) {
                    const uint32_t rowu = (uint32_t)R.idx[s] < (uint32_t)p.n_in_rows ? (uint32_t)R.idx[s] : (uint32_t)p.n_in_rows;
                    R.a[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rowu * row_bytes + (uint32_t)lp * 16u, 0, 0));
                    R.idx[s] = 0;                              // (nothing to zero afterwards)
                } else R.a[s] = load_a<VEC>(p, R.idx[s], kc_cur * 16 + 4 * lp);
            }
            const float *wk = wl + ((size_t)t_ld * p.kc + kc_cur) * w_chunk;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (CPD_GC_ABLATE & 1) R.b[nt] = f32x4{(float)nt, 1.f, (float)g, 2.f};
                else R.b[nt] = *reinterpret_cast<const f32x4 *>(wk + (size_t)nt * 64);
            }
        };
        // move the load cursor to the next (chunk, tap) -- taps INNER, so the taps of one 16-channel chunk re-read
        // neighbouring rows' same 64-byte segments back to back (L2 hits rather than fabric traffic on the
        // multi-chunk dense layers that land here). t_cur = -1 once the last step is loaded
        // (the cursor then keeps pointing at valid memory: the trailing load is a harmless dummy)
        const int t_first = t_cur;
        auto advance = [&]() {
            t_cur = next_tap(t_cur);
            if (t_cur < 0 && kc_cur + 1 < p.kc) {
                ++kc_cur;
                t_cur = t_first;
            }
            if (t_cur >= 0) t_ld = t_cur;
        };
        auto mma = [&](StepRegs<MS, NT> &R) {
            if constexpr (MS == 1) {
                step_mma<MS, NT, 1u, VEC, H16>(R, acc, tsrc);
            } else if constexpr (MS == 2 && H16 && NT == 2) {       // (the 16 -> 32 layer: -13 %; the 16 -> 16 layers +3 % this way: left alone)
                // both sub-tiles of every active step: the rows of an inactive one are out of range (zeros), and ONE form of the
                // step, with the occupancy bound that makes the compiler keep the accumulators in ordinary registers (no AGPR copies),
                // is 46 registers less: 5 -> 7 waves per SIMD
                step_mma<MS, NT, 3u, VEC, H16>(R, acc, tsrc);
            } else if constexpr (MS == 2) {
                if (R.act == 3u) step_mma<MS, NT, 3u, VEC, H16>(R, acc, tsrc);
                else if (R.act == 1u) step_mma<MS, NT, 1u, VEC, H16>(R, acc, tsrc);
                else step_mma<MS, NT, 2u, VEC, H16>(R, acc, tsrc);
            } else {
                step_mma<MS, NT, (1u << MS) - 1u, VEC, H16>(R, acc, tsrc);
            }
        };

        load(R0);
        while (true) {
            advance();
            load(R1);
            mma(R0);
            if (t_cur < 0) break;
            advance();
            load(R0);
            mma(R1);
            if (t_cur < 0) break;
        }
    }
    if constexpr ((H16 || (!VEC && NT == 1)) && NT <= 2 && 16 * (16 * NT + 4) <= 32 * 16 * MS) {     // (!VEC: the 5 -> 16 input layer)
        if (p.epi_lds) {
            // Epilogue through LDS (round 4; the level-1 layers: 16-channel pair rows in, 16- or 32-channel pair rows out, in place):
            // sub-tile by sub-tile the accumulators go to a wave-private tile (this wave's rulebook columns: not needed any more) in
            // row-major order; a lane then owns one 16-byte piece of a row -- [hi 4 | lo 4] of a k-group (16 columns) or the hi / lo
            // halves of 8 channels (32 columns) -- and of the residual row: whole pieces instead of 2-byte loads and 4-byte stores in
            // fragment coordinates. The waves of a workgroup are independent items: wave-local ordering only (DS operations of a wave
            // execute in order).
            constexpr int LD = 16 * NT + 4;
            float *const stile = reinterpret_cast<float *>(&s_idx[wave][0][0]);
            float sc[NT], sh[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = col0 + 16 * nt + r;
                sc[nt] = p.scale ? p.scale[col] : 1.f;
                if (p.dsc) sc[nt] *= p.dsc[col];
                sh[nt] = p.shift ? p.shift[col] : 0.f;
            }
            uint32_t vmax = 0;
            const int urow = lane >> 2, piece = lane & 3;
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) stile[(4 * g + i) * LD + 16 * nt + r] = acc[s][nt][i] * sc[nt] + sh[nt];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int row = row0 + 16 * s + urow;
                const int rowc = row < p.n_out ? row : p.n_out - 1;
                if constexpr (NT == 1) {                     // [hi 4 | lo 4] of k-group `piece`
                    f32x4 v = *reinterpret_cast<const f32x4 *>(stile + urow * LD + 4 * piece);
                    if (p.residual) {
                        const f16x8 x = *reinterpret_cast<const f16x8 *>(reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + (((col0 >> 2) + piece) << 4));
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += (float)x[q] + (float)x[4 + q];
                    }
                    f16x8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (p.relu) v[q] = v[q] > 0.f ? v[q] : 0.f;
                        const uint32_t vb = __float_as_uint(v[q]) & 0x7fffffffu;
                        vmax = (row < p.n_out && vb > vmax) ? vb : vmax;
                        o[q] = (_Float16)v[q];
                        o[4 + q] = (_Float16)(v[q] - (float)o[q]);
                    }
                    if (row < p.n_out) *reinterpret_cast<f16x8 *>(reinterpret_cast<char *>(p.out + (size_t)row * p.out_ld) + (((col0 >> 2) + piece) << 4)) = o;
                } else {                                     // 8 channels of a 32-channel block: high terms, low terms 64 bytes on
                    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stile + urow * LD + 8 * piece);
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stile + urow * LD + 8 * piece + 4);
                    const int unit = (col0 >> 3) + piece;
                    const int off = ((unit >> 2) << 7) + ((unit & 3) << 4);
                    f16x8 rh = {}, rl = {};
                    if (p.residual) {
                        const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + off;
                        rh = *reinterpret_cast<const f16x8 *>(rp);
                        rl = *reinterpret_cast<const f16x8 *>(rp + 64);
                    }
                    f16x8 h, l;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = q < 4 ? v0[q] : v1[q - 4];
                        if (p.residual) v += (float)rh[q] + (float)rl[q];
                        if (p.relu) v = v > 0.f ? v : 0.f;
                        const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                        vmax = (row < p.n_out && vb > vmax) ? vb : vmax;
                        h[q] = (_Float16)v;
                        l[q] = (_Float16)(v - (float)h[q]);
                    }
                    if (row < p.n_out) {
                        char *op = reinterpret_cast<char *>(p.out + (size_t)row * p.out_ld) + off;
                        *reinterpret_cast<f16x8 *>(op) = h;
                        *reinterpret_cast<f16x8 *>(op + 64) = l;
                    }
                }
            }
            if (p.out_absmax) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
                    vmax = t > vmax ? t : vmax;
                }
                if (lane == 0) {
                    uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                    if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
                }
            }
            return;
        }
    }
    epilogue<MS, NT, true>(p, acc, row0, col0, r, g);
}

// ================================ workgroup kernel ===========================================
// LDS images of one 32-channel stage (pieces = 16 B):
//   A: piece (c,g) of tile row m at ((c*4+g)*BM + (m ^ (c*4+g))) * 16   (XOR keeps both the
//      8-lanes-per-row staging writes and the 16-rows-per-group fragment reads conflict-free)
//   B: piece (c,g) of tile col n at ((c*4+g)*BN + n) * 16               (lane-linear both ways)
template <int BM, int BN>
__global__ void __launch_bounds__(256) tile_conv_kernel(GcParams p) {
    constexpr int MS = BM / 32, NT = BN / 32;  // 2 x 2 waves, wave tile (BM/2) x (BN/2)
    constexpr int AJ = BM / 32;                // A pieces staged per thread per stage
    constexpr int BJ = BN / 32;                // B pieces staged per thread per stage
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: what derives from it stays scalar
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * BM, col0 = cb * BN;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging roles: 8 lanes cover one 128-B line (32 channels) of a gathered row
    const int a_piece = tid & 7;  // c = a_piece >> 2, g = a_piece & 3
    const int a_row = tid >> 3;   // + 32*j
    int a_rowc[AJ];
    bool a_ok[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = row0 + a_row + 32 * j;
        a_ok[j] = row < p.n_out;
        a_rowc[j] = a_ok[j] ? row : p.n_out - 1;
    }
    const int sk = p.kc >> 1;  // stages per tap (c_in is a multiple of 32 here)
    const int n_stage = p.kv * sk;
    const size_t w_chunk = (size_t)4 * p.np * 4;  // floats per 16-channel chunk of P

    int idx_cur[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) idx_cur[j] = p.nbr ? p.nbr[a_rowc[j]] : a_rowc[j];

    f32x4 ra[AJ], rbv[BJ];
    auto stage_load = [&](int st) {
        const int t = st / sk, kk = st - t * sk;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int id = a_ok[j] ? idx_cur[j] : -1;
            ra[j] = zero_if(load_a<true>(p, id, kk * 32 + a_piece * 4), id < 0);
        }
        const float *wt = p.w + ((size_t)t * p.kc + kk * 2) * w_chunk;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;  // piece id in the B stage image
            const int cg = id / BN, n = id - cg * BN;
            rbv[j] = *reinterpret_cast<const f32x4 *>(wt + ((size_t)cg * p.np + col0 + n) * 4);
        }
    };
    auto stage_store = [&](int buf) {
        char *sa = smem + buf * (A_BYTES + B_BYTES);
        char *sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = a_row + 32 * j;
            *reinterpret_cast<f32x4 *>(sa + ((a_piece * BM + (m ^ a_piece)) << 4)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4 *>(sb + ((j * 256 + tid) << 4)) = rbv[j];
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int st = 0; st < n_stage; ++st) {
        const int nx = st + 1;
        if (nx < n_stage) {
            const int t_nx = nx / sk;
            if (nx - t_nx * sk == 0) {  // the next stage opens a new tap: switch to its rulebook column
#pragma unroll
                for (int j = 0; j < AJ; ++j) idx_cur[j] = p.nbr[(size_t)t_nx * p.n_out + a_rowc[j]];
            }
            stage_load(nx);
        }
        {
            const char *sa = smem + (st & 1) * (A_BYTES + B_BYTES);
            const char *sb = sa + A_BYTES;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int piece = c * 4 + g;
                f32x4 a[MS], b[NT];
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    const int m = wr * (BM / 2) + 16 * s + r;
                    a[s] = *reinterpret_cast<const f32x4 *>(sa + ((piece * BM + (m ^ piece)) << 4));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = wc * (BN / 2) + 16 * nt + r;
                    b[nt] = *reinterpret_cast<const f32x4 *>(sb + ((piece * BN + n) << 4));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int s = 0; s < MS; ++s)
                            acc[s][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][q], b[nt][q], acc[s][nt], 0, 0, 0);
            }
        }
        if (nx < n_stage) stage_store(nx & 1);
        __syncthreads();
    }
    epilogue<MS, NT>(p, acc, row0 + wr * (BM / 2), col0 + wc * (BN / 2), r, g);
}

// ================================ split-operand arithmetic ==================================
// fp32-equivalent convolution on the 16-bit matrix pipe (16x the fp32 MFMA rate). Two ways of writing an fp32
// operand as a short sum of 16-bit terms whose pairwise products are EXACT in fp32:
//
//   SplitBf16x3: x = h + m + l, three bf16 terms (3 x 8 significand bits; h = rne(x), m = rne(x - h), l = rne(x - h - m),
//     each difference exact in fp32: the split itself is exact over the whole fp32 range). Six partial products of
//     relative weight >= 2^-18 (hh, hm, mh, hl, lh, mm) are accumulated in fp32 by v_mfma_f32_16x16x32_bf16; the dropped
//     ones (ml, lm, ll) are < 2^-26 of the product. 6 MFMAs of K = 32 replace 8 fp32 MFMAs of K = 4 at half the cycles: 2.67x.
//   SplitF16x2: x = h + l, two fp16 terms (2 x 11 significand bits + the sign of l: h = rne(x), l = rne(x - h) represent x to
//     2^-24 relative, i.e. to fp32's own half-ulp, while |x| >= 2^-1; below that l enters fp16's subnormal range and the
//     error is <= 2^-25 ABSOLUTE -- gfx950's MFMA honours fp16 subnormals, tools/f16_probe.hip). Three partial products
//     (hh, hl, lh) on v_mfma_f32_16x16x32_f16; the dropped ll is <= 2^-24 of the product. 3 MFMAs instead of 6: 5.3x the
//     fp32 pipe. fp16's RANGE is the price: activations must stay below 65504 in magnitude (an overflow gives inf / NaN in the
//     output -- loud, not silent), and the weights are pre-scaled per output column by a power of two (exact; folded back
//     in the epilogue) so that a column's largest weight sits in [2^13, 2^14) whatever its magnitude.
// Weights are split once at pack time (images Pb / Ph [t][k32][piece][g][n][8] after the fp32 image); gathered activation
// rows are split while they are staged.
// LDS images of one 32-channel stage, per piece (slots = 16 B = 8 channels):
//   A: k-group g of tile row m at (g*BM + (m ^ 2g)) * 16   (8-byte staging writes of two rows x
//      8 lanes and 16-row fragment reads are both conflict-free)
//   B: k-group g of tile col n at (g*BN + n) * 16          (lane-linear both ways)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4)));

struct SplitBf16x3 {
    static constexpr int NP = 3;
    typedef bf16x8 frag;
    typedef bf16x4 half;
    static __device__ __forceinline__ void split(const f32x4 &x, half (&p)[NP]) {
        p[0] = __builtin_convertvector(x, bf16x4);
        const f32x4 r1 = x - __builtin_convertvector(p[0], f32x4);
        p[1] = __builtin_convertvector(r1, bf16x4);
        const f32x4 r2 = r1 - __builtin_convertvector(p[1], f32x4);
        p[2] = __builtin_convertvector(r2, bf16x4);
    }
    static __device__ __forceinline__ f32x4 mma(const frag (&a)[NP], const frag (&b)[NP], f32x4 c) {   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
struct SplitF16x2 {
    static constexpr int NP = 2;
    typedef f16x8 frag;
    typedef f16x4 half;
    static __device__ __forceinline__ void split(const f32x4 &x, half (&p)[NP]) {
        p[0] = __builtin_convertvector(x, f16x4);
        const f32x4 r1 = x - __builtin_convertvector(p[0], f32x4);
        p[1] = __builtin_convertvector(r1, f16x4);
    }
    static __device__ __forceinline__ f32x4 mma(const frag (&a)[NP], const frag (&b)[NP], f32x4 c) {
        if (!(CPD_GC_ABLATE & 2048)) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);   // 2048: two of three products (timing only)
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
template <class S>
__device__ __forceinline__ typename S::frag join_halves(const typename S::half &lo, const typename S::half &hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ----------------------------- split workgroup kernel (rulebook) -----------------------------
// SH = row sub-tiles of a wave whose A fragments are live at once (the B fragments are re-read MS / SH times per stage);
// LATE_B = the weight stage is fetched AFTER the MFMA block, so its registers are not live across it.
template <class S, int BM, int BN, bool DB, int SH, bool LATE_B, bool SC = false>
__device__ __forceinline__ void tile_conv_split_body(const GcParams &p) {
    float in_s = 1.f, in_inv = 1.f;
    if (SC) in_pow2_scale(p.in_absmax, in_s, in_inv);
    constexpr int NP = S::NP;
    constexpr int MS = BM / 32, NT = BN / 32;       // 2 x 2 waves, wave tile (BM/2) x (BN/2)
    constexpr int AJ = BM / 32;                     // fp32 A pieces (4 channels) staged per thread per stage
    constexpr int BJ = NP * BN / 64;                // 16-byte B pieces staged per thread per stage
    constexpr int A_IMG = BM * 64, B_IMG = BN * 64; // bytes of one piece image
    constexpr int STAGE = NP * (A_IMG + B_IMG);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: what derives from it stays scalar
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * BM, col0 = cb * BN;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_piece = tid & 7;   // 4 channels: k-group a_piece >> 1, half a_piece & 1
    const int a_row = tid >> 3;    // + 32*j
    int a_rowc[AJ];
    bool a_ok[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = row0 + a_row + 32 * j;
        a_ok[j] = row < p.n_out;
        a_rowc[j] = a_ok[j] ? row : p.n_out - 1;
    }
    const int sk = p.c_in >> 5;    // 32-channel stages per tap
    const int n_stage = p.kv * sk;
    const size_t b_stage = (size_t)NP * 4 * p.np * 16;   // bytes of one (tap, k32) block of the split image

    // stage split (small launches, see rowwave_split): workgroup z = blockIdx.y walks its contiguous share of the stage sequence
    const int st_begin = p.split > 1 ? (int)((long long)blockIdx.y * n_stage / p.split) : 0;
    const int st_end = p.split > 1 ? (int)((long long)(blockIdx.y + 1) * n_stage / p.split) : n_stage;
    int idx_cur[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) idx_cur[j] = p.nbr ? p.nbr[(size_t)(st_begin % p.kv) * p.n_out + a_rowc[j]] : a_rowc[j];

    f32x4 ra[AJ];
    bool rz[AJ];                    // "no neighbour": applied when the piece is split, so that nothing
    f32x4u rbv[BJ];                 // between the loads and the MFMAs of the current stage waits on them
    // stage st = (32-channel block kk, tap t), TAP INNER: the nine taps of one channel block re-read (nearly) the
    // same 128-byte row segments back to back, so the re-reads hit L2 instead of going out to the fabric
    auto stage_load_b = [&](int st) {
        const int kk = st / p.kv, t = st - kk * p.kv;
        const char *wt = reinterpret_cast<const char *>(p.wb) + ((size_t)t * sk + kk) * b_stage;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;        // slot in the B stage image: (piece*4 + g)*BN + n
            const int pg = id / BN, n = id - pg * BN;
            rbv[j] = *reinterpret_cast<const f32x4u *>(wt + ((size_t)pg * p.np + col0 + n) * 16);
        }
    };
    auto stage_load = [&](int st) {
        const int kk = st / p.kv;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int id = a_ok[j] ? idx_cur[j] : -1;
            ra[j] = load_a<true>(p, id, kk * 32 + a_piece * 4);
            rz[j] = id < 0;
        }
        if (!LATE_B) stage_load_b(st);
    };
    auto stage_store = [&](int buf) {
        char *sa = smem + (DB ? buf : 0) * STAGE;
        char *sb = sa + NP * A_IMG;
        const int ag = a_piece >> 1, half = a_piece & 1;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = a_row + 32 * j;
            typename S::half pc[NP];
            S::split(SC ? zero_if(ra[j], rz[j]) * in_s : zero_if(ra[j], rz[j]), pc);
            char *dst = sa + (((ag * BM + (m ^ (2 * ag))) << 4) + half * 8);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<typename S::half *>(dst + q * A_IMG) = pc[q];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4u *>(sb + ((j * 256 + tid) << 4)) = rbv[j];
    };

    // the rulebook column of stage st + 1 is fetched while stage st's rows are (one step ahead of its use)
    auto idx_load = [&](int st) {
        if (!p.nbr) return;
        const int t = st % p.kv;
#pragma unroll
        for (int j = 0; j < AJ; ++j) idx_cur[j] = p.nbr[(size_t)t * p.n_out + a_rowc[j]];
    };
    stage_load(st_begin);
    if (st_begin + 1 < st_end) idx_load(st_begin + 1);
    if (LATE_B) stage_load_b(st_begin);
    stage_store(0);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
        const int nx = st + 1;
        if (nx < st_end) {
            stage_load(nx);
            if (nx + 1 < st_end) idx_load(nx + 1);
        }
        {
            const char *sa = smem + (DB ? ((st - st_begin) & 1) : 0) * STAGE;
            const char *sb = sa + NP * A_IMG;
#pragma unroll
            for (int s0 = 0; s0 < MS; s0 += SH) {
                typename S::frag a[SH][NP];
#pragma unroll
                for (int s = 0; s < SH; ++s) {
                    const int m = wr * (BM / 2) + 16 * (s0 + s) + r;
                    const char *src = sa + ((g * BM + (m ^ (2 * g))) << 4);
#pragma unroll
                    for (int q = 0; q < NP; ++q) a[s][q] = *reinterpret_cast<const typename S::frag *>(src + q * A_IMG);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = wc * (BN / 2) + 16 * nt + r;
                    const char *src = sb + ((g * BN + n) << 4);
                    typename S::frag b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
                    for (int s = 0; s < SH; ++s) acc[s0 + s][nt] = S::mma(a[s], b, acc[s0 + s][nt]);
                }
                if (SH < MS) asm volatile("" ::: "memory");     // keep the groups' LDS reads apart (register budget)
            }
        }
        if (!DB) __syncthreads();          // single buffer: everyone is done reading before it is overwritten
        if (LATE_B && nx < st_end) stage_load_b(nx);
        if (nx < st_end) stage_store((nx - st_begin) & 1);
        __syncthreads();
    }
    if (p.split > 1) {                     // raw partial sums: split_finish_kernel adds the parts and runs the epilogue
        float *part = p.part + (size_t)blockIdx.y * p.n_out * p.c_out;
#pragma unroll
        for (int s = 0; s < MS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + wr * (BM / 2) + 16 * s + 4 * g + i;
                if (row >= p.n_out) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) part[(size_t)row * p.c_out + col0 + wc * (BN / 2) + nt * 16 + r] = acc[s][nt][i];
            }
        return;
    }
    if constexpr (BM == 128 && BN == 128) {
        // Epilogue through LDS (round 4; as the window kernels'): 32 rows per pass -- two sub-tiles of the waves of one wave row -- go to
        // a row-major tile, then a thread owns (row, four adjacent columns): 16-byte residual loads and stores, also through the row
        // map / the column-group scatter of a ConvTranspose run as one GEMM (a group is a multiple of four columns wide there).
        // p.epi_lds = 0: `out` / `residual` / the group width do not allow 16-byte pieces -- same walk, element accesses.
        constexpr int LD = BN + 4, C4 = BN / 4, RPI = 256 / C4, UNITS = 32 / RPI;
        float *const stile = reinterpret_cast<float *>(smem);
        float sc[NT], sh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + wc * (BN / 2) + nt * 16 + r;
            sc[nt] = p.scale ? p.scale[col] : 1.f;
            if (p.dsc) sc[nt] *= p.dsc[col];
            sc[nt] *= in_inv;
            sh[nt] = p.shift ? p.shift[col] : 0.f;
        }
        uint32_t vmax = 0;
        const int c4 = tid % C4, urow = tid / C4;
        const int col = col0 + 4 * c4;
        const int grp = p.col_group ? col / p.col_group : 0, cloc = col - grp * p.col_group;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {       // (unrolled: the accumulator indices must be constants)
            __syncthreads();                         // every wave is done with the last stage's LDS (or the previous pass's tile)
            if (wr == (pass >> 1)) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            stile[(16 * s2 + 4 * g + i) * LD + wc * (BN / 2) + 16 * nt + r] = acc[(pass & 1) * 2 + s2][nt][i] * sc[nt] + sh[nt];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int lrow = urow + k * RPI;
                const int row = row0 + pass * 32 + lrow;
                if (row >= p.n_out) continue;
                f32x4 v = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 4 * c4);
                if (p.residual) {
                    const float *rp = p.residual + (size_t)row * p.res_ld + col;
                    if (p.epi_lds) v += *reinterpret_cast<const f32x4 *>(rp);
                    else v += f32x4{rp[0], rp[1], rp[2], rp[3]};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (p.relu) v[q] = v[q] > 0.f ? v[q] : 0.f;
                    const uint32_t vb = __float_as_uint(v[q]) & 0x7fffffffu;
                    vmax = vb > vmax ? vb : vmax;
                }
                if (p.epi_lds) {
                    float *op;
                    if (p.col_group) op = p.out + (size_t)p.out_row_map[(size_t)grp * p.n_out + row] * p.out_ld + cloc;
                    else op = p.out + (p.out_row_map ? (size_t)p.out_row_map[row] : (size_t)row) * p.out_ld + col;
                    *reinterpret_cast<f32x4 *>(op) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (p.col_group) {
                            const int gq = (col + q) / p.col_group;
                            p.out[(size_t)p.out_row_map[(size_t)gq * p.n_out + row] * p.out_ld + (col + q - gq * p.col_group)] = v[q];
                        } else {
                            p.out[(p.out_row_map ? (size_t)p.out_row_map[row] : (size_t)row) * p.out_ld + col + q] = v[q];
                        }
                    }
                }
            }
        }
        if (p.out_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
                vmax = t > vmax ? t : vmax;
            }
            if (lane == 0) {
                uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
            }
        }
    } else {
        epilogue<MS, NT>(p, acc, row0 + wr * (BM / 2), col0 + wc * (BN / 2), r, g, in_inv);
    }
}

template <int BM, int BN, bool DB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BM == 128 && !DB) ? (BN == 128 ? 3 : 4) : 1, (BM == 128 && !DB) ? (BN == 128 ? 3 : 4) : 8)))
tile_conv_bf16_kernel(GcParams p) {
    constexpr bool OCC3 = BM == 128 && !DB;   // the variant squeezed into 3 waves per SIMD (168 registers)
    tile_conv_split_body<SplitBf16x3, BM, BN, DB, OCC3 ? 2 : BM / 32, OCC3>(p);
}
template <int BM, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && BM == 128 ? 3 : 4, BN == 128 && BM == 128 ? 3 : 8)))
tile_conv_f16_kernel(GcParams p) {
    tile_conv_split_body<SplitF16x2, BM, BN, false, BM / 32, true>(p);
}
// ... with the input pre-scaled into fp16's range (GcParams::in_absmax; gradients). A separate kernel, not a branch: the
// unscaled kernels stay exactly what they were (both paths in one kernel cost the inference bench 1.8 %)
template <int BM, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && BM == 128 ? 3 : 4, BN == 128 && BM == 128 ? 3 : 8)))
tile_conv_f16s_kernel(GcParams p) {
    tile_conv_split_body<SplitF16x2, BM, BN, false, BM / 32, true, true>(p);
}

// Split kernel for the dense 3x3 / stride 1 / pad 1 convolutions (BaseBEVBackbone blocks, CenterHead convs) over
// channels-last pixel rows [frames * H * W, C], WITHOUT a rulebook: the input row of output row R at tap (dy, dx) is
// R + dy*W + dx whenever that pixel exists, so the three dx taps of one dy read the same 130-row WINDOW
// [row0 + dy*W - 1, row0 + dy*W + 128] shifted by one row. The window is gathered, split and written to LDS ONCE per
// (32-channel block, dy) and the three taps read their fragments from it at row offsets 0, 1, 2: a third of the
// gathers, splits and LDS writes of the rulebook kernel and no rulebook reads at all. Taps that leave the image
// (y + dy or x + dx out of range; also what keeps frames apart) are zeroed on the A fragments of the affected
// lanes -- a wave-uniform branch that is taken for ~1 in 6 (sub-tile, dx != 0) pairs at W = 188.
// Same tile, fragment layout, XOR-2g swizzle (the image has 136 rows per k-group so that 129 ^ 6 stays inside), register
// diet (SH row sub-tiles live at a time, weights fetched after the MFMA block) as the rulebook kernel.
template <class S, int BN, int SH, bool GLDS = false, int BM = 128, bool SC = false>
__device__ __forceinline__ void window_conv_split_body(const GcParams &p) {
    float in_s = 1.f, in_inv = 1.f;
    if (SC) in_pow2_scale(p.in_absmax, in_s, in_inv);
    constexpr int NP = S::NP;
    // wave grid: 2 x 2 (128 rows x 64 / 128 columns), or 4 x 1: the 16-column tile (tiny c_out heads) and the 256-row tiles
    // (single-column-tile layers, c_out = 64 or <= 16: a wave gets 64 rows instead of 32, twice the MFMAs per staged byte)
    constexpr int WC = (BN >= 64 && BM == 128) ? 2 : 1, WR = 4 / WC;
    constexpr int WM = BM / WR;                         // rows per wave
    constexpr int MS = WM / 16, NT = BN / WC / 16;
    // A window image per piece: [k-group g][window row w][16 B], k-group stride BMW = 0 (mod 16) slots and NO swizzle: a
    // ds_read_b128 lane group is 8 rows of an even k-group and the complementary 8 rows of the odd one out of 16 CONSECUTIVE
    // window rows, so its 16 slots are distinct (mod 16) at every row offset -- the three dx taps read at offsets 0 / 1 / 2
    // (an XOR swizzle is conflict-free only at aligned offsets: 23 % of this kernel's LDS cycles were conflicts with it).
    // The 8-byte writes stay conflict-free through the thread -> piece map instead: 16 consecutive lanes = one k-group,
    // both halves, 8 consecutive rows = 128 contiguous bytes
    constexpr int WROWS = BM + 2, BMW = BM + 16;
    constexpr int AJ = (WROWS * 8 + 255) / 256;     // fp32 A pieces (4 channels) per thread per window
    constexpr int B_SLOTS = NP * 4 * BN;            // 16-byte B pieces of one stage
    constexpr int BJ = (B_SLOTS + 255) / 256;       // ... per thread
    constexpr int A_IMG = BMW * 64, B_IMG = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const sa = smem;
    char *const sb = smem + NP * A_IMG;       // GLDS: two weight buffers, sb and sb + NP * B_IMG
    static_assert(!GLDS || B_SLOTS % 256 == 0, "direct-to-LDS weight stages are whole wave instructions");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;     // (a scalar wave id cost the 16-column instantiation a third of its speed)
    const int wr = wave / WC, wc = wave - wr * WC;
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * BM, col0 = cb * BN;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // which of the four directions exist for this lane's row of sub-tile s: bits 4s + {0 up, 1 down, 2 left, 3 right}
    uint32_t dir_ok = 0;
    {
        const int hw = p.img_h * p.img_w;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int pix = (row0 + wr * WM + 16 * s + r) % hw;
            const int y = pix / p.img_w, x = pix - y * p.img_w;
            const uint32_t b = (y > 0 ? 1u : 0u) | (y < p.img_h - 1 ? 2u : 0u) | (x > 0 ? 4u : 0u) | (x < p.img_w - 1 ? 8u : 0u);
            dir_ok |= b << (4 * s);
        }
    }
    // per (tap, sub-tile): does this lane's row have the tap's pixel (lane_ok, bit 4*tap + s), and -- wave-uniform, in
    // scalar registers -- does any lane of the sub-tile lack it (need_mask, same bit)? Computed once: the stage loop then
    // spends one scalar bit test per sub-tile on borders, and the selects only where an image edge crosses the sub-tile
    uint64_t lane_ok = 0, need_mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const uint32_t b = dir_ok >> (4 * s);
            const bool ok = (dy < 0 ? (b & 1u) : dy > 0 ? (b & 2u) : 1u) && (dx < 0 ? (b & 4u) : dx > 0 ? (b & 8u) : 1u);
            lane_ok |= (uint64_t)(ok ? 1 : 0) << (4 * t + s);
            need_mask |= (uint64_t)(__all(ok) ? 0 : 1) << (4 * t + s);
        }
    }
    need_mask = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(need_mask >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)need_mask);

    const int a_piece = ((lane >> 4) << 1) | (lane & 1);   // 4 channels: k-group a_piece >> 1, half a_piece & 1
    const int a_row = wave * 8 + ((lane >> 1) & 7);        // window row, + 32*j
    const int sk = p.c_in >> 5;
    // stage st = (32-channel block st / 9, tap st % 9): taps inner
    const size_t b_stage = (size_t)NP * 4 * p.np * 16;

    f32x4 ra[AJ];
    f32x4u rbv[BJ];
    // Window rows and (GLDS) weight stages are addressed through BUFFER resources with 32-bit offsets (round 4): a lane's offset is a
    // loop-invariant register plus a per-window / per-stage scalar -- one v_add per load instead of 64-bit multiply-adds, shifts and
    // clamps (12 v_lshl_add_u64 + the compare / select pairs per stage: a third of the loop's non-MFMA vector instructions, which compete
    // with the MFMAs for issue). A row outside the tensor is out of range and reads zeros (it only feeds masked taps): the offset of a
    // negative row wraps past the 4 GB the launcher guarantees the tensor to be smaller than.
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                             (int)(uint32_t)((size_t)p.n_out * p.in_ld * sizeof(float)), 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
    uint32_t voff_a[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) voff_a[j] = (uint32_t)(a_row + 32 * j) * row_bytes + (uint32_t)a_piece * 16u;
    auto load_window = [&](int kk, int dy) {
        const uint32_t s_base = (uint32_t)(row0 + dy * p.img_w - 1) * row_bytes + (uint32_t)kk * 128u;      // (mod 2^32: see above)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff_a[j] + s_base, 0, 0));
        }
    };
    auto store_window = [&]() {
        const int ag = a_piece >> 1, half = a_piece & 1;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS) {
                typename S::half pc[NP];
                S::split(SC ? ra[j] * in_s : ra[j], pc);
                char *dst = sa + (((ag * BMW + w) << 4) + half * 8);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<typename S::half *>(dst + q * A_IMG) = pc[q];
            }
        }
    };
    auto load_weights = [&](int t, int kk) {
        const char *wt = reinterpret_cast<const char *>(p.wb) + ((size_t)t * sk + kk) * b_stage;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;        // slot in the B stage image: (piece*4 + g)*BN + n
            const int pg = id / BN, n = id - pg * BN;
            if (B_SLOTS % 256 == 0 || id < B_SLOTS)
                rbv[j] = *reinterpret_cast<const f32x4u *>(wt + ((size_t)pg * p.np + col0 + n) * 16);
        }
    };
    auto store_weights = [&]() {
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            if (B_SLOTS % 256 == 0 || j * 256 + tid < B_SLOTS) *reinterpret_cast<f32x4u *>(sb + ((j * 256 + tid) << 4)) = rbv[j];
    };
    // GLDS: the weight stage goes global -> LDS directly (global_load_lds_dwordx4: per-lane source address, destination =
    // wave-uniform base + 16 * lane; the stage image is lane-linear in its slot id), into the buffer the PREVIOUS stage read,
    // under this stage's MFMAs: no staging registers, no ds_write pass, one barrier per stage
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wb), 0, (int)(uint32_t)(9u * (uint32_t)sk * (uint32_t)b_stage), 0x00020000);
    uint32_t voff_b[GLDS ? BJ : 1];
    if (GLDS) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;
            const int pg = id / BN, n = id - pg * BN;
            voff_b[j] = (uint32_t)(pg * p.np + col0 + n) * 16u;
        }
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave << 10);     // this wave's 64 slots of a 256-slot pass of the weight image (scalar, once)
    auto issue_weights = [&](int t, int kk, int buf) {
        const uint32_t s_stage = (uint32_t)(t * sk + kk) * (uint32_t)b_stage;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            char *lbase = sb + buf * (NP * B_IMG) + (j << 12) + wave_lds;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void *)lbase, 16, voff_b[GLDS ? j : 0] + s_stage, 0, 0, 0);
        }
    };

    typename S::frag abl_a[(CPD_GC_ABLATE & 256) ? MS : 1][NP], abl_b[(CPD_GC_ABLATE & 256) ? NT : 1][NP];   // diagnostic builds only
    load_window(0, -1);
    if (GLDS) issue_weights(0, 0, 0); else load_weights(0, 0);
    store_window();
    if (!GLDS) store_weights();
    __syncthreads();
    const int n_stage = 9 * sk;
    // stage = (channel block kk, tap t = 3 dyi + dxi), walked with counters: the flat index cost a handful of divisions by 9 and 3 per stage
    int t = 0, dxi = 0, kk = 0;
    for (int st = 0; st < n_stage; ++st) {
        const int dx = dxi - 1;
        const int nx = st + 1;
        int tn = t + 1, kkn = kk, dxn = dxi + 1;                 // the stage after this one
        if (dxn == 3) dxn = 0;
        if (tn == 9) { tn = 0; ++kkn; }
        const bool new_window = nx < n_stage && dxn == 0;        // the next stage starts another dy (or channel block)
        // CPD_GC_ABLATE (diagnostic builds only, wrong results): 64 no weight stages, 128 no window loads / splits / stores,
        // 256 fragments read from LDS in the first stage only, 512 no MFMAs
        if (GLDS && nx < n_stage && !(CPD_GC_ABLATE & 64)) issue_weights(tn, kkn, nx & 1);
        if (new_window && !(CPD_GC_ABLATE & 128)) load_window(kkn, (tn == 0 ? 0 : (tn == 3 ? 1 : 2)) - 1);   // in flight under this stage's MFMAs
        const char *const sbr = sb + (GLDS ? (st & 1) * (NP * B_IMG) : 0);
        {
            const typename S::frag zero = {};
#pragma unroll
            for (int s0 = 0; s0 < MS; s0 += SH) {
                typename S::frag a[SH][NP];
#pragma unroll
                for (int s = 0; s < SH; ++s) {
                    const int w = wr * WM + 16 * (s0 + s) + r + 1 + dx;
                    const char *src = sa + ((g * BMW + w) << 4);
                    if (!(CPD_GC_ABLATE & 256) || st == 0) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[s][q] = *reinterpret_cast<const typename S::frag *>(src + q * A_IMG);
                    } else {
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[s][q] = abl_a[s0 + s][q];
                    }
                    if (CPD_GC_ABLATE & 256) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) { abl_a[s0 + s][q] = a[s][q]; asm volatile("" : "+v"(abl_a[s0 + s][q])); }
                    }
                    if (!(CPD_GC_ABLATE & 16)) {
                        if ((need_mask >> (4 * t + s0 + s)) & 1) {          // scalar test; rarely taken
                            asm volatile("" ::: "memory");                  // keeps this a branch (no if-conversion into 8 selects)
                            const bool ok = (lane_ok >> (4 * t + s0 + s)) & 1;
#pragma unroll
                            for (int q = 0; q < NP; ++q) a[s][q] = ok ? a[s][q] : zero;
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = wc * (BN / WC) + 16 * nt + r;
                    const char *src = sbr + ((g * BN + n) << 4);
                    typename S::frag b[NP];
                    if (!(CPD_GC_ABLATE & 256) || st == 0) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
                    } else {
#pragma unroll
                        for (int q = 0; q < NP; ++q) b[q] = abl_b[nt][q];
                    }
                    if (CPD_GC_ABLATE & 256) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) { abl_b[nt][q] = b[q]; asm volatile("" : "+v"(abl_b[nt][q])); }
                    }
                    if (CPD_GC_ABLATE & 512) {
#pragma unroll
                        for (int s = 0; s < SH; ++s)
#pragma unroll
                            for (int q = 0; q < NP; ++q) asm volatile("" :: "v"(a[s][q]), "v"(b[q]));
                    } else {
#pragma unroll
                        for (int s = 0; s < SH; ++s) acc[s0 + s][nt] = S::mma(a[s], b, acc[s0 + s][nt]);
                    }
                }
                if (SH < MS) asm volatile("" ::: "memory");     // keep the groups' LDS reads apart (register budget)
            }
        }
        if (GLDS) {
            if (new_window && !(CPD_GC_ABLATE & 128)) {
                __syncthreads();           // everyone is done reading the window before it is overwritten
                store_window();
            }
            __syncthreads();               // the next stage's weights have landed (vmcnt(0) is part of it), window visible
        } else {
            __syncthreads();                   // everyone is done reading the images before they are overwritten
            if (nx < n_stage) {
                load_weights(tn, kkn);
                if (new_window) store_window();
                store_weights();
            }
            __syncthreads();
        }
        t = tn; kk = kkn; dxi = dxn;
    }
    if (CPD_GC_ABLATE & 1024) {             // diagnostic builds only: no epilogue (one store keeps the accumulators live)
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < MS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) t += acc[s][nt][0] + acc[s][nt][1] + acc[s][nt][2] + acc[s][nt][3];
        if (t == 123.456f) p.out[tid] = t;
        return;
    }
    if constexpr (BN >= 64) {
        // Epilogue through LDS (round 4; the 64- and 128-column tiles: c_out is a multiple of the tile there): the tile's rows are
        // consecutive rows of `out`, so after a transpose in LDS a thread owns (row, four adjacent columns): one 16-byte store (and
        // residual load), 16 lanes = 256 contiguous bytes of a row -- instead of the fragment-shaped 4-byte accesses (a lane = one column
        // of four rows) of the shared epilogue. 64 rows per pass: the waves of wave-row `pass` write acc * scale + shift in fragment
        // coordinates, then everybody adds the residual, applies the ReLU and stores. p.epi_lds = 0: `out` / `residual` are not
        // 16-byte addressable -- same walk, element accesses.
        constexpr int LD = BN + 4, WN = BN / WC;
        constexpr int C4 = BN / 4, RPI = 256 / C4, UNITS = 64 / RPI;        // 16-byte units per row; rows covered per instruction; units per thread and pass
        static_assert(WM == 64, "a pass is one wave row");
        float *const stile = reinterpret_cast<float *>(smem);
        float sc[NT], sh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + wc * WN + nt * 16 + r;
            sc[nt] = p.scale ? p.scale[col] : 1.f;
            if (p.dsc) sc[nt] *= p.dsc[col];
            sc[nt] *= in_inv;
            sh[nt] = p.shift ? p.shift[col] : 0.f;
        }
        uint32_t vmax = 0;
        const int c4 = tid % C4, urow = tid / C4;
#pragma unroll 1
        for (int pass = 0; pass < WR; ++pass) {
            __syncthreads();                         // every wave is done with the last stage's LDS (or the previous pass's tile)
            if (wr == pass) {
#pragma unroll
                for (int s = 0; s < MS; ++s)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) stile[(16 * s + 4 * g + i) * LD + wc * WN + 16 * nt + r] = acc[s][nt][i] * sc[nt] + sh[nt];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int lrow = urow + k * RPI;
                const int row = row0 + pass * 64 + lrow;
                if (row >= p.n_out) continue;
                f32x4 v = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 4 * c4);
                float *const op = p.out + (size_t)row * p.out_ld + col0 + 4 * c4;
                if (p.residual) {
                    const float *rp = p.residual + (size_t)row * p.res_ld + col0 + 4 * c4;
                    if (p.epi_lds) v += *reinterpret_cast<const f32x4 *>(rp);
                    else v += f32x4{rp[0], rp[1], rp[2], rp[3]};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (p.relu) v[q] = v[q] > 0.f ? v[q] : 0.f;
                    const uint32_t vb = __float_as_uint(v[q]) & 0x7fffffffu;
                    vmax = vb > vmax ? vb : vmax;
                }
                if (p.epi_lds) *reinterpret_cast<f32x4 *>(op) = v;
                else { op[0] = v[0]; op[1] = v[1]; op[2] = v[2]; op[3] = v[3]; }
            }
        }
        if (p.out_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
                vmax = t > vmax ? t : vmax;
            }
            if (lane == 0) {
                uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
            }
        }
    } else {
        epilogue<MS, NT>(p, acc, row0 + wr * WM, col0 + wc * (BN / WC), r, g, in_inv);
    }
}

template <int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 3 : 4, BN == 128 ? 3 : 4)))
window_conv_bf16_kernel(GcParams p) {
    window_conv_split_body<SplitBf16x3, BN, 2>(p);
}
// f16x2: all four row sub-tiles' fragments live at once (the weight fragments are read once per stage); for the 64- and
// 128-column tiles the weight stage goes direct-to-LDS, double buffered (-4...-8 % on the 128-column layers, tools/conv_bench.py)
template <int BN, int BM = 128>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 || BM == 256 ? 3 : 4, BN == 128 || BM == 256 ? 3 : 4)))
window_conv_f16_kernel(GcParams p) {
    window_conv_split_body<SplitF16x2, BN, (BN >= 64 && BM == 128) || (BM == 256 && BN < 64) ? 4 : 2, BN >= 64, BM>(p);
}
template <int BN, int BM = 128>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 || BM == 256 ? 3 : 4, BN == 128 || BM == 256 ? 3 : 4)))
window_conv_f16s_kernel(GcParams p) {             // pre-scaled input (see tile_conv_f16s_kernel)
    window_conv_split_body<SplitF16x2, BN, (BN >= 64 && BM == 128) || (BM == 256 && BN < 64) ? 4 : 2, BN >= 64, BM, true>(p);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// fp16-PAIR DENSE activations (round 5). The BEV maps between two split-fp16 dense layers are stored the way the sparse
// levels have been since round 3: a row keeps its 4 * C bytes, every 32-channel block holds its 32 fp16 HIGH terms (64 B) then its 32
// fp16 LOW terms (64 B) -- the split x = h + l made ONCE by the producing epilogue. The kernels below take the stored bits as MFMA
// fragments: the window / tile staging is a 16-byte load and ONE ds_write_b128 per piece -- no convert / subtract / convert and half
// the LDS write instructions of the fp32-row kernels (whose staging shares the SIMD's issue port with the MFMAs).
// LDS image of the staged rows: ROW-MAJOR, 128 bytes per row = the row's (32-channel block) bytes as they lie in HBM, with the eight
// 16-byte pieces of row w XOR-swizzled by (w & 7): slot(w, piece) = 8 w + (piece ^ (w & 7)). Eight lanes stage one row (one 128-byte
// line, 8 distinct slots), two rows of different parity fill all 16 slot residues: conflict-free b128 writes; a fragment read (lane (r, g)
// reads piece 4 q + g of row base + r) touches 8 consecutive rows of piece P and 8 of piece P ^ 1 per 16-lane group -- again all 16
// residues, at EVERY row offset (the three dx taps read at offsets 0 / 1 / 2). Same partial products as the fp32-row kernels (the split
// of a value is the same two halves whoever makes it), summed in the same order: results equal the fp32-row path's on pair-exact inputs.
template <int BN, int SH, int BM>
__device__ __forceinline__ void window_conv_pairs_body(const GcParams &p) {
    using S = SplitF16x2;
    constexpr int NP = 2;
    constexpr bool GLDS = BN >= 64;
    constexpr int WC = (BN >= 64 && BM == 128) ? 2 : 1, WR = 4 / WC;
    constexpr int WM = BM / WR;                         // rows per wave
    constexpr int MS = WM / 16, NT = BN / WC / 16;
    constexpr int WROWS = BM + 2;
    constexpr int AJ = (WROWS * 8 + 255) / 256;         // 16-byte pieces per thread per window
    constexpr int A_BYTES = (BM + 8) * 128;             // window image (row-major, swizzled)
    constexpr int B_SLOTS = NP * 4 * BN;                // 16-byte B pieces of one stage
    constexpr int BJ = (B_SLOTS + 255) / 256;
    constexpr int B_IMG = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const sa = smem;
    char *const sb = smem + A_BYTES;                    // GLDS: two weight buffers, sb and sb + NP * B_IMG
    static_assert(!GLDS || B_SLOTS % 256 == 0, "direct-to-LDS weight stages are whole wave instructions");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave - wr * WC;
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * BM, col0 = cb * BN;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t dir_ok = 0;
    {
        const int hw = p.img_h * p.img_w;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int pix = (row0 + wr * WM + 16 * s + r) % hw;
            const int y = pix / p.img_w, x = pix - y * p.img_w;
            const uint32_t b = (y > 0 ? 1u : 0u) | (y < p.img_h - 1 ? 2u : 0u) | (x > 0 ? 4u : 0u) | (x < p.img_w - 1 ? 8u : 0u);
            dir_ok |= b << (4 * s);
        }
    }
    uint64_t lane_ok = 0, need_mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const uint32_t b = dir_ok >> (4 * s);
            const bool ok = (dy < 0 ? (b & 1u) : dy > 0 ? (b & 2u) : 1u) && (dx < 0 ? (b & 4u) : dx > 0 ? (b & 8u) : 1u);
            lane_ok |= (uint64_t)(ok ? 1 : 0) << (4 * t + s);
            need_mask |= (uint64_t)(__all(ok) ? 0 : 1) << (4 * t + s);
        }
    }
    need_mask = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(need_mask >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)need_mask);

    const int a_piece = tid & 7;                        // 16-byte piece of the row's 128-byte block: half (h / l) a_piece >> 2, k-group a_piece & 3
    const int a_row = tid >> 3;                         // window row, + 32 * j
    const int sk = p.c_in >> 5;
    const size_t b_stage = (size_t)NP * 4 * p.np * 16;

    f32x4u ra[AJ];
    f32x4u rbv[GLDS ? 1 : BJ];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                             (int)(uint32_t)((size_t)p.n_out * p.in_ld * sizeof(float)), 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
    uint32_t voff_a[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) voff_a[j] = (uint32_t)(a_row + 32 * j) * row_bytes + (uint32_t)a_piece * 16u;
    const int a_dst = (a_row << 7) + ((a_piece ^ (a_row & 7)) << 4);      // (+ 4096 j: 32 rows further on, same swizzle)
    auto load_window = [&](int kk, int dy) {
        const uint32_t s_base = (uint32_t)(row0 + dy * p.img_w - 1) * row_bytes + (uint32_t)kk * 128u;      // (mod 2^32: rows outside the tensor read zeros)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS)
                ra[j] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff_a[j] + s_base, 0, 0));
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS) *reinterpret_cast<f32x4u *>(sa + a_dst + j * 4096) = ra[j];
        }
    };
    auto load_weights = [&](int t, int kk) {
        const char *wt = reinterpret_cast<const char *>(p.wb) + ((size_t)t * sk + kk) * b_stage;
#pragma unroll
        for (int j = 0; j < (GLDS ? 1 : BJ); ++j) {
            const int id = j * 256 + tid;        // slot in the B stage image: (piece*4 + g)*BN + n
            const int pg = id / BN, n = id - pg * BN;
            if (B_SLOTS % 256 == 0 || id < B_SLOTS)
                rbv[j] = *reinterpret_cast<const f32x4u *>(wt + ((size_t)pg * p.np + col0 + n) * 16);
        }
    };
    auto store_weights = [&]() {
#pragma unroll
        for (int j = 0; j < (GLDS ? 1 : BJ); ++j)
            if (B_SLOTS % 256 == 0 || j * 256 + tid < B_SLOTS) *reinterpret_cast<f32x4u *>(sb + ((j * 256 + tid) << 4)) = rbv[j];
    };
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wb), 0, (int)(uint32_t)(9u * (uint32_t)sk * (uint32_t)b_stage), 0x00020000);
    uint32_t voff_b[GLDS ? BJ : 1];
    if (GLDS) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;
            const int pg = id / BN, n = id - pg * BN;
            voff_b[j] = (uint32_t)(pg * p.np + col0 + n) * 16u;
        }
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave << 10);
    auto issue_weights = [&](int t, int kk, int buf) {
        const uint32_t s_stage = (uint32_t)(t * sk + kk) * (uint32_t)b_stage;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            char *lbase = sb + buf * (NP * B_IMG) + (j << 12) + wave_lds;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void *)lbase, 16, voff_b[GLDS ? j : 0] + s_stage, 0, 0, 0);
        }
    };

    load_window(0, -1);
    if (GLDS) issue_weights(0, 0, 0); else load_weights(0, 0);
    store_window();
    if (!GLDS) store_weights();
    __syncthreads();
    const int n_stage = 9 * sk;
    const int w_lane = wr * WM + r + 1;                  // window row of this lane's sub-tile 0 at dx = 0
    int t = 0, dxi = 0, kk = 0;
    for (int st = 0; st < n_stage; ++st) {
        const int nx = st + 1;
        int tn = t + 1, kkn = kk, dxn = dxi + 1;
        if (dxn == 3) dxn = 0;
        if (tn == 9) { tn = 0; ++kkn; }
        const bool new_window = nx < n_stage && dxn == 0;
        if (GLDS && nx < n_stage) issue_weights(tn, kkn, nx & 1);
        if (new_window) load_window(kkn, (tn == 0 ? 0 : (tn == 3 ? 1 : 2)) - 1);   // in flight under this stage's MFMAs
        const char *const sbr = sb + (GLDS ? (st & 1) * (NP * B_IMG) : 0);
        {
            // fragment addresses of this stage (dx = dxi - 1): row w = w_lane + dx + 16 s, piece 4 q + g at slot (4 q + g) ^ (w & 7); 16 s leaves
            // w & 7 alone and 4 q flips address bit 6: one base per stage, sub-tiles and halves are immediates / one XOR
            // (offsets into the __shared__ array, not pointers: an XOR on a pointer goes through a 64-bit integer and comes back as a FLAT
            // pointer -- flat_load_dwordx4 instead of ds_read_b128, found in the ISA)
            const int w0 = w_lane + dxi - 1;
            const uint32_t a_off = (uint32_t)((w0 << 7) + ((g ^ (w0 & 7)) << 4));       // (the window image starts at smem + 0)
            const uint32_t a_off_l = a_off ^ 64u;
            const typename S::frag zero = {};
#pragma unroll
            for (int s0 = 0; s0 < MS; s0 += SH) {
                typename S::frag a[SH][NP];
#pragma unroll
                for (int s = 0; s < SH; ++s) {
                    a[s][0] = *reinterpret_cast<const typename S::frag *>(smem + a_off + (s0 + s) * 2048);
                    a[s][1] = *reinterpret_cast<const typename S::frag *>(smem + a_off_l + (s0 + s) * 2048);
                    if ((need_mask >> (4 * t + s0 + s)) & 1) {          // scalar test; rarely taken
                        asm volatile("" ::: "memory");                  // keeps this a branch (no if-conversion into 8 selects)
                        const bool ok = (lane_ok >> (4 * t + s0 + s)) & 1;
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[s][q] = ok ? a[s][q] : zero;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = wc * (BN / WC) + 16 * nt + r;
                    const char *src = sbr + ((g * BN + n) << 4);
                    typename S::frag b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
                    for (int s = 0; s < SH; ++s) acc[s0 + s][nt] = S::mma(a[s], b, acc[s0 + s][nt]);
                }
                if (SH < MS) asm volatile("" ::: "memory");
            }
        }
        if (GLDS) {
            if (new_window) {
                __syncthreads();           // everyone is done reading the window before it is overwritten
                store_window();
            }
            __syncthreads();               // the next stage's weights have landed (vmcnt(0) is part of it), window visible
        } else {
            __syncthreads();
            if (nx < n_stage) {
                load_weights(tn, kkn);
                if (new_window) store_window();
                store_weights();
            }
            __syncthreads();
        }
        t = tn; kk = kkn; dxi = dxn;
    }
    if constexpr (BN >= 64) {
        // Epilogue through LDS, PAIR rows out: 64 rows per pass to a row-major tile, then a thread owns (row, 8 adjacent columns): the high
        // terms of its 8 channels are one 16-byte store, the low terms another 64 bytes on
        constexpr int LD = BN + 4, WN = BN / WC;
        constexpr int C8 = BN / 8, RPI = 256 / C8, UNITS = 64 / RPI;
        static_assert(WM == 64, "a pass is one wave row");
        float *const stile = reinterpret_cast<float *>(smem);
        float sc[NT], sh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + wc * WN + nt * 16 + r;
            sc[nt] = p.scale ? p.scale[col] : 1.f;
            if (p.dsc) sc[nt] *= p.dsc[col];
            sh[nt] = p.shift ? p.shift[col] : 0.f;
        }
        uint32_t vmax = 0;
        const int c8 = tid % C8, urow = tid / C8;
        const int pair_off = (((col0 >> 3) + c8) >> 2 << 7) + ((((col0 >> 3) + c8) & 3) << 4);
#pragma unroll 1
        for (int pass = 0; pass < WR; ++pass) {
            __syncthreads();
            if (wr == pass) {
#pragma unroll
                for (int s = 0; s < MS; ++s)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) stile[(16 * s + 4 * g + i) * LD + wc * WN + 16 * nt + r] = acc[s][nt][i] * sc[nt] + sh[nt];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int lrow = urow + k * RPI;
                const int row = row0 + pass * 64 + lrow;
                if (row >= p.n_out) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * c8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * c8 + 4);
                f16x8 h, l;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = q < 4 ? v0[q] : v1[q - 4];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                    vmax = vb > vmax ? vb : vmax;
                    h[q] = (_Float16)v;
                    l[q] = (_Float16)(v - (float)h[q]);
                }
                char *const orow = reinterpret_cast<char *>(p.out + (size_t)row * p.out_ld);
                *reinterpret_cast<f16x8 *>(orow + pair_off) = h;
                *reinterpret_cast<f16x8 *>(orow + pair_off + 64) = l;
            }
        }
        if (p.out_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t2 = (uint32_t)__shfl_xor((int)vmax, o);
                vmax = t2 > vmax ? t2 : vmax;
            }
            if (lane == 0) {
                uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
            }
        }
    } else {
        epilogue<MS, NT>(p, acc, row0 + wr * WM, col0 + wc * (BN / WC), r, g, 1.f);       // the 16-column head tile: fp32 rows out (decode reads them)
    }
}
// pair rows in; pair rows out (64- / 128-column tiles) or fp32 rows out (the 16-column head tile)
template <int BN, int BM = 128>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BN == 128 || BM == 256) && BN >= 64 ? 3 : 4, (BN == 128 || BM == 256) && BN >= 64 ? 3 : 4)))
window_conv_f16p_kernel(GcParams p) {
    window_conv_pairs_body<BN, (BN >= 64 && BM == 128) || (BM == 256 && BN < 64) ? 4 : 2, BM>(p);
}

// The 16-column head tile on pair rows (the fused SeparateHead output convs, 320 -> 11): 12 MFMAs per wave and stage cannot hide a
// memory round trip, and the per-stage form above pays one per stage for its 2 KB weight block and one per dy for the window (round 4:
// 1.2 us per stage, 0.13 of the matrix ceiling, 2.7 TB/s of HBM for a kernel that only streams its input). Here a STAGE GROUP = the three
// dx taps of one (32-channel block, dy): at the group's first instruction the NEXT group's window rows (registers) and its three weight
// blocks (6 KB, buffer_load ... lds into the other half of a double buffer) are issued, the group's 36 MFMAs per wave run from LDS with
// no barrier in between, and one barrier pair per group swaps the window -- a round trip is covered by a whole group of every
// workgroup on the CU (3 per CU: 46 KB of LDS each).
template <int BM>
__device__ __forceinline__ void window_conv_pairs16_body(const GcParams &p) {
    using S = SplitF16x2;
    constexpr int NP = 2, BN = 16;
    constexpr int WM = BM / 4;                          // rows per wave (4 x 1 wave grid)
    constexpr int MS = WM / 16;
    constexpr int WROWS = BM + 2;
    constexpr int AJ = (WROWS * 8 + 255) / 256;
    constexpr int A_BYTES = (BM + 8) * 128;
    constexpr int B_TAP = NP * 4 * BN * 16;             // bytes of one tap's weight block (2 KB)
    constexpr int B_GRP = 3 * B_TAP;
    constexpr int B_IMG = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const sa = smem;
    char *const sb = smem + A_BYTES;                    // two group buffers of B_GRP bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = item * BM;

    f32x4 acc[MS][1];
#pragma unroll
    for (int s = 0; s < MS; ++s) acc[s][0] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t dir_ok = 0;
    {
        const int hw = p.img_h * p.img_w;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const int pix = (row0 + wave * WM + 16 * s + r) % hw;
            const int y = pix / p.img_w, x = pix - y * p.img_w;
            const uint32_t b = (y > 0 ? 1u : 0u) | (y < p.img_h - 1 ? 2u : 0u) | (x > 0 ? 4u : 0u) | (x < p.img_w - 1 ? 8u : 0u);
            dir_ok |= b << (4 * s);
        }
    }
    uint64_t lane_ok = 0, need_mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const uint32_t b = dir_ok >> (4 * s);
            const bool ok = (dy < 0 ? (b & 1u) : dy > 0 ? (b & 2u) : 1u) && (dx < 0 ? (b & 4u) : dx > 0 ? (b & 8u) : 1u);
            lane_ok |= (uint64_t)(ok ? 1 : 0) << (4 * t + s);
            need_mask |= (uint64_t)(__all(ok) ? 0 : 1) << (4 * t + s);
        }
    }
    need_mask = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(need_mask >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)need_mask);

    const int a_piece = tid & 7, a_row = tid >> 3;
    const int sk = p.c_in >> 5;
    const uint32_t b_stage = (uint32_t)NP * 4u * (uint32_t)p.np * 16u;       // bytes of one (tap, 32-channel block) of the packed image
    f32x4u ra[AJ];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                             (int)(uint32_t)((size_t)p.n_out * p.in_ld * sizeof(float)), 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
    uint32_t voff_a[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) voff_a[j] = (uint32_t)(a_row + 32 * j) * row_bytes + (uint32_t)a_piece * 16u;
    const int a_dst = (a_row << 7) + ((a_piece ^ (a_row & 7)) << 4);
    auto load_window = [&](int kk, int dy) {
        const uint32_t s_base = (uint32_t)(row0 + dy * p.img_w - 1) * row_bytes + (uint32_t)kk * 128u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS)
                ra[j] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff_a[j] + s_base, 0, 0));
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int w = a_row + 32 * j;
            if (AJ * 32 <= WROWS || w < WROWS) *reinterpret_cast<f32x4u *>(sa + a_dst + j * 4096) = ra[j];
        }
    };
    // a group's weights: three taps x 128 slots = six wave instructions of 64 slots; wave w issues instructions w and w + 4 (< 6).
    // slot id of a tap's block = (piece * 4 + k-group) * 16 + column  ->  packed image offset (pg * np + n) * 16
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wb), 0, (int)(9u * (uint32_t)sk * b_stage), 0x00020000);
    const uint32_t voff_w = (uint32_t)((lane >> 4) * p.np + (lane & 15)) * 16u;      // (the second half of a tap's block: + 4 * np * 16 bytes, a scalar)
    auto issue_weights = [&](int kk, int dyi, int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = wave_s + 4 * k;                  // wave instruction 0 ... 5: tap i >> 1 of the group, half i & 1
            if (i < 6) {
                const int tap = 3 * dyi + (i >> 1);
                const uint32_t s_off = (uint32_t)(tap * sk + kk) * b_stage;
                char *lbase = sb + buf * B_GRP + (i >> 1) * B_TAP + (i & 1) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void *)lbase, 16, voff_w, s_off + (uint32_t)(i & 1) * 64u * (uint32_t)p.np, 0, 0);
            }
        }
    };

    load_window(0, -1);
    issue_weights(0, 0, 0);
    store_window();
    __syncthreads();
    const int n_group = 3 * sk;
    const int w_lane = wave * WM + r + 1;
    int dyi = 0, kk = 0;
    for (int gi = 0; gi < n_group; ++gi) {
        int dyn = dyi + 1, kkn = kk;
        if (dyn == 3) { dyn = 0; ++kkn; }
        const bool more = gi + 1 < n_group;
        if (more) {                                        // the next group's rows and weights: in flight under this group's three taps
            load_window(kkn, dyn - 1);
            issue_weights(kkn, dyn, (gi + 1) & 1);
        }
        const char *const sbr = sb + (gi & 1) * B_GRP;
        const typename S::frag zero = {};
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int t = 3 * dyi + dxi;
            const int w0 = w_lane + dxi - 1;
            const uint32_t a_off = (uint32_t)((w0 << 7) + ((g ^ (w0 & 7)) << 4));
            const uint32_t a_off_l = a_off ^ 64u;
            typename S::frag a[MS][NP];
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                a[s][0] = *reinterpret_cast<const typename S::frag *>(smem + a_off + s * 2048);
                a[s][1] = *reinterpret_cast<const typename S::frag *>(smem + a_off_l + s * 2048);
                if ((need_mask >> (4 * t + s)) & 1) {
                    asm volatile("" ::: "memory");
                    const bool ok = (lane_ok >> (4 * t + s)) & 1;
#pragma unroll
                    for (int q = 0; q < NP; ++q) a[s][q] = ok ? a[s][q] : zero;
                }
            }
            const char *src = sbr + dxi * B_TAP + ((g * BN + r) << 4);
            typename S::frag b[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
            for (int s = 0; s < MS; ++s) acc[s][0] = S::mma(a[s], b, acc[s][0]);
        }
        __syncthreads();               // everyone is done reading the window (and this group's weight buffer)
        if (more) store_window();
        __syncthreads();               // window visible; the next group's weights have landed (vmcnt(0) precedes the barrier)
        dyi = dyn; kk = kkn;
    }
    epilogue<MS, 1>(p, acc, row0 + wave * WM, 0, r, g, 1.f);       // fp32 rows out (decode reads them)
}
template <int BM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM == 256 ? 3 : 4, BM == 256 ? 3 : 4)))      // (256 rows: 46 KB of LDS, three workgroups per CU)
window_conv_f16p16_kernel(GcParams p) {
    window_conv_pairs16_body<BM>(p);
}

// The 128 x 128 split tile kernel on fp16-pair rows (strided dense conv through its pixel table, 1 x 1 / ConvTranspose GEMMs): rows
// gathered through a BUFFER resource with 32-bit offsets (a row without a neighbour, or past the tile's end, is the index clamped to
// n_in: out of range, reads zeros -- no selects, no 64-bit address arithmetic), staged as they lie (no split), the same
// swizzled row-major LDS image as the window kernel above; weights as in tile_conv_f16_kernel (registers -> LDS, fetched after the MFMA
// block); LDS epilogue writing pair rows, also through the row map / the ConvTranspose column-group scatter.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
tile_conv_f16p_kernel(GcParams p) {
    using S = SplitF16x2;
    constexpr int NP = 2, BM = 128, BN = 128;
    constexpr int MS = BM / 32, NT = BN / 32;
    constexpr int AJ = BM / 32;
    constexpr int BJ = NP * BN / 64;
    constexpr int A_BYTES = BM * 128, B_IMG = BN * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const sa = smem;
    char *const sb = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, g = lane >> 4;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * BM, col0 = cb * BN;

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_piece = tid & 7;
    const int a_row = tid >> 3;    // + 32*j
    int a_rowc[AJ];
    bool a_ok[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = row0 + a_row + 32 * j;
        a_ok[j] = row < p.n_out;
        a_rowc[j] = a_ok[j] ? row : p.n_out - 1;
    }
    const int sk = p.c_in >> 5;
    const int n_stage = p.kv * sk;
    const size_t b_stage = (size_t)NP * 4 * p.np * 16;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                             (int)(uint32_t)((size_t)p.n_in_rows * p.in_ld * sizeof(float)), 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.in_ld * 4u, n_in = (uint32_t)p.n_in_rows;
    int idx_cur[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) idx_cur[j] = a_ok[j] ? (p.nbr ? p.nbr[a_rowc[j]] : a_rowc[j]) : -1;

    f32x4u ra[AJ];
    f32x4u rbv[BJ];
    auto stage_load_b = [&](int st) {
        const int kk = st / p.kv, t = st - kk * p.kv;
        const char *wt = reinterpret_cast<const char *>(p.wb) + ((size_t)t * sk + kk) * b_stage;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int id = j * 256 + tid;
            const int pg = id / BN, n = id - pg * BN;
            rbv[j] = *reinterpret_cast<const f32x4u *>(wt + ((size_t)pg * p.np + col0 + n) * 16);
        }
    };
    auto stage_load = [&](int st) {
        const int kk = st / p.kv;
        const uint32_t c_off = (uint32_t)kk * 128u + (uint32_t)a_piece * 16u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const uint32_t id = (uint32_t)idx_cur[j];                        // -1 = no neighbour: 0xffffffff -> clamped to n_in -> out of range -> zeros
            const uint32_t rowi = id < n_in ? id : n_in;
            ra[j] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rowi * row_bytes + c_off, 0, 0));
        }
    };
    const int a_dst = (a_row << 7) + ((a_piece ^ (a_row & 7)) << 4);
    auto stage_store = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4u *>(sa + a_dst + j * 4096) = ra[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4u *>(sb + ((j * 256 + tid) << 4)) = rbv[j];
    };
    auto idx_load = [&](int st) {
        if (!p.nbr) return;
        const int t = st % p.kv;
#pragma unroll
        for (int j = 0; j < AJ; ++j) idx_cur[j] = a_ok[j] ? p.nbr[(size_t)t * p.n_out + a_rowc[j]] : -1;
    };
    stage_load(0);
    if (1 < n_stage) idx_load(1);
    stage_load_b(0);
    stage_store();
    __syncthreads();
    const int m_lane = wr * (BM / 2) + r;
    const uint32_t a_off = (uint32_t)((m_lane << 7) + ((g ^ (m_lane & 7)) << 4));   // + 2048 s; ^ 64 for the low terms (offsets, not pointers: see the window kernel)
    const uint32_t a_off_l = a_off ^ 64u;
    for (int st = 0; st < n_stage; ++st) {
        const int nx = st + 1;
        if (nx < n_stage) {
            stage_load(nx);
            if (nx + 1 < n_stage) idx_load(nx + 1);
        }
        {
            typename S::frag a[MS][NP];
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                a[s][0] = *reinterpret_cast<const typename S::frag *>(smem + a_off + s * 2048);
                a[s][1] = *reinterpret_cast<const typename S::frag *>(smem + a_off_l + s * 2048);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = wc * (BN / 2) + 16 * nt + r;
                const char *src = sb + ((g * BN + n) << 4);
                typename S::frag b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
                for (int s = 0; s < MS; ++s) acc[s][nt] = S::mma(a[s], b, acc[s][nt]);
            }
        }
        __syncthreads();                   // single buffer: everyone is done reading before it is overwritten
        if (nx < n_stage) {
            stage_load_b(nx);
            stage_store();
        }
        __syncthreads();
    }
    {
        // Epilogue through LDS, pair rows out: 32 rows per pass; a thread owns (row, 8 adjacent columns) = two 16-byte stores, also through
        // the row map / the column-group scatter (groups are multiples of 32 columns wide: a unit never straddles a group or a pair block)
        constexpr int LD = BN + 4, C8 = BN / 8, RPI = 256 / C8, UNITS = 32 / RPI;
        float *const stile = reinterpret_cast<float *>(smem);
        float sc[NT], sh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + wc * (BN / 2) + nt * 16 + r;
            sc[nt] = p.scale ? p.scale[col] : 1.f;
            if (p.dsc) sc[nt] *= p.dsc[col];
            sh[nt] = p.shift ? p.shift[col] : 0.f;
        }
        uint32_t vmax = 0;
        const int c8 = tid % C8, urow = tid / C8;
        const int col = col0 + 8 * c8;
        const int grp = p.col_group ? col / p.col_group : 0, cloc = col - grp * p.col_group;
        const int pair_off = ((cloc >> 5) << 7) + (((cloc >> 3) & 3) << 4);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {       // (unrolled: the accumulator indices must be constants)
            __syncthreads();
            if (wr == (pass >> 1)) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            stile[(16 * s2 + 4 * g + i) * LD + wc * (BN / 2) + 16 * nt + r] = acc[(pass & 1) * 2 + s2][nt][i] * sc[nt] + sh[nt];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                const int lrow = urow + k * RPI;
                const int row = row0 + pass * 32 + lrow;
                if (row >= p.n_out) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * c8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * c8 + 4);
                f16x8 h, l;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = q < 4 ? v0[q] : v1[q - 4];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                    vmax = vb > vmax ? vb : vmax;
                    h[q] = (_Float16)v;
                    l[q] = (_Float16)(v - (float)h[q]);
                }
                size_t orow_i;
                if (p.col_group) orow_i = (size_t)p.out_row_map[(size_t)grp * p.n_out + row];
                else orow_i = p.out_row_map ? (size_t)p.out_row_map[row] : (size_t)row;
                char *const orow = reinterpret_cast<char *>(p.out + orow_i * p.out_ld);
                *reinterpret_cast<f16x8 *>(orow + pair_off) = h;
                *reinterpret_cast<f16x8 *>(orow + pair_off + 64) = l;
            }
        }
        if (p.out_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t2 = (uint32_t)__shfl_xor((int)vmax, o);
                vmax = t2 > vmax ? t2 : vmax;
            }
            if (lane == 0) {
                uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
            }
        }
    }
}

// Split kernel for SPARSE layers: a workgroup owns 64*MS output rows x BN columns, wave w the
// rows [16*MS*w, 16*MS*(w+1)) x all BN columns. Only the weights go through LDS (one (tap, 32-channel)
// block of the split image per stage, shared by the four waves); a wave gathers ITS rows' 128-byte channel
// blocks straight into registers (4 lanes per row, 32 B each) and splits them there -- no row is
// fetched or split twice. Taps that none of the workgroup's 16-row sub-tiles has are not
// staged at all; a wave skips the loads and MFMAs of a sub-tile without a neighbour at the tap.
// MS = row sub-tiles per wave: 2 (128-row workgroups) or 1 (64-row workgroups, for layers too small to give every CU a
// 128-row workgroup: twice the workgroups, each staging the same weights for half the rows).
// Taps a row-wave launch can have: a wave's rulebook columns live in LDS (CPD_RW_TAPS x 16 MS words per wave; 28 rather than 32 keeps
// the 32-column kernel at 7 workgroups per CU next to its two weight buffers); larger kernels go to the workgroup kernels.
#define CPD_RW_TAPS 28
// CPD_RW_BOTH (128-column tiles): a stage with ANY active sub-tile loads, transposes and multiplies BOTH of the wave's sub-tiles (the
// rows of an inactive one have no neighbours: their loads are out of range, the fragments zeros) -- one scalar test per stage step
// instead of one per sub-tile: 101 -> 67 scalar instructions per stage; 1.6 of a wave's 16 stages have one active sub-tile only
// (tap-pattern rows). Same box, three pairs: <128,2> 0.1616 -> 0.1531 ms/frame; <32,2> unchanged (0.0913 / 0.0909: left as it was);
// <64,2> 0.1179 -> 0.1266 (95 -> 102 registers: 5 -> 4 waves per SIMD) -- so the 128-column tiles only
#ifndef CPD_RW_BOTH
#define CPD_RW_BOTH 1
#endif
#ifndef CPD_RW_BOTH_MIN
#define CPD_RW_BOTH_MIN 128      // narrowest column tile that does it (64: 0.1171 -> 0.1162, bench equal: not worth a second variant)
#endif
#ifndef CPD_RW_NT
#define CPD_RW_NT 0          // bit 0: output stores of the LDS epilogue non-temporal, bit 1: residual loads, bit 2: rulebook columns (round 5 experiment:
#endif                       // the streams that are read / written once should not evict the gathered rows -- 13.5 reads each -- from the XCD's 4 MB L2)
#ifndef CPD_RW_GLW
#define CPD_RW_GLW 0         // 1 (diagnostic builds): the fp16-pair row-wave kernels' weight stages direct-to-LDS, as the window kernels do.
#endif                       // Measured and NOT kept (round 4, 48 frames, same box): <32,2> 1012 -> 1054 us, <64,2> 1203 -> 1261, <128,2> 1568 -> 1589
                             // (67 instead of 72 registers at 32 columns, same occupancy): the LDS-direct load lands a stage later than the
                             // register copy it replaces can be consumed, and these kernels live on loads in flight

#ifndef CPD_RW_WB
#define CPD_RW_WB 2          // weight buffers of the f16x2 row-wave kernels (diagnostic builds: 1 = round 2's single buffer, two barriers)
#endif
// WB = weight buffers in LDS: with two, a stage's block is committed to the buffer the previous stage is NOT reading, and the barrier
// before the commit ("every wave is done with this stage's weights") goes: one barrier per stage instead of two
// WV = waves per workgroup: 4, or (round 4, fp16-pair rows) 8 -- 256-row workgroups: the (tap, channel block) weight images every
// workgroup re-fetches through the vector L1 serve twice the rows (at 128 columns they are HALF of a 128-row workgroup's L1 traffic)
// EPI != 0: the instantiation's epilogue goes through LDS (see the end of the body; 1: `out` / `residual` are fp16-pair rows, 2: fp32
// rows) -- separate kernels rather than branches: with both epilogues in one kernel, or with the row format a run-time flag, the
// 64-column instantiation spilled an accumulator inside its stage loop
template <class S, int BN, int MS, bool SC = false, bool PS = false, int WB = 1, int WV = 4, int EPI = 0>
__device__ __forceinline__ void rowwave_conv_split_body(const GcParams &p, char *const sb0, int *const sidx) {
    float in_s = 1.f, in_inv = 1.f;
    if (SC) in_pow2_scale(p.in_absmax, in_s, in_inv);
    constexpr int NP = S::NP;
    constexpr int NT = BN / 16;
    constexpr int THREADS = 64 * WV;
    constexpr int WG_ROWS = 16 * WV * MS, WG_SUBS = WV * MS;
    constexpr bool BOTH = CPD_RW_BOTH && MS == 2 && BN >= CPD_RW_BOTH_MIN;
    constexpr int B_SLOTS = NP * 4 * BN;       // 16-byte B pieces of one stage
    constexpr int BJ = (B_SLOTS + THREADS - 1) / THREADS;  // ... staged per thread
    constexpr int B_IMG = BN * 64;             // bytes of one piece image: 4 k-groups x BN x 16

    // the wave id in a SCALAR register: everything derived from it (this wave's tap masks, its row range) then branches on SCC
    // instead of masking EXEC -- the stage loop of the 32-column kernel spent 156 scalar instructions per 12 MFMAs, most of them
    // EXEC bookkeeping around conditions the compiler could not prove wave-uniform
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;          // MFMA fragment coordinates: row of the sub-tile, k-group
    // GATHER coordinates (round 3): a quad of lanes fetches 64 CONTIGUOUS bytes of ONE row (lane -> row lane >> 2, 16-byte piece
    // lane & 3 of each 64-byte half), and a 4 x 16 lane transpose (ds_bpermute) turns the two registers into the fragment of
    // lane (r, g) = channels {4g..4g+3, 16+4g..16+4g+3} of the 32-channel block. Fragment-shaped loads (a quad = 4 different rows)
    // cost the vector L1 one tag lookup PER LANE: 64 per instruction at one per clock -- profiles/r03_rowwave_pmc.json has these
    // kernels at 0.64-0.84 L1 accesses per clock per CU, tools/gather_probe.hip prices the shapes: 14.9 B/clk/CU fragment-shaped,
    // 22 quad-shaped, 20.4 with the transpose. The weights' k order follows through their LDS image (stage_commit).
    const int qr = lane >> 2, qj = lane & 3;
    const int tsrc = (4 * r + g) << 2;               // ds_bpermute source of fragment lane (r, g): quad lane g of row r
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = item / p.n_cb, cb = item - rb * p.n_cb;
    const int row0 = rb * WG_ROWS + wave * (16 * MS), col0 = cb * BN;

    // tap activity: workgroup-wide (which stages exist) and per sub-tile of this wave
    // (no tap masks: every tap of every sub-tile -- kv <= 32 here, the launcher sends larger kernels elsewhere)
    uint32_t wg_mask = p.kv >= 32 ? 0xffffffffu : ((1u << p.kv) - 1u), my_mask[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) my_mask[s] = wg_mask;
    if (p.tapmask) {
        // one vector load of the workgroup's WG_SUBS words (lane i = sub-tile i), then lane reads: as a loop of scalar loads with a
        // bound check each, the compiler waited for every word before asking for the next
        const int sub = rb * WG_SUBS + (lane < WG_SUBS ? lane : 0);
        const uint32_t mine = (lane < WG_SUBS && sub < p.n_sub) ? p.tapmask[sub] : 0u;
        wg_mask = 0;
#pragma unroll
        for (int i = 0; i < WG_SUBS; ++i) wg_mask |= (uint32_t)__builtin_amdgcn_readlane((int)mine, i);
#pragma unroll
        for (int s = 0; s < MS; ++s) my_mask[s] = (uint32_t)__builtin_amdgcn_readlane((int)mine, MS * wave + s);
    }

    if (p.split > 1) {                               // this workgroup's share of the tile's taps: every split-th active one
        const int z = blockIdx.y;
        uint32_t keep = 0, left = wg_mask;
        for (int ord = 0; left; ++ord) {
            const uint32_t bit = left & (0u - left);
            left ^= bit;
            if (ord % p.split == z) keep |= bit;
        }
        wg_mask = keep;
#pragma unroll
        for (int s = 0; s < MS; ++s) my_mask[s] &= keep;
    }

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int rowc[MS];
    bool row_ok[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        const int row = row0 + 16 * s + qr;          // the row this lane GATHERS for
        row_ok[s] = row < p.n_out;
        rowc[s] = row_ok[s] ? row : p.n_out - 1;
    }
    const int sk = p.c_in >> 5;
    const size_t b_stage = (size_t)NP * 4 * p.np * 16;

    // Stage bookkeeping in plain integer / bit arithmetic on scalar registers: the scalar unit is shared by the CU's four
    // SIMDs (one instruction per clock for ~20 resident waves), and with loops over taps and bool-valued lambdas the stage
    // loop of the 32-column kernel cost 166 scalar instructions per 12 MFMAs -- the kernel was bound by THAT.
    if (wg_mask) {
        auto sub_bits = [&](int t) -> uint32_t {           // bit s: this wave's sub-tile s has a neighbour at tap t
            uint32_t b = 0;
#pragma unroll
            for (int s = 0; s < MS; ++s) b |= ((my_mask[s] >> t) & 1u) << s;
            return b;
        };
        // The rulebook columns of the wave's rows go to LDS once, for every tap one of its sub-tiles has (round 3). Fetched stage by
        // stage into registers they were copied from one register set to the next at the loop's end, and the compiler's wait for
        // that copy -- s_waitcnt vmcnt(0) -- landed at the TOP of the loop, right after the stage's gathers and weight loads had
        // been issued: every stage waited for all its loads before its MFMAs (the ISA of round 2's kernel shows it).
        int *const my_idx = sidx + wave * (CPD_RW_TAPS * 16 * MS);
        {
            uint32_t my_any = 0;
#pragma unroll
            for (int s = 0; s < MS; ++s) my_any |= my_mask[s];
            const int wrow0 = row0;
            // all the columns' loads first, then the LDS writes (round 4): as a fetch-and-store loop this was 14 SERIAL global round
            // trips per wave -- load, s_waitcnt vmcnt(0), ds_write, next -- before the first stage could start
            constexpr int U = (CPD_RW_TAPS * 16 * MS + 63) / 64;
            const int n_el = p.kv * 16 * MS;
            int v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = u * 64 + lane, t = e / (16 * MS), row = wrow0 + (e & (16 * MS - 1));
                v[u] = -1;
                if (e < n_el && ((my_any >> t) & 1u) && row < p.n_out)
                    v[u] = p.nbr ? ((CPD_RW_NT & 4) ? __builtin_nontemporal_load(p.nbr + (size_t)t * p.n_out + row) : p.nbr[(size_t)t * p.n_out + row]) : row;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = u * 64 + lane;
                if (e < n_el && ((my_any >> (e / (16 * MS))) & 1u)) my_idx[e] = v[u];
            }
        }

        // Stage order: (tap outer, 32-channel block inner), or -- p.taps_inner, the default -- (block outer, tap inner): the
        // taps of one channel block re-gather neighbouring rows' same 128-byte segments back to back (-6...-8 % on the 32- and
        // 128-channel layers). Either way the stages form one flat sequence; the rulebook column of a stage is fetched two
        // stages ahead of its use (one stage ahead of the gathers it addresses).
        // A stage cursor is (rem, kk): the taps still to come in this round as a bit set (its lowest bit = the cursor's tap) and
        // the 32-channel block; stepping it is a handful of scalar instructions.
        const bool inner = p.taps_inner != 0;
        auto advance = [&](uint32_t &rem, int &kk) -> bool {    // -> the stage after it; false at the end
            if (inner) {
                rem &= rem - 1u;
                if (!rem) { rem = wg_mask; ++kk; }
                return kk < sk;
            }
            if (++kk == sk) { kk = 0; rem &= rem - 1u; }
            return rem != 0u;
        };

        f32x4 araw[MS][2];
        bool az[MS];
        // Rows are fetched through a BUFFER resource over the input tensor (raw buffer, byte offsets): a load beyond num_records
        // returns zeros, so a row without a neighbour (idx < 0, taken as unsigned and clamped to n_in) needs no select afterwards, and
        // the address is one 32-bit multiply instead of 64-bit arithmetic per load (-20 vector instructions per stage). The launcher
        // sends tensors of 4 GB and more to the workgroup kernels.
        const size_t in_bytes = ((size_t)p.n_in_rows * p.in_ld) * sizeof(float);
        constexpr bool use_buf = true;
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, (int)(uint32_t)in_bytes, 0x00020000);
        const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
        f32x4u rbv[BJ];
        auto load_rows = [&](uint32_t on, int kk, int t) {
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                if (BOTH ? on != 0u : ((on >> s) & 1u) != 0u) {
                    const int id = my_idx[t * 16 * MS + 16 * s + qr];       // (-1 for rows past the end: out of range below)
                    az[s] = id < 0;
                    if (CPD_GC_ABLATE & 2) { araw[s][0] = f32x4{(float)id, 1.f, (float)kk, 2.f}; araw[s][1] = araw[s][0]; az[s] = false; continue; }
                    if (use_buf) {
                        const uint32_t rowu = (uint32_t)id < (uint32_t)p.n_in_rows ? (uint32_t)id : (uint32_t)p.n_in_rows;   // -1 -> n_in: out of range
                        const uint32_t off = rowu * row_bytes + (uint32_t)qj * 16u;
                        araw[s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, kk * 128, 0));
                        araw[s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, kk * 128 + 64, 0));
                        az[s] = false;
                        continue;
                    }
                    araw[s][0] = load_a<true>(p, id, kk * 32 + qj * 4);
                    araw[s][1] = load_a<true>(p, id, kk * 32 + 16 + qj * 4);
                }
            }
        };
        const uint32_t b_stage32 = (uint32_t)b_stage;              // (a packed image is far below 4 GB: 32-bit stage offsets)
        auto load_weights = [&](int t, int kk) {
            const char *wt = reinterpret_cast<const char *>(p.wb) + (uint32_t)(t * sk + kk) * b_stage32;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int id = j * THREADS + tid;
                const int pg = id / BN, n = id - pg * BN;
                if (B_SLOTS % THREADS == 0 || id < B_SLOTS) {
                    if (CPD_GC_ABLATE & 1) rbv[j] = f32x4u{(float)t, 1.f, (float)kk, (float)n};
                    else rbv[j] = *reinterpret_cast<const f32x4u *>(wt + (uint32_t)(pg * p.np + col0 + n) * 16u);
                }
            }
        };
        // GLW (fp16-pair kernels, two weight buffers): the stage's weight block goes global -> LDS directly (global_load_lds_dwordx4: the
        // packed image is copied lane-linearly there), into the buffer the PREVIOUS stage read: no staging registers, no ds_write pass
        constexpr bool GLW = CPD_RW_GLW && PS && WB == 2 && B_SLOTS % THREADS == 0;
        auto issue_weights = [&](int t, int kk, int slot) {
            const char *wt = reinterpret_cast<const char *>(p.wb) + (uint32_t)(t * sk + kk) * b_stage32;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int id = j * THREADS + tid;
                const int pg = id / BN, n = id - pg * BN;
                char *lbase = sb0 + slot * (NP * B_IMG) + ((j * THREADS + wave * 64) << 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wt + (uint32_t)(pg * p.np + col0 + n) * 16u),
                                                 (__attribute__((address_space(3))) void *)lbase, 16, 0, 0);
            }
        };
        typename S::frag a[MS][NP];
        auto stage_commit = [&](uint32_t on, int slot) {      // B -> LDS buffer `slot`, A -> split fragments
            char *const sb = sb0 + (WB > 1 ? slot * (NP * B_IMG) : 0);
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                if (!GLW && (B_SLOTS % THREADS == 0 || j * THREADS + tid < B_SLOTS)) {
                    // k order of the LDS weight image = the gathered fragments': the packed slot (piece image q, k-group go, column n)
                    // holds channels 8 go .. 8 go + 7; its half hf (channels 4 (2 go + hf) ..) goes to k-group (2 go + hf) & 3, position go >> 1
                    const int id = j * THREADS + tid;
                    if (PS) {               // fp16-pair input: the fragments come in the natural k order, and so does the image
                        *reinterpret_cast<f32x4u *>(sb + (id << 4)) = rbv[j];
                        continue;
                    }
                    const int pgo = id / BN, n = id - pgo * BN, go = pgo & 3;
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 lo = {__float_as_uint(rbv[j][0]), __float_as_uint(rbv[j][1])}, hi = {__float_as_uint(rbv[j][2]), __float_as_uint(rbv[j][3])};
                    char *img = sb + (pgo >> 2) * B_IMG + (go >> 1) * 8;
                    *reinterpret_cast<u32x2 *>(img + ((((2 * go) & 3) * BN + n) << 4)) = lo;
                    *reinterpret_cast<u32x2 *>(img + ((((2 * go + 1) & 3) * BN + n) << 4)) = hi;
                }
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                if (BOTH ? on != 0u : ((on >> s) & 1u) != 0u) {
                    // rows without a neighbour become zeros in GATHER coordinates, then the lane transpose
                    if (!use_buf) {
                        araw[s][0] = zero_if(araw[s][0], az[s]);
                        araw[s][1] = zero_if(araw[s][1], az[s]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        araw[s][0][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(tsrc, __float_as_int(araw[s][0][k])));
                        araw[s][1][k] = __int_as_float(__builtin_amdgcn_ds_bpermute(tsrc, __float_as_int(araw[s][1][k])));
                    }
                    if (PS || (CPD_GC_ABLATE & 4096)) {     // fp16-pair rows: piece 0 = the high terms of channels 8g .. 8g + 7, piece 1 = the low terms
#pragma unroll
                        for (int q = 0; q < NP; ++q) a[s][q] = __builtin_bit_cast(typename S::frag, araw[s][q & 1]);
                        continue;
                    }
                    typename S::half lo[NP], hi[NP];
                    S::split(SC ? araw[s][0] * in_s : araw[s][0], lo);
                    S::split(SC ? araw[s][1] * in_s : araw[s][1], hi);
#pragma unroll
                    for (int q = 0; q < NP; ++q) a[s][q] = join_halves<S>(lo[q], hi[q]);
                }
            }
        };
        auto stage_mma = [&](uint32_t on, int slot) {
            const char *const sb = sb0 + (WB > 1 ? slot * (NP * B_IMG) : 0);
            if (on) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const char *src = sb + ((g * BN + 16 * nt + r) << 4);
                    typename S::frag b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
                    for (int s = 0; s < MS; ++s) {
                        if (CPD_GC_ABLATE & 4) {               // no MFMAs: keep the operands live
                            if ((on >> s) & 1u) asm volatile("" :: "v"(a[s][0]), "v"(a[s][1]), "v"(b[0]), "v"(b[1]));
                        } else if (BOTH || ((on >> s) & 1u)) acc[s][nt] = S::mma(a[s], b, acc[s][nt]);
                    }
                }
            }
        };

        // stage cursors: c = computing, 1 / 2 = the stages after it (ok = exists); on* = the stage's sub-tile activity bits
        uint32_t rem = wg_mask;                    // cursor of stage 2 (two ahead); t* = the taps of the three stages in flight
        int k2 = 0;
        int tc = __builtin_ctz(rem), kc = 0;
        uint32_t onc = sub_bits(tc);
        bool ok1 = advance(rem, k2);
        int t1 = ok1 ? __builtin_ctz(rem) : 0, k1 = k2;
        uint32_t on1 = ok1 ? sub_bits(t1) : 0u;
        bool ok2 = ok1 && advance(rem, k2);
        int t2 = ok2 ? __builtin_ctz(rem) : 0;
        load_rows(onc, kc, tc);
        if (GLW) issue_weights(tc, kc, 0); else load_weights(tc, kc);
        stage_commit(onc, 0);
        __syncthreads();
        int par = 0;
        while (true) {
            if (ok1) {
                if (GLW) issue_weights(t1, k1, par ^ 1); else load_weights(t1, k1);
                load_rows(on1, k1, t1);
            }
            stage_mma(onc, par);
            if (!ok1) break;
            if (WB == 1 && !(CPD_GC_ABLATE & 8)) __syncthreads();             // every wave is done with this stage's weights
            stage_commit(on1, par ^ 1);
            if (!(CPD_GC_ABLATE & 8)) __syncthreads();
            par ^= 1;
            tc = t1; kc = k1; onc = on1;
            t1 = t2; k1 = k2; ok1 = ok2;
            on1 = ok1 ? sub_bits(t1) : 0u;
            if (ok2) {
                ok2 = advance(rem, k2);
                t2 = ok2 ? __builtin_ctz(rem) : 0;
            }
        }
    }
    if (p.split > 1) {                               // raw partial sums; scale / shift / residual / ReLU happen in split_finish_kernel
        float *part = p.part + (size_t)blockIdx.y * p.n_out * p.c_out;
#pragma unroll
        for (int s = 0; s < MS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 16 * s + 4 * g + i;
                if (row >= p.n_out) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) part[(size_t)row * p.c_out + col0 + nt * 16 + r] = acc[s][nt][i];
            }
        return;
    }
    if constexpr (EPI == 0) {
        epilogue<MS, NT, true>(p, acc, row0, col0, r, g, in_inv);
    } else {
        constexpr bool OP = EPI == 1;                // pair rows (out and residual) or fp32 rows
        // Epilogue through LDS (round 4; rows written in place, pair rows or fp32 rows on either side): a wave's rows are consecutive rows
        // of `out` and of the residual. Its accumulators go to a wave-private LDS tile in row-major order; a lane then owns (row, 8 adjacent
        // channels) units: 32 contiguous bytes of accumulators, two 16-byte pieces of the residual row, two 16-byte stores -- against the
        // shared epilogue's 2- / 4-byte residual loads and 4-byte stores in fragment coordinates (a lane = one column of four rows).
        // The tile aliases the weight buffers and the rulebook columns (nobody needs them any more): EPI_R rows per wave and pass.
        constexpr int EPI_R = (BN == 32 && MS == 2 && WV == 4) ? 32 : 16;
        constexpr int LD = BN + 4, UPR = BN / 8, RPI = 64 / UPR, UNITS = EPI_R / RPI;   // floats per tile row; units per row; rows per instruction; units per lane and pass
        float *const stile = reinterpret_cast<float *>(sb0) + wave * (EPI_R * LD);
        int ln = lane, rc = r;
        asm volatile("" : "+v"(ln), "+v"(rc));       // nothing of the epilogue is computed (or loaded) above this point: the stage loop of the
                                                     // 64-column kernel sits at its register budget (95 of 96) and spilled an accumulator otherwise
        const int cg = ln % UPR, urow = ln / UPR;
        float sc[NT], sh[NT];                        // scale / shift in fragment coordinates, as the accumulators leave the registers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + 16 * nt + rc;
            sc[nt] = p.scale ? p.scale[col] : 1.f;
            if (p.dsc) sc[nt] *= p.dsc[col];
            sc[nt] *= in_inv;
            sh[nt] = p.shift ? p.shift[col] : 0.f;
        }
        uint32_t vmax = 0;
        // a unit's two 16-byte pieces in a row of `out` / `residual`: pair rows -- the high terms of its 8 channels, the low terms 64 bytes
        // on --, fp32 rows -- channels 0-3 and 4-7
        const int pair_off = (((col0 >> 3) + cg) >> 2 << 7) + ((((col0 >> 3) + cg) & 3) << 4);
        const int f32_off = (col0 + 8 * cg) * 4;
        const int res_off = OP ? pair_off : f32_off;
        constexpr int res_2nd = OP ? 64 : 16;
#pragma unroll
        for (int pass = 0; pass < 16 * MS / EPI_R; ++pass) {
            __syncthreads();                         // every wave is done with the last stage's LDS (or, wave by wave, the previous pass's tile)
#pragma unroll
            for (int s = 0; s < EPI_R / 16; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) stile[(16 * s + 4 * g + i) * LD + 16 * nt + r] = acc[pass * (EPI_R / 16) + s][nt][i] * sc[nt] + sh[nt];
            __syncthreads();
            constexpr int UC = UNITS > 2 ? 2 : UNITS;        // units whose residual pieces are in flight together
            f32x4 ra[UC], rb[UC];
#pragma unroll
            for (int k = 0; k < UNITS; ++k) {
                if (p.residual && k % UC == 0) {
#pragma unroll
                    for (int kc = 0; kc < UC; ++kc) {
                        const int row = row0 + pass * EPI_R + urow + (k + kc) * RPI;
                        const int rowc = row < p.n_out ? row : p.n_out - 1;
                        const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + res_off;
                        if (CPD_RW_NT & 2) {
                            ra[kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rp));
                            rb[kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rp + res_2nd));
                        } else {
                            ra[kc] = *reinterpret_cast<const f32x4 *>(rp);
                            rb[kc] = *reinterpret_cast<const f32x4 *>(rp + res_2nd);
                        }
                    }
                }
                const int lrow = urow + k * RPI;
                const int row = row0 + pass * EPI_R + lrow;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * cg);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * cg + 4);
                float t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = q < 4 ? v0[q] : v1[q - 4];
                    if (p.residual) {
                        if constexpr (OP) v += (float)__builtin_bit_cast(f16x8, ra[k % UC])[q] + (float)__builtin_bit_cast(f16x8, rb[k % UC])[q];   // h + l is exact in fp32
                        else v += q < 4 ? ra[k % UC][q] : rb[k % UC][q - 4];
                    }
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    t[q] = v;
                    const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                    vmax = (row < p.n_out && vb > vmax) ? vb : vmax;
                }
                if (row < p.n_out) {
                    char *const orow = reinterpret_cast<char *>(p.out + (size_t)row * p.out_ld);
                    if constexpr (OP) {
                        f16x8 h, l;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            h[q] = (_Float16)t[q];
                            l[q] = (_Float16)(t[q] - (float)h[q]);
                        }
                        if (CPD_RW_NT & 1) {
                            __builtin_nontemporal_store(h, reinterpret_cast<f16x8 *>(orow + pair_off));
                            __builtin_nontemporal_store(l, reinterpret_cast<f16x8 *>(orow + pair_off + 64));
                        } else {
                            *reinterpret_cast<f16x8 *>(orow + pair_off) = h;
                            *reinterpret_cast<f16x8 *>(orow + pair_off + 64) = l;
                        }
                    } else {
                        *reinterpret_cast<f32x4 *>(orow + f32_off) = f32x4{t[0], t[1], t[2], t[3]};
                        *reinterpret_cast<f32x4 *>(orow + f32_off + 16) = f32x4{t[4], t[5], t[6], t[7]};
                    }
                }
            }
        }
        if (p.out_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
                vmax = t > vmax ? t : vmax;
            }
            if (lane == 0) {
                uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
                if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
            }
        }
    }
}

// The second half of a tap-split launch: thread = (row, four adjacent columns); parts summed in z order (deterministic), then the
// shared epilogue's arithmetic in its order -- (sum * (scale * dsc * in_inv)) + shift, + residual, ReLU, absmax, fp32 or pair rows.
__global__ void __launch_bounds__(256) split_finish_kernel(GcParams p) {
    float in_s = 1.f, in_inv = 1.f;
    in_pow2_scale(p.in_absmax, in_s, in_inv);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int quarter = p.c_out >> 2;                 // thread = (row, four adjacent columns): 16-byte accesses throughout
    const int row = (int)(tid / quarter), col = 4 * (int)(tid % quarter);
    uint32_t vmax = 0;
    if (row < p.n_out) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.split; ++z) v += *reinterpret_cast<const f32x4 *>(p.part + ((size_t)z * p.n_out + row) * p.c_out + col);
        const size_t orow = p.out_row_map ? (size_t)p.out_row_map[row] : (size_t)row;
        f32x4 sc = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
        if (p.dsc) sc *= *reinterpret_cast<const f32x4 *>(p.dsc + col);
        sc *= in_inv;
        const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int pair_off = ((col >> 5) << 7) + ((col & 31) << 1);      // byte offset of the four high terms inside a pair row
        f32x4 res = {0.f, 0.f, 0.f, 0.f};
        if (p.residual) {
            if (p.res_pairs) {
                const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)row * p.res_ld) + pair_off;
                const f16x4 h = *reinterpret_cast<const f16x4 *>(rp), l = *reinterpret_cast<const f16x4 *>(rp + 64);
                res = __builtin_convertvector(h, f32x4) + __builtin_convertvector(l, f32x4);
            } else {
                res = *reinterpret_cast<const f32x4 *>(p.residual + (size_t)row * p.res_ld + col);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = v[k] * sc[k] + sh[k];
            if (p.residual) t += res[k];
            if (p.relu) t = t > 0.f ? t : 0.f;
            v[k] = t;
            const uint32_t vb = __float_as_uint(t) & 0x7fffffffu;
            vmax = vb > vmax ? vb : vmax;
        }
        if (p.out_pairs) {
            const f16x4 h = __builtin_convertvector(v, f16x4);
            const f16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
            char *op = reinterpret_cast<char *>(p.out + orow * p.out_ld) + pair_off;
            *reinterpret_cast<f16x4 *>(op) = h;
            *reinterpret_cast<f16x4 *>(op + 64) = l;
        } else {
            *reinterpret_cast<f32x4 *>(p.out + orow * p.out_ld + col) = v;
        }
    }
    if (p.out_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)vmax, o);
            vmax = t > vmax ? t : vmax;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
            if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
        }
    }
}

// (bf16x3, BN = 128, MS = 2 sits at the 3-waves-per-SIMD budget)
template <int BN, int MS = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : 4, BN == 128 && MS == 2 ? 3 : 8)))
rowwave_conv_bf16_kernel(GcParams p) {
    __shared__ __attribute__((aligned(16))) char sb[SplitBf16x3::NP * BN * 64];     // one weight stage: pieces x 4 k-groups x BN x 16 B
    __shared__ int sidx[4 * CPD_RW_TAPS * 16 * MS];                                         // rulebook columns of the waves' rows
    rowwave_conv_split_body<SplitBf16x3, BN, MS>(p, sb, sidx);
}
template <int BN, int MS = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : (BN == 64 && MS == 2 ? 5 : 4), 8)))
rowwave_conv_f16_kernel(GcParams p) {
    __shared__ __attribute__((aligned(16))) char sb[CPD_RW_WB * SplitF16x2::NP * BN * 64];   // weight buffers
    __shared__ int sidx[4 * CPD_RW_TAPS * 16 * MS];
    rowwave_conv_split_body<SplitF16x2, BN, MS, false, false, CPD_RW_WB>(p, sb, sidx);
}
template <int BN, int MS = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : (BN == 64 && MS == 2 ? 5 : 4), 8)))
rowwave_conv_f16s_kernel(GcParams p) {            // pre-scaled input (see tile_conv_f16s_kernel)
    __shared__ __attribute__((aligned(16))) char sb[CPD_RW_WB * SplitF16x2::NP * BN * 64];   // weight buffers
    __shared__ int sidx[4 * CPD_RW_TAPS * 16 * MS];
    rowwave_conv_split_body<SplitF16x2, BN, MS, true, false, CPD_RW_WB>(p, sb, sidx);
}

template <int BN, int MS = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : (BN == 64 && MS == 2 ? 5 : 4), 8)))
rowwave_conv_f16p_kernel(GcParams p) {            // fp16-pair input rows (GcParams::in_pairs): no split in the stage loop
    __shared__ __attribute__((aligned(16))) char sb[CPD_RW_WB * SplitF16x2::NP * BN * 64];   // weight buffers
    __shared__ int sidx[4 * CPD_RW_TAPS * 16 * MS];
    rowwave_conv_split_body<SplitF16x2, BN, MS, false, true, CPD_RW_WB>(p, sb, sidx);
}
#define CPD_RW_EPI_KERNEL(NAME, SCALED, PAIRS_IN)                                                                                             \
    template <int BN, int MS = 2>                                                                                                             \
    __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : (BN == 64 && MS == 2 ? 5 : 4), 8))) \
    NAME(GcParams p) {                                                                                                                         \
        constexpr int SB = CPD_RW_WB * SplitF16x2::NP * BN * 64;                                                                              \
        __shared__ __attribute__((aligned(16))) char sm[SB + 4 * CPD_RW_TAPS * 16 * MS * 4];                                                  \
        static_assert(sizeof(sm) >= 4 * ((BN == 32 && MS == 2) ? 32 : 16) * (BN + 4) * 4, "epilogue tile");                                   \
        rowwave_conv_split_body<SplitF16x2, BN, MS, SCALED, PAIRS_IN, CPD_RW_WB, 4, 2>(p, sm, reinterpret_cast<int *>(sm + SB));               \
    }
CPD_RW_EPI_KERNEL(rowwave_conv_f16e_kernel, false, false)      // fp32 rows in: the LDS-epilogue forms of rowwave_conv_f16_kernel / _f16s_kernel
CPD_RW_EPI_KERNEL(rowwave_conv_f16se_kernel, true, false)
template <int BN, int MS = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 && MS == 2 ? 3 : (BN == 64 && MS == 2 ? 5 : 4), 8)))
rowwave_conv_f16pe_kernel(GcParams p) {           // ... pair rows out too, in place: the epilogue through LDS (GcParams::epi_lds)
    constexpr int SB = CPD_RW_WB * SplitF16x2::NP * BN * 64;                    // weight buffers, then the rulebook columns: ONE array, the
    __shared__ __attribute__((aligned(16))) char sm[SB + 4 * CPD_RW_TAPS * 16 * MS * 4];    // epilogue's tile aliases both
    static_assert(sizeof(sm) >= 4 * ((BN == 32 && MS == 2) ? 32 : 16) * (BN + 4) * 4, "epilogue tile");
    rowwave_conv_split_body<SplitF16x2, BN, MS, false, true, CPD_RW_WB, 4, 1>(p, sm, reinterpret_cast<int *>(sm + SB));
}
#ifndef CPD_RW8_DEFAULT
#define CPD_RW8_DEFAULT 0            // column-tile widths (sum of 32 / 64 / 128) that take the wide-workgroup variant by default
#endif
// ... on wide workgroups: WV = 8 waves / 256 rows (32 and 64 columns) or 6 waves / 192 rows (128 columns: 150 registers = 3 waves per
// SIMD = two 6-wave workgroups per CU; at 8 waves a CU would need 4 per SIMD = 128 registers, which spills 253)
#define CPD_RW_WIDE_WV(BN) ((BN) == 128 ? 6 : 8)
#ifndef CPD_RW8_OCC
#define CPD_RW8_OCC(BN) ((BN) == 128 ? 3 : ((BN) == 64 ? 4 : 6))     // waves per SIMD the wide variants are compiled for
#endif
template <int BN, int WV = CPD_RW_WIDE_WV(BN)>
__global__ void __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(CPD_RW8_OCC(BN), CPD_RW8_OCC(BN))))
rowwave_conv_f16pw_kernel(GcParams p) {
    __shared__ __attribute__((aligned(16))) char sb[CPD_RW_WB * SplitF16x2::NP * BN * 64];
    __shared__ int sidx[WV * CPD_RW_TAPS * 16 * 2];
    rowwave_conv_split_body<SplitF16x2, BN, 2, false, true, CPD_RW_WB, WV>(p, sb, sidx);
}


// ================================ staged row-wave kernel (round 4) ================================
// The row-wave kernels above gather every (row, tap) pair's 128-byte channel block through the vector L1: ~13 gathers per row and
// channel block on the sub-manifold levels, of which (profiles/r03_rowwave_pmc.json) the texture-data return path is 0.96 / 0.92 /
// 0.78 busy -- the kernels run at the L1's gather rate (tools/gather_probe.hip: ~20 B/clk/CU for this shape), neither roof near.
// But a tile of 128 output rows touches few DISTINCT input rows when its rows are neighbours in space: 2.9 per output row in 8 x 8
// (y, x) brick order against 13.5 pairs (tools/unique_probe2.py). This kernel stages those distinct rows ONCE in LDS -- per dz group
// of nine taps: the three z-planes of a 3 x 3 x 3 stencil share no input row, and one plane's list (<= 205 rows on the Waymo-shape
// levels) fits a 224-row window -- as whole 128-byte lines (8 lanes per row: the cheapest gather shape), and forms the MFMA
// fragments of all nine taps with ds_read_b128 at 256 B/clk/CU: a (row, tap) pair costs an LDS read instead of an L1 gather.
// What it needs beyond the rulebook: the ROW PLAN (cpd_rulebook_plan: per tile and group the sorted list of distinct input rows,
// per (tap, row) the position in it). Weights go through LDS per (tap, 32-channel block) exactly as above; tap masks skip
// (16-row sub-tile, tap) pairs without a neighbour as above. A group whose list is longer than the window (never on the measured
// levels; possible in principle up to 9 x 128) is walked in several window passes: rows outside the pass read the window's zero row.
// Input: fp16-pair rows (CPD_GC_IN_PAIRS), c_in % 32 == 0; one column tile (BN = c_out = 32 / 64 / 128); kv = 27.
#define CPD_PLAN_LIST 1152           // list stride per (tile, group): 9 taps x 128 rows can never give more distinct rows
#define CPD_RP_WIN 224               // window slots; slot CPD_RP_WIN is the zero row
#define CPD_RP_P16 225               // plane stride in 16-byte slots: odd, so that the 8 lanes that stage one row (8 planes, same slot) write 8 different
                                     // bank groups; == 1 (mod 16), so that a fragment read of consecutive slots has one 2-way conflict in 16 lanes
// Pipeline (one LDS copy of everything, the NEXT stage's data in registers): a STAGE is (dz group, window pass, 32-channel block,
// tap batch) -- a batch = TB taps whose weights are staged together (all nine of a group at 32 columns, the three dx taps of one dy at
// 64 / 128: 36 / 24 / 48 KB). Per stage: barrier; registers -> LDS (weights; the window and the slot table when they change); barrier;
// the global loads of the stage after it are issued (weights, window rows: whole lines, all in flight together); then the stage's
// taps run from LDS with no barrier in between -- slot read, two fragment reads, B fragments, MFMAs -- so the loads fly under a
// whole batch of MFMAs, and a tap costs no global round trip. (The first version staged one tap's weights per barrier interval,
// as the row-wave kernels do: at 3-4 workgroups per CU the L2 latency of every 4 KB weight block was exposed: 0.128 vs 0.093
// ms/frame on the 32-channel level.)
// ROWS = output rows per workgroup: 128 (4 waves) or 256 (8 waves: a stage's weights serve twice the rows -- the weight blocks every
// tile re-fetches are the part of the L1 traffic that staging the rows does not touch -- and a CU holds 16 waves instead of 8-12;
// the window grows to 384 slots: 366 distinct rows per dz group at most on the measured levels in 8 x 8 brick order).
template <int ROWS> struct RpGeom;
template <> struct RpGeom<128> { static constexpr int WIN = 224, P16 = 225; };   // P16: plane stride in 16-byte slots, odd and == 1 (mod 16)
template <> struct RpGeom<256> { static constexpr int WIN = 384, P16 = 385; };
template <int BN, int TB, int ROWS>
__device__ __forceinline__ void rowplan_conv_body(const GcParams &p, char *const smem) {
    constexpr int NTHR = 2 * ROWS, WIN = RpGeom<ROWS>::WIN, P16 = RpGeom<ROWS>::P16;
    typedef SplitF16x2 S;
    constexpr int NP = 2, NT = BN / 16, MS = 2;
    constexpr int NB = 9 / TB;                 // stages (tap batches) per window; TB = taps per weight stage
    char *const sw = smem;                                                  // one stage's weights
    char *const swin = smem + TB * NP * BN * 64;                            // the window: 8 planes x P16 slots
    uint16_t *const sslot = reinterpret_cast<uint16_t *>(swin + 8 * P16 * 16);
    constexpr int B_SLOTS = NP * 4 * BN;       // 16-byte B pieces of one tap (a multiple of 256)
    constexpr int B_IMG = BN * 64;             // bytes of one piece image of a tap
    constexpr int W_PIECES = TB * B_SLOTS;     // 16-byte pieces of one stage's weights
    constexpr int WJ = (W_PIECES + NTHR - 1) / NTHR;                       // ... per thread
    constexpr int PLANE = P16 * 16;
    constexpr int SJ = (WIN * 8 + NTHR - 1) / NTHR;                        // window pieces per thread: 7 / 6
    constexpr int SSTEP = NTHR / 8;                                        // slots between a thread's pieces
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);

    uint32_t wg_mask = (1u << 27) - 1u, my_mask[MS] = {wg_mask, wg_mask};
    if (p.tapmask) {
        wg_mask = 0;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) {
            const int sub = tile * (ROWS / 16) + i;
            const uint32_t m = sub < p.n_sub ? p.tapmask[sub] : 0u;
            wg_mask |= m;
#pragma unroll
            for (int s = 0; s < MS; ++s)
                if (i == MS * wave + s) my_mask[s] = m;
        }
#pragma unroll
        for (int s = 0; s < MS; ++s) my_mask[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_mask[s]);
        wg_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)wg_mask);
    }
    int n_list[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) n_list[q] = __builtin_amdgcn_readfirstlane(p.plan_count[tile * 4 + q]);

    f32x4 acc[MS][NT];
#pragma unroll
    for (int s = 0; s < MS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[s][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the zero row of the window (slot CPD_RP_WIN of every plane): written once, never overwritten
    if (tid < 8) *reinterpret_cast<f32x4u *>(swin + tid * PLANE + WIN * 16) = f32x4u{0.f, 0.f, 0.f, 0.f};

    const int sk = p.c_in >> 5;
    const uint32_t b_stage32 = (uint32_t)((size_t)NP * 4 * p.np * 16);     // bytes of one (tap, channel block) of the packed image
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0,
                                                                              (int)(uint32_t)(((size_t)p.n_in_rows * p.in_ld) * sizeof(float)), 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.in_ld * 4u;
    const int pc = tid & 7, slot0 = tid >> 3;                               // window piece of this thread: plane pc, slots slot0 + SSTEP j

    // ---- the stage sequence (scalar state; every workgroup-uniform)
    struct Stage { int grp, base, kk, bt; };
    auto batch_mask = [&](int grp, int bt) -> uint32_t { return (wg_mask >> (9 * grp + TB * bt)) & ((1u << TB) - 1u); };
    // -> the stage after `st`; new_win: it needs another window (its group / pass / channel block differs). false at the end
    auto advance = [&](Stage &st, bool &new_win) -> bool {
        new_win = false;
        while (true) {
            if (++st.bt < NB) {
                if (batch_mask(st.grp, st.bt)) return true;
                continue;
            }
            st.bt = -1;
            new_win = true;
            if (++st.kk < sk) continue;
            st.kk = 0;
            st.base += WIN;
            if (st.base < n_list[st.grp]) continue;
            st.base = 0;
            do { ++st.grp; } while (st.grp < 3 && !((wg_mask >> (9 * st.grp)) & 0x1ffu));
            if (st.grp >= 3) return false;
        }
    };

    // ---- registers of the stage in flight
    f32x4u wreg[WJ];                 // weights of the next stage
    f32x4u rows[SJ];                 // window rows of the next stage (when it opens a window)
    f32x4u sreg;                     // slot table piece (threads 0 .. 143) when the group changes
    auto issue_weights = [&](const Stage &st) {
        const uint32_t bm = batch_mask(st.grp, st.bt);
        const int t0 = 9 * st.grp + TB * st.bt;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int i = j * NTHR + tid;                                   // piece of the stage image: tap i / B_SLOTS, piece i % B_SLOTS of it
            const int tt = i / B_SLOTS, within = i - tt * B_SLOTS;
            if ((W_PIECES % NTHR != 0 && i >= W_PIECES) || !((bm >> tt) & 1u)) continue;
            const char *wt = reinterpret_cast<const char *>(p.wb) + (uint32_t)((t0 + tt) * sk + st.kk) * b_stage32;
            if (CPD_GC_ABLATE & 65536) wreg[j] = f32x4u{(float)tt, 1.f, (float)j, 2.f};       // diagnostic builds: no weight loads
            else wreg[j] = *reinterpret_cast<const f32x4u *>(wt + (uint32_t)within * 16u);
        }
    };
    auto write_weights = [&](const Stage &st) {
        const uint32_t bm = batch_mask(st.grp, st.bt);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int i = j * NTHR + tid;
            const int tt = i / B_SLOTS;
            if ((W_PIECES % NTHR != 0 && i >= W_PIECES) || !((bm >> tt) & 1u)) continue;
            *reinterpret_cast<f32x4u *>(sw + (i << 4)) = wreg[j];           // (tap tt's image starts at tt * NP * B_IMG = tt * B_SLOTS * 16)
        }
    };
    auto issue_window = [&](const Stage &st, bool new_grp) {
        const int32_t *const ulist = p.plan_ulist + ((size_t)tile * 3 + st.grp) * (9 * ROWS) + st.base;
        const int n_stage = n_list[st.grp] - st.base < WIN ? n_list[st.grp] - st.base : WIN;
        uint32_t roff[SJ];
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int slot = slot0 + SSTEP * j;
            const int id = slot < n_stage ? ulist[slot] : p.n_in_rows;      // beyond the list: out of range = zeros, no memory access
            roff[j] = (uint32_t)id * row_bytes + (uint32_t)pc * 16u;
        }
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            if (CPD_GC_ABLATE & 131072) rows[j] = f32x4u{(float)roff[j], 1.f, 0.f, 2.f};                      // diagnostic builds: no row loads
            else rows[j] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, roff[j], st.kk * 128, 0));
        }
        if (new_grp && tid < 9 * ROWS / 8)                                  // 9 taps x ROWS rows x 2 bytes, in 16-byte pieces
            sreg = *reinterpret_cast<const f32x4u *>(reinterpret_cast<const char *>(p.plan_slots + ((size_t)tile * 27 + 9 * st.grp) * ROWS) + tid * 16);
    };
    auto write_window = [&](bool new_grp) {
#pragma unroll
        for (int j = 0; j < SJ; ++j)
            if (WIN % SSTEP == 0 || slot0 + SSTEP * j < WIN) *reinterpret_cast<f32x4u *>(swin + pc * PLANE + (slot0 + SSTEP * j) * 16) = rows[j];
        if (new_grp && tid < 9 * ROWS / 8) *reinterpret_cast<f32x4u *>(reinterpret_cast<char *>(sslot) + tid * 16) = sreg;
    };

    // the epilogue's units: a thread owns (row, 8 adjacent channels) pieces of the tile; their residual pieces are fetched while the
    // LAST stage computes (the registers of "the next stage's data" are free then)
    constexpr int EPI_ROWS = (ROWS == 256 && BN == 128) ? 128 : ROWS;      // rows per epilogue pass (the LDS tile of 256 x 132 floats does not fit)
    constexpr int UNITS = EPI_ROWS * (BN / 8) / NTHR;  // units per thread and pass
    constexpr int UR = NTHR / (BN / 8);
    const int cg = tid % (BN / 8);                   // this thread's 8-channel group: the same for all its units
    const int urow0 = tid / (BN / 8);                // ... in rows urow0 + k * UR of the tile
    constexpr int UC = UNITS > 4 ? 4 : UNITS;        // units whose residual pieces are in registers at a time (128 columns: two halves,
    constexpr bool EARLY = UNITS <= 4 && ROWS == 128;  // fetched in the epilogue itself: 64 more live registers cost that kernel a wave per SIMD;
                                                     // the 8-wave form has 128 registers per lane and 16 waves per CU to hide the fetch behind)
    f16x8 rh[UC], rl[UC];
    bool res_issued = false;
    auto issue_residual = [&](int k0 = 0, int pass = 0) {
#pragma unroll
        for (int kc = 0; kc < UC; ++kc) {
            const int k = k0 + kc;
            const int row = tile * ROWS + pass * EPI_ROWS + urow0 + k * UR;
            const int rowc = row < p.n_out ? row : p.n_out - 1;
            const char *rp = reinterpret_cast<const char *>(p.residual + (size_t)rowc * p.res_ld) + ((cg >> 2) << 7) + ((cg & 3) << 4);
            rh[kc] = *reinterpret_cast<const f16x8 *>(rp);
            rl[kc] = *reinterpret_cast<const f16x8 *>(rp + 64);
        }
        res_issued = true;
    };

    Stage cur = {0, 0, 0, -1};
    while (cur.grp < 3 && !((wg_mask >> (9 * cur.grp)) & 0x1ffu)) ++cur.grp;
    bool have = cur.grp < 3, cur_win = true, cur_grp = true;
    if (have) {
        bool nw;
        have = advance(cur, nw);            // the first batch of the first group with a tap (bt: -1 -> first live batch)
    }
    if (have) {
        issue_window(cur, true);
        issue_weights(cur);
    }
    while (have) {
        Stage nxt = cur;
        bool nxt_win = false;
        const bool more = advance(nxt, nxt_win);
        const bool nxt_grp = more && nxt.grp != cur.grp;
        __syncthreads();                                                   // every wave is done with the previous stage's LDS
        write_weights(cur);
        if (cur_win) write_window(cur_grp);
        __syncthreads();
        if (more) {                                                        // the next stage's loads fly under this stage's MFMAs
            if (nxt_win) issue_window(nxt, nxt_grp);
            issue_weights(nxt);
        } else if (EARLY && p.residual) {
            issue_residual();                                              // ... and under the last stage's, the epilogue's residual pieces
        }
        // ---- this stage's taps, from LDS
        {
            const int n_stage = n_list[cur.grp] - cur.base < WIN ? n_list[cur.grp] - cur.base : WIN;
            uint32_t rem = batch_mask(cur.grp, cur.bt);
            const int tl0 = TB * cur.bt;                                   // first tap of the batch within the group
            const uint32_t so0 = (my_mask[0] >> (9 * cur.grp + tl0)), so1 = (my_mask[1] >> (9 * cur.grp + tl0));
            if (CPD_GC_ABLATE & 262144) rem = 0;                          // diagnostic builds: no fragment reads, no MFMAs
            while (rem) {
                const int tt = __builtin_ctz(rem);
                rem &= rem - 1u;
                const uint32_t on = ((so0 >> tt) & 1u) | (((so1 >> tt) & 1u) << 1);
                if (!on) continue;
                typename S::frag a[MS][NP];
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    if ((on >> s) & 1u) {
                        const uint32_t sl = sslot[(tl0 + tt) * ROWS + wave * 32 + 16 * s + r];
                        uint32_t w = sl - (uint32_t)cur.base;              // 0xffff (no neighbour) and rows of other passes land beyond the window:
                        w = w < (uint32_t)n_stage ? w : (uint32_t)WIN;         // the zero row
                        const char *src = swin + g * PLANE + w * 16;
                        a[s][0] = *reinterpret_cast<const typename S::frag *>(src);
                        a[s][1] = *reinterpret_cast<const typename S::frag *>(src + 4 * PLANE);
                    }
                }
                const char *const sb = sw + tt * (NP * B_IMG);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const char *src = sb + ((g * BN + 16 * nt + r) << 4);
                    typename S::frag b[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const typename S::frag *>(src + q * B_IMG);
#pragma unroll
                    for (int s = 0; s < MS; ++s)
                        if ((on >> s) & 1u) acc[s][nt] = S::mma(a[s], b, acc[s][nt]);
                }
            }
        }
        cur = nxt; cur_win = nxt_win; cur_grp = nxt_grp; have = more;
    }
    if (CPD_GC_ABLATE & 524288) { if (acc[0][0][0] == 123.f) p.out[0] = 1.f; return; }         // diagnostic builds: no epilogue
    // ---- epilogue through LDS: the tile's 128 output rows are CONSECUTIVE rows of `out` (and of the residual): a linear block of
    // 128 * 4 BN bytes. The accumulators go to LDS in row-major order; a thread then owns (row, 8 adjacent channels) units: 32
    // contiguous bytes of accumulators, one 16-byte piece of the residual row's high terms and one of its low terms, two 16-byte
    // stores -- whole cache lines in, whole cache lines out. (The fragment-shaped epilogue of the other kernels reads / writes
    // 2- and 4-byte elements, 8 x 32-byte segments per instruction: 354 of 821 us at two workgroups per CU, tools/rowplan_bench.py.)
    constexpr int LD = BN + 4;                       // floats per tile row in LDS (+4: the transposed writes of the four k-groups hit different banks)
    float *const stile = reinterpret_cast<float *>(smem);
    float sc[8], sh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int col = 8 * cg + q;
        sc[q] = p.scale ? p.scale[col] : 1.f;
        if (p.dsc) sc[q] *= p.dsc[col];
        sh[q] = p.shift ? p.shift[col] : 0.f;
    }
    uint32_t vmax = 0;
#pragma unroll
    for (int pass = 0; pass < ROWS / EPI_ROWS; ++pass) {
        if (p.residual && !(pass == 0 && res_issued)) issue_residual(0, pass);
        __syncthreads();                             // every wave is done with the last stage's LDS (or the previous pass's tile)
        if (ROWS == EPI_ROWS || (wave * 32) / EPI_ROWS == pass) {
#pragma unroll
            for (int s = 0; s < MS; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) stile[((wave * 32) % EPI_ROWS + 16 * s + 4 * g + i) * LD + 16 * nt + r] = acc[s][nt][i];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
            if (UNITS > UC && k == UC && p.residual) issue_residual(UC, pass);   // second half of the residual pieces
            const int lrow = urow0 + k * UR;
            const int row = tile * ROWS + pass * EPI_ROWS + lrow;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stile + lrow * LD + 8 * cg + 4);
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = (q < 4 ? v0[q] : v1[q - 4]) * sc[q] + sh[q];
                if (p.residual) v += (float)rh[k % UC][q] + (float)rl[k % UC][q];   // h + l is exact in fp32
                if (p.relu) v = v > 0.f ? v : 0.f;
                t[q] = v;
                const uint32_t vb = __float_as_uint(v) & 0x7fffffffu;
                vmax = (row < p.n_out && vb > vmax) ? vb : vmax;
            }
            if (row < p.n_out) {
                f16x8 h, l;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    h[q] = (_Float16)t[q];
                    l[q] = (_Float16)(t[q] - (float)h[q]);
                }
                char *op = reinterpret_cast<char *>(p.out + (size_t)row * p.out_ld) + ((cg >> 2) << 7) + ((cg & 3) << 4);
                *reinterpret_cast<f16x8 *>(op) = h;
                *reinterpret_cast<f16x8 *>(op + 64) = l;
            }
        }
    }
    if (p.out_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t t2 = (uint32_t)__shfl_xor((int)vmax, o);
            vmax = t2 > vmax ? t2 : vmax;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t *slot = p.out_absmax + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE;
            if (vmax > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, vmax);
        }
    }
}

// LDS: [one stage's weights | window | slot table]; the epilogue's EPI_ROWS x (BN + 4) float tile aliases the first two
template <int BN, int TB, int ROWS>
struct RowplanLds {
    static constexpr int kStage = TB * SplitF16x2::NP * BN * 64 + 8 * RpGeom<ROWS>::P16 * 16 + 9 * ROWS * 2;
    static constexpr int kTile = ((ROWS == 256 && BN == 128) ? 128 : ROWS) * (BN + 4) * 4;
    static constexpr int kBytes = kStage > kTile ? kStage : kTile;
};
template <int BN, int TB, int ROWS = 128>
__global__ void __launch_bounds__(2 * ROWS) rowplan_conv_f16p_kernel(GcParams p) {
    __shared__ __attribute__((aligned(16))) char smem[RowplanLds<BN, TB, ROWS>::kBytes];
    rowplan_conv_body<BN, TB, ROWS>(p, smem);
}

// Split-bf16 image of the weights: Pb[t][k32][piece][g][n][8], piece = h, m, l of
// W[t_src][ci][co] (optionally the adjoint: tap-flipped and/or transposed source).
__global__ void __launch_bounds__(256) pack_weight_bf16_kernel(const float *__restrict__ w, int kv, int c_in, int c_out, int np,
                                                               int adjoint, int flip, __bf16 *__restrict__ pb) {
    // (c_in, c_out) are those of the conv this image is FOR; the source tensor is [kv][c_in][c_out],
    // or [kv][c_out][c_in] when `adjoint` is set.
    const int k32 = c_in >> 5;
    const size_t total = (size_t)kv * k32 * 4 * np * 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 7);
    size_t rest = i >> 3;
    const int n = (int)(rest % np); rest /= np;
    const int g = (int)(rest & 3); rest >>= 2;
    const int kk = (int)(rest % k32);
    const int t = (int)(rest / k32);
    const int ch = kk * 32 + g * 8 + q;
    const int ts = flip ? kv - 1 - t : t;
    float v = 0.f;
    if (n < c_out) v = adjoint ? w[((size_t)ts * c_out + n) * c_in + ch] : w[((size_t)ts * c_in + ch) * c_out + n];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const size_t blk = ((size_t)t * k32 + kk) * 3;       // (tap, k32) block: 3 pieces x 4 g x np x 8
    const size_t off = ((size_t)g * np + n) * 8 + q;
    pb[(blk + 0) * 4 * np * 8 + off] = h;
    pb[(blk + 1) * 4 * np * 8 + off] = m;
    pb[(blk + 2) * 4 * np * 8 + off] = l;
}


// f16x2 image. Step 1: per output column n, the exponent e with max_t,ci |W[t][ci][n]| * 2^e in [2^13, 2^14) (so the
// largest weight uses fp16's top binades, the low term of every weight down to 2^-11 of it stays a NORMAL fp16 number,
// and nothing can overflow); dsc[n] = 2^-e is what the epilogue multiplies back (exact). One block per column.
__global__ void __launch_bounds__(256) weight_col_scale_kernel(const float *__restrict__ w, int kv, int c_in, int c_out, int adjoint,
                                                               float *__restrict__ dsc) {
    // (c_in, c_out) are those of the conv the image is FOR; the source is [kv][c_in][c_out], or [kv][c_out][c_in] if adjoint
    const int n = blockIdx.x;
    float m = 0.f;
    if (n < c_out) {
        const int per = kv * c_in;
        for (int i = threadIdx.x; i < per; i += 256) {
            const int t = i / c_in, ch = i - t * c_in;
            const float v = adjoint ? w[((size_t)t * c_out + n) * c_in + ch] : w[((size_t)t * c_in + ch) * c_out + n];
            m = fmaxf(m, fabsf(v));                           // NaN weights: fmaxf drops them, the products still carry them
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);                            // mx = f * 2^ex, f in [0.5, 1)
            e = 14 - ex;
            e = e > 110 ? 110 : (e < -110 ? -110 : e);
        }
        dsc[n] = ldexpf(1.f, -e);
    }
}
// Step 2: Ph[t][k32][piece][g][n][8], piece = h, l of W * 2^e[n].
__global__ void __launch_bounds__(256) pack_weight_f16_kernel(const float *__restrict__ w, int kv, int c_in, int c_out, int np,
                                                              int adjoint, int flip, const float *__restrict__ dsc,
                                                              _Float16 *__restrict__ ph) {
    const int k32 = c_in >> 5;
    const size_t total = (size_t)kv * k32 * 4 * np * 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 7);
    size_t rest = i >> 3;
    const int n = (int)(rest % np); rest /= np;
    const int g = (int)(rest & 3); rest >>= 2;
    const int kk = (int)(rest % k32);
    const int t = (int)(rest / k32);
    const int ch = kk * 32 + g * 8 + q;
    const int ts = flip ? kv - 1 - t : t;
    float v = 0.f;
    if (n < c_out) v = adjoint ? w[((size_t)ts * c_out + n) * c_in + ch] : w[((size_t)ts * c_in + ch) * c_out + n];
    v = v / dsc[n];                                          // * 2^e, exact
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t blk = ((size_t)t * k32 + kk) * 2;          // (tap, k32) block: 2 pieces x 4 g x np x 8
    const size_t off = ((size_t)g * np + n) * 8 + q;
    ph[(blk + 0) * 4 * np * 8 + off] = h;
    ph[(blk + 1) * 4 * np * 8 + off] = l;
}


// Ph16[t][g][n][hi 4 | lo 4] of W[t][4g + q][n] * 2^e[n] (c_in = 16; (c_in, c_out) are those of the conv the image is FOR, the
// source is [kv][c_in][c_out], or [kv][c_out][c_in] when `adjoint`, tap-flipped when `flip`: as pack_weight_f16_kernel)
__global__ void __launch_bounds__(256) pack_weight_h16_kernel(const float *__restrict__ w, int kv, int c_out, int np, int adjoint, int flip,
                                                              const float *__restrict__ dsc, _Float16 *__restrict__ ph) {
    const size_t total = (size_t)kv * 4 * np * 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 3);
    size_t rest = i >> 2;
    const int n = (int)(rest % np); rest /= np;
    const int g = (int)(rest & 3);
    const int t = (int)(rest >> 2);
    const int ch = 4 * g + q, ts = flip ? kv - 1 - t : t;
    float v = 0.f;
    if (n < c_out) v = (adjoint ? w[((size_t)ts * c_out + n) * 16 + ch] : w[((size_t)ts * 16 + ch) * c_out + n]) / dsc[n];       // * 2^e, exact
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t slot = (((size_t)t * 4 + g) * np + n) * 8;
    ph[slot + q] = h;
    ph[slot + 4 + q] = l;
}

__global__ void __launch_bounds__(256) pack_weight_kernel(const float *__restrict__ w, int kv, int c_in, int c_out, int kc,
                                                          int np, float *__restrict__ packed) {
    // packed[(((t*KC + kc)*4 + g)*NP + n)*4 + q] = W[t][kc*16 + 4*g + q][n]   (zero padded)
    size_t total = (size_t)kv * kc * 4 * np * 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int q = (int)(i & 3);
    size_t rest = i >> 2;
    int n = (int)(rest % np);
    rest /= np;
    int g = (int)(rest & 3);
    rest >>= 2;
    int k = (int)(rest % kc);
    int t = (int)(rest / kc);
    int ch = k * 16 + 4 * g + q;
    float v = 0.f;
    if (ch < c_in && n < c_out) v = w[((size_t)t * c_in + ch) * c_out + n];
    packed[i] = v;
}

// Packed image of the ADJOINT weights used by the input gradient: Wd[t'][co][ci] = W[t][ci][co] with
// t = kv-1-t' when flip is set (SubM / stride-1 convs reuse their own rulebook) else t = t'.
__global__ void __launch_bounds__(256) pack_weight_adjoint_kernel(const float *__restrict__ w, int kv, int c_in, int c_out,
                                                                  int flip, int kc_o, int np_o, float *__restrict__ packed) {
    // adjoint conv has c_in' = c_out, c_out' = c_in ; kc_o = ceil(c_out/16), np_o = 16*ceil(c_in/16)
    size_t total = (size_t)kv * kc_o * 4 * np_o * 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int q = (int)(i & 3);
    size_t rest = i >> 2;
    int n = (int)(rest % np_o);
    rest /= np_o;
    int g = (int)(rest & 3);
    rest >>= 2;
    int k = (int)(rest % kc_o);
    int tp = (int)(rest / kc_o);
    int ch = k * 16 + 4 * g + q;             // adjoint input channel  = original output channel
    int t = flip ? kv - 1 - tp : tp;
    float v = 0.f;
    if (ch < c_out && n < c_in) v = w[((size_t)t * c_in + n) * c_out + ch];
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
static gc_kernel_t pick_h16(int ms, int nt) {           // 16-channel fp16-pair input: c_out = 16 or 32
    if (ms == 1 && nt == 1) return gather_conv_kernel<1, 1, true, true>;
    if (ms == 2 && nt == 1) return gather_conv_kernel<2, 1, true, true>;
    if (ms == 1 && nt == 2) return gather_conv_kernel<1, 2, true, true>;
    if (ms == 2 && nt == 2) return gather_conv_kernel<2, 2, true, true>;
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
static gc_kernel_t pick_tile(int bm, int bn) {
    if (bm == 128 && bn == 128) return tile_conv_kernel<128, 128>;
    if (bm == 64 && bn == 128) return tile_conv_kernel<64, 128>;
    if (bm == 128 && bn == 64) return tile_conv_kernel<128, 64>;
    if (bm == 64 && bn == 64) return tile_conv_kernel<64, 64>;
    return nullptr;
}

// Wave tile choice (measured, tools/sweep_tiles.py on MI355X): per-wave efficiency grows with the
// tile (B pieces reused across MS row sub-tiles, A pieces across NT column tiles) but fp32 MFMA
// only needs one wave per SIMD, so what matters first is having >= ~4 wave tiles per SIMD (4096
// items) to keep 256 CUs x 4 SIMDs evenly loaded; take the largest tile that still gives that.
static void choose_wave_tile(int n_out, int ntot, int *ms_out, int *nt_out) {
    static const int cand[][2] = {{2, 4}, {2, 5}, {2, 2}, {1, 4}, {1, 5}, {1, 2}, {2, 1}, {1, 1}};
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

// Workgroup tile for the dense path (measured): 64 x 128 (else 64 x 64) once there are >= 2048
// workgroups (8 per CU); below that the wave kernel's finer items balance the chip better.
static void choose_wg_tile(int n_out, int c_in, int c_out, int *bm_out, int *bn_out) {
    *bm_out = *bn_out = 0;
    if (c_in % 32 || c_out % 64) return;
    const int bn = (c_out % 128 == 0) ? 128 : 64;
    long long items = (long long)((n_out + 63) / 64) * (c_out / bn);
    if (items >= 2048) { *bm_out = 64; *bn_out = bn; }
}

struct GcPlan {
    int use_wg;     // 0: gather_conv_kernel<a,b,vec>, 1: tile_conv_kernel<a,b>, 2: split workgroup kernel, 3: split row-wave kernel
    int a, b, vec;  // (bm,bn) or (ms,nt)
    int math;       // split kernels: 1 = bf16x3, 2 = f16x2
};

// which split arithmetic the flags (and, when tuning, CPD_GC_BF16X3 = 0 | 1 | 2) ask for: 0 none, 1 bf16x3, 2 f16x2
static int split_math(int flags, bool tn) {
    int m = (flags & CPD_GC_F16X2) ? 2 : ((flags & CPD_GC_BF16X3) ? 1 : 0);
    if (const char *e = cpd_knob(tn, "CPD_GC_BF16X3")) m = atoi(e);
    return m < 0 || m > 2 ? 0 : m;
}

static GcPlan plan(int n_out, int c_in, int c_out, int in_ld, const void *in, int flags) {
    GcPlan pl;
    const bool tn = cpd_tuning();
    const int ntot = (c_out + 15) / 16;
    pl.vec = (c_in % 16 == 0) && (in_ld % 4 == 0) && (((uintptr_t)in & 15) == 0);
    pl.use_wg = 0;
    // split path (CPD_GC_BF16X3 / CPD_GC_F16X2): needs whole 32-channel stages
    pl.math = split_math(flags, tn);
    const int allow_bf16 = pl.math != 0;
    long long bf16_min_wgs = 600;       // > 2 workgroups per CU, else the narrower/shorter tile (measured: single-frame bench, train step)
    if (const char *e = cpd_knob(tn, "CPD_GC_BF16_MIN")) bf16_min_wgs = atoll(e);
    int dense_rowwave = 0;
    if (const char *e = cpd_knob(tn, "CPD_GC_DENSE_ROWWAVE")) dense_rowwave = atoi(e);
    if (allow_bf16 && (!(flags & 1) || dense_rowwave) && pl.vec && c_in % 32 == 0 && c_out % 32 == 0) {     // sparse layers
        const int bn = c_out % 128 == 0 ? 128 : (c_out % 64 == 0 ? 64 : 32);
        int force_bn = 0;
        if (const char *e = cpd_knob(tn, "CPD_GC_ROWWAVE_BN")) force_bn = atoi(e);
        if ((force_bn == 64 || force_bn == 128) && c_out % force_bn == 0) { pl.use_wg = 3; pl.a = 128; pl.b = force_bn; return pl; }
        // widest column tile that still gives every CU a workgroup: a narrower tile re-gathers the rows once per
        // column tile, which small layers (the 1/4 and 1/8 stages of a single frame) can afford; below that, the
        // narrowest tile as long as it covers half the chip (sweep: train step and single-frame bench, +-1%)
        long long rw_min = 256, rw_floor = 128;
        if (const char *e = cpd_knob(tn, "CPD_GC_ROWWAVE_MIN")) rw_min = atoll(e);
        if (const char *e = cpd_knob(tn, "CPD_GC_ROWWAVE_FLOOR")) rw_floor = atoll(e);
        const long long row_tiles = (n_out + 127) / 128;
        if (row_tiles * (c_out / bn) >= rw_min) {
            pl.use_wg = 3; pl.a = 128; pl.b = bn;
            if (pl.math == 2 && (flags & CPD_GC_IN_PAIRS)) {
                // fp16-pair rows: wide workgroups (256 rows / 8 waves; 192 / 6 at 128 columns) for the column tiles in CPD_GC_RW8 (a sum of
                // widths) once the layer has >= CPD_GC_RW8_MIN of them
                int widths = CPD_RW8_DEFAULT;
                long long min_wgs = 1024;
                if (const char *e = cpd_knob(tn, "CPD_GC_RW8")) widths = atoi(e);
                if (const char *e = cpd_knob(tn, "CPD_GC_RW8_MIN")) min_wgs = atoll(e);
                const int wide = 32 * CPD_RW_WIDE_WV(bn);
                if ((widths & bn) && (long long)((n_out + wide - 1) / wide) * (c_out / bn) >= min_wgs) pl.a = wide;
            }
            return pl;
        }
        int small_rows = 1;                     // 64-row workgroups before narrower column tiles
        if (const char *e = cpd_knob(tn, "CPD_GC_ROWWAVE_64")) small_rows = atoi(e);
        const long long row_tiles64 = (n_out + 63) / 64;
        if (small_rows && row_tiles64 * (c_out / bn) >= rw_min) {
            pl.use_wg = 3; pl.a = 64; pl.b = bn;
            return pl;
        }
        for (int b = bn >> 1; b >= 32; b >>= 1) {
            if ((small_rows ? row_tiles64 : row_tiles) * (c_out / b) >= rw_min) {
                pl.use_wg = 3; pl.a = small_rows ? 64 : 128; pl.b = b;
                return pl;
            }
        }
        if ((small_rows ? row_tiles64 : row_tiles) * (c_out / 32) >= rw_floor || (flags & CPD_GC_IN_PAIRS)) {   // (fp16-pair rows: only this kernel reads them)
            pl.use_wg = 3; pl.a = small_rows ? 64 : 128; pl.b = 32;
            return pl;
        }
    }
    if (allow_bf16 && (flags & 1) && pl.vec && c_in % 32 == 0 && c_out % 64 == 0) {
        int bn = c_out % 128 == 0 ? 128 : 64;
        if (const char *e = cpd_knob(tn, "CPD_GC_BF16_BN")) { if (atoi(e) == 64) bn = 64; }
        if ((long long)((n_out + 127) / 128) * (c_out / bn) >= bf16_min_wgs) {
            pl.use_wg = 2; pl.a = 128; pl.b = bn;
            return pl;
        }
        long long min64 = 256;                              // 64-row tiles when 128-row tiling would leave CUs idle
        if (const char *e = cpd_knob(tn, "CPD_GC_BF16_MIN64")) min64 = atoll(e);
        if ((long long)((n_out + 63) / 64) * (c_out / bn) >= min64) {
            pl.use_wg = 2; pl.a = 64; pl.b = bn;
            return pl;
        }
        // K-heavy GEMMs with few rows (the RoI head's 27648 -> 256 shared FC on a few thousand RoIs: 864 stages): the stage split
        // (rowwave_split, up to 8 parts) makes the workgroups the row tiles cannot -- on the wave kernel this layer ran at 62 TFLOP/s
        const long long parts = c_in / 32 / 16 > 8 ? 8 : c_in / 32 / 16;
        if (c_in >= 2048 && (long long)((n_out + 127) / 128) * (c_out / bn) * parts >= min64) {
            pl.use_wg = 2; pl.a = 128; pl.b = bn;
            return pl;
        }
    }
    int force_wg = -1;
    if (const char *e = cpd_knob(tn, "CPD_GC_WG")) force_wg = atoi(e);
    if (((flags & 1) && force_wg != 0) || force_wg > 0) {
        int bm, bn;
        choose_wg_tile(n_out, c_in, c_out, &bm, &bn);
        if (force_wg > 0 && !bm && c_in % 32 == 0 && c_out % 64 == 0) { bm = 64; bn = 64; }
        if (const char *e = cpd_knob(tn, "CPD_GC_BM")) { int v = atoi(e); if ((v == 64 || v == 128) && bm) bm = v; }
        if (const char *e = cpd_knob(tn, "CPD_GC_BN")) { int v = atoi(e); if ((v == 64 || v == 128) && bn && c_out % v == 0) bn = v; }
        if (bm && bn && pl.vec) { pl.use_wg = 1; pl.a = bm; pl.b = bn; return pl; }
    }
    choose_wave_tile(n_out, ntot, &pl.a, &pl.b);
    if (const char *e = cpd_knob(tn, "CPD_GC_MS")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) pl.a = v; }
    if (const char *e = cpd_knob(tn, "CPD_GC_NT")) { int v = atoi(e); if ((v == 1 || v == 2 || v == 4 || v == 5 || v == 8) && ntot % v == 0) pl.b = v; }
    return pl;
}

}  // namespace

static size_t packed_f32_floats(int kv, int c_in, int c_out) {
    return (size_t)kv * ((c_in + 15) / 16) * 4 * (((c_out + 15) / 16) * 16) * 4;
}
// the split-bf16 image exists when the conv has whole 32-channel stages (6 bytes per weight)
static size_t packed_bf16_floats(int kv, int c_in, int c_out) {
    if (c_in % 32) return 0;
    return (size_t)kv * (c_in / 32) * 3 * 4 * (((c_out + 15) / 16) * 16) * 8 / 2;
}
// ... and so does the f16x2 image (4 bytes per weight) followed by its np per-column descale floats
static size_t packed_f16_image_floats(int kv, int c_in, int c_out) {
    if (c_in % 32) return 0;
    return (size_t)kv * (c_in / 32) * 2 * 4 * (((c_out + 15) / 16) * 16) * 8 / 2;
}
static size_t packed_f16_floats(int kv, int c_in, int c_out) {
    if (c_in % 32) return 0;
    return packed_f16_image_floats(kv, c_in, c_out) + (size_t)(((c_out + 15) / 16) * 16);
}
// 16-channel layers (level 1 of the backbone): a split-fp16 image for the K = 16 MFMA, Ph16[t][g][n][hi 4 | lo 4] (16 B per
// (tap, k-group, column): the four high terms of channels 4g..4g+3, then the four low terms), followed by its np descale floats.
// Forward images only (cpd_pack_weight); it sits at the END of the buffer, so the other images' offsets do not move.
static size_t packed_h16_image_floats(int kv, int c_in, int c_out) {
    if (c_in != 16) return 0;
    return (size_t)kv * 4 * (((c_out + 15) / 16) * 16) * 4;
}
static size_t packed_h16_floats(int kv, int c_in, int c_out) {
    if (c_in != 16) return 0;
    return packed_h16_image_floats(kv, c_in, c_out) + (size_t)(((c_out + 15) / 16) * 16);
}
extern "C" size_t cpd_packed_weight_floats(int kv, int c_in, int c_out) {
    if (kv <= 0 || c_in <= 0 || c_out <= 0) return 0;
    return packed_f32_floats(kv, c_in, c_out) + packed_bf16_floats(kv, c_in, c_out) + packed_f16_floats(kv, c_in, c_out) +
           packed_h16_floats(kv, c_in, c_out);
}
static const float *packed_h16_ptr(const float *packed, int kv, int c_in, int c_out) {
    return packed + packed_f32_floats(kv, c_in, c_out) + packed_bf16_floats(kv, c_in, c_out) + packed_f16_floats(kv, c_in, c_out);
}
// where the images of one packed buffer start
static const float *packed_bf16_ptr(const float *packed, int kv, int c_in, int c_out) { return packed + packed_f32_floats(kv, c_in, c_out); }
static const float *packed_f16_ptr(const float *packed, int kv, int c_in, int c_out) {
    return packed + packed_f32_floats(kv, c_in, c_out) + packed_bf16_floats(kv, c_in, c_out);
}
static const float *packed_dsc_ptr(const float *packed, int kv, int c_in, int c_out) {
    return packed_f16_ptr(packed, kv, c_in, c_out) + packed_f16_image_floats(kv, c_in, c_out);
}
static void pack_bf16_image(const float *w, int kv, int c_in, int c_out, int adjoint, int flip, float *packed, hipStream_t s) {
    if (!packed_bf16_floats(kv, c_in, c_out)) return;
    const int np = ((c_out + 15) / 16) * 16;
    const size_t total = (size_t)kv * (c_in / 32) * 4 * np * 8;
    pack_weight_bf16_kernel<<<cpd_div_up((long long)total, 256), 256, 0, s>>>(
        w, kv, c_in, c_out, np, adjoint, flip, reinterpret_cast<__bf16 *>(packed + packed_f32_floats(kv, c_in, c_out)));
    float *dsc = const_cast<float *>(packed_dsc_ptr(packed, kv, c_in, c_out));
    weight_col_scale_kernel<<<np, 256, 0, s>>>(w, kv, c_in, c_out, adjoint, dsc);
    pack_weight_f16_kernel<<<cpd_div_up((long long)total, 256), 256, 0, s>>>(
        w, kv, c_in, c_out, np, adjoint, flip, dsc, reinterpret_cast<_Float16 *>(const_cast<float *>(packed_f16_ptr(packed, kv, c_in, c_out))));
}

static void pack_h16_image(const float *w, int kv, int c_in, int c_out, int adjoint, int flip, float *packed, hipStream_t s) {
    if (!packed_h16_floats(kv, c_in, c_out)) return;           // (c_in, c_out): of the conv the image is for
    const int np = ((c_out + 15) / 16) * 16;
    float *img = const_cast<float *>(packed_h16_ptr(packed, kv, c_in, c_out));
    float *dsc = img + packed_h16_image_floats(kv, c_in, c_out);
    weight_col_scale_kernel<<<np, 256, 0, s>>>(w, kv, c_in, c_out, adjoint, dsc);
    const size_t n16 = (size_t)kv * 4 * np * 4;
    pack_weight_h16_kernel<<<cpd_div_up((long long)n16, 256), 256, 0, s>>>(w, kv, c_out, np, adjoint, flip, dsc, reinterpret_cast<_Float16 *>(img));
}

extern "C" int cpd_pack_weight(const float *w_kio, int kv, int c_in, int c_out, float *packed, cpd_stream_t stream) {
    if (!w_kio || !packed || kv <= 0 || c_in <= 0 || c_out <= 0) return CPD_ERR_ARG;
    int kc = (c_in + 15) / 16, np = ((c_out + 15) / 16) * 16;
    size_t total = packed_f32_floats(kv, c_in, c_out);
    pack_weight_kernel<<<cpd_div_up((long long)total, 256), 256, 0, cpd_s(stream)>>>(w_kio, kv, c_in, c_out, kc, np, packed);
    pack_bf16_image(w_kio, kv, c_in, c_out, 0, 0, packed, cpd_s(stream));
    pack_h16_image(w_kio, kv, c_in, c_out, 0, 0, packed, cpd_s(stream));
    return cpd_check_launch();
}

extern "C" int cpd_pack_weight_adjoint(const float *w_kio, int kv, int c_in, int c_out, int flip_taps, float *packed,
                                       cpd_stream_t stream) {
    if (!w_kio || !packed || kv <= 0 || c_in <= 0 || c_out <= 0) return CPD_ERR_ARG;
    int kc_o = (c_out + 15) / 16, np_o = ((c_in + 15) / 16) * 16;
    size_t total = packed_f32_floats(kv, c_out, c_in);
    pack_weight_adjoint_kernel<<<cpd_div_up((long long)total, 256), 256, 0, cpd_s(stream)>>>(w_kio, kv, c_in, c_out, flip_taps,
                                                                                            kc_o, np_o, packed);
    pack_bf16_image(w_kio, kv, c_out, c_in, 1, flip_taps, packed, cpd_s(stream));
    pack_h16_image(w_kio, kv, c_out, c_in, 1, flip_taps, packed, cpd_s(stream));
    return cpd_check_launch();
}

// ---- all packed images of a model in three launches (the train step rewrites ~70 of them after every optimiser step) ----
// A job = one packed buffer to rebuild from one [kv][c_in][c_out] tensor (forward image, or -- adjoint -- the image of the
// input-gradient conv, tap-flipped or not). The device table carries, per job, its first block in each of the three grids.
namespace {
struct PackJobDev {
    const float *w;
    float *packed;
    int kv, c_in, c_out, adjoint, flip;     // c_in / c_out of the SOURCE tensor
    int ci, co, kc, np;                     // of the conv the image is for; kc = 16-channel chunks, np = padded columns
    int split;                              // has the split images (ci % 32 == 0)
    int h16;                                // has the K = 16 split-fp16 image (forward image of a 16-channel layer)
    unsigned b_f32, b_scale, b_split;       // first block of the job in the fp32-image / column-scale / split-image grids
    size_t off_bf16, off_f16, off_dsc;      // float offsets of the images inside `packed`
    size_t off_h16, off_dsc16;
};
template <unsigned PackJobDev::*FIRST>
__device__ __forceinline__ int pack_job_of(const PackJobDev *jobs, int n_jobs, unsigned block) {
    int lo = 0, hi = n_jobs - 1;            // last job whose first block <= block (jobs without blocks in this grid share a start)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].*FIRST <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void __launch_bounds__(256) pack_batch_f32_kernel(const PackJobDev *__restrict__ jobs, int n_jobs) {
    const PackJobDev j = jobs[pack_job_of<&PackJobDev::b_f32>(jobs, n_jobs, blockIdx.x)];
    const size_t total = (size_t)j.kv * j.kc * 4 * j.np * 4;
    const size_t i = (size_t)(blockIdx.x - j.b_f32) * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 3);
    size_t rest = i >> 2;
    const int n = (int)(rest % j.np); rest /= j.np;
    const int g = (int)(rest & 3); rest >>= 2;
    const int k = (int)(rest % j.kc);
    const int tp = (int)(rest / j.kc);
    const int ch = k * 16 + 4 * g + q;
    float v = 0.f;
    if (ch < j.ci && n < j.co) {
        const int t = j.flip ? j.kv - 1 - tp : tp;
        v = j.adjoint ? j.w[((size_t)t * j.c_in + n) * j.c_out + ch] : j.w[((size_t)t * j.c_in + ch) * j.c_out + n];
    }
    j.packed[i] = v;
}
// one block per (job, padded column): dsc = 2^-e, as weight_col_scale_kernel
__global__ void __launch_bounds__(256) pack_batch_scale_kernel(const PackJobDev *__restrict__ jobs, int n_jobs) {
    const PackJobDev j = jobs[pack_job_of<&PackJobDev::b_scale>(jobs, n_jobs, blockIdx.x)];
    const int n = (int)(blockIdx.x - j.b_scale);
    if ((!j.split && !j.h16) || n >= j.np) return;
    float m = 0.f;
    if (n < j.co) {
        const int per = j.kv * j.ci;
        for (int i = threadIdx.x; i < per; i += 256) {
            const int t = i / j.ci, ch = i - t * j.ci;
            const float v = j.adjoint ? j.w[((size_t)t * j.c_in + n) * j.c_out + ch] : j.w[((size_t)t * j.c_in + ch) * j.c_out + n];
            m = fmaxf(m, fabsf(v));
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            e = 14 - ex;
            e = e > 110 ? 110 : (e < -110 ? -110 : e);
        }
        j.packed[(j.h16 ? j.off_dsc16 : j.off_dsc) + n] = ldexpf(1.f, -e);
    }
}
// images: bit 0 = the split-bf16 image, bit 1 = the split-fp16 image (needs the column scales of the launch before)
__global__ void __launch_bounds__(256) pack_batch_split_kernel(const PackJobDev *__restrict__ jobs, int n_jobs, int images) {
    const PackJobDev j = jobs[pack_job_of<&PackJobDev::b_split>(jobs, n_jobs, blockIdx.x)];
    if (j.h16) {                            // Ph16[t][g][n][hi 4 | lo 4], as pack_weight_h16_kernel
        const size_t total16 = (size_t)j.kv * 4 * j.np * 4;
        const size_t e = (size_t)(blockIdx.x - j.b_split) * 256 + threadIdx.x;
        if (!(images & 2) || e >= total16) return;
        const int q = (int)(e & 3);
        size_t rest = e >> 2;
        const int n = (int)(rest % j.np); rest /= j.np;
        const int g = (int)(rest & 3);
        const int t = (int)(rest >> 2);
        const int ch = 4 * g + q, ts = j.flip ? j.kv - 1 - t : t;
        float v = 0.f;
        if (n < j.co) v = (j.adjoint ? j.w[((size_t)ts * j.c_in + n) * j.c_out + ch] : j.w[((size_t)ts * j.c_in + ch) * j.c_out + n]) / j.packed[j.off_dsc16 + n];
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        _Float16 *ph = reinterpret_cast<_Float16 *>(j.packed + j.off_h16);
        const size_t slot = (((size_t)t * 4 + g) * j.np + n) * 8;
        ph[slot + q] = h;
        ph[slot + 4 + q] = l;
        return;
    }
    if (!j.split) return;
    const int k32 = j.ci >> 5;
    const size_t total = (size_t)j.kv * k32 * 4 * j.np * 8;
    const size_t i = (size_t)(blockIdx.x - j.b_split) * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & 7);
    size_t rest = i >> 3;
    const int n = (int)(rest % j.np); rest /= j.np;
    const int g = (int)(rest & 3); rest >>= 2;
    const int kk = (int)(rest % k32);
    const int t = (int)(rest / k32);
    const int ch = kk * 32 + g * 8 + q;
    const int ts = j.flip ? j.kv - 1 - t : t;
    float v = 0.f;
    if (n < j.co) v = j.adjoint ? j.w[((size_t)ts * j.c_in + n) * j.c_out + ch] : j.w[((size_t)ts * j.c_in + ch) * j.c_out + n];
    const size_t off = ((size_t)g * j.np + n) * 8 + q;
    if (images & 1) {
        __bf16 *pb = reinterpret_cast<__bf16 *>(j.packed + j.off_bf16);
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        const size_t blk = ((size_t)t * k32 + kk) * 3;
        pb[(blk + 0) * 4 * j.np * 8 + off] = h;
        pb[(blk + 1) * 4 * j.np * 8 + off] = m;
        pb[(blk + 2) * 4 * j.np * 8 + off] = l;
    }
    if (images & 2) {
        _Float16 *ph = reinterpret_cast<_Float16 *>(j.packed + j.off_f16);
        const float vs = v / j.packed[j.off_dsc + n];                 // * 2^e, exact
        const _Float16 h = (_Float16)vs;
        const _Float16 l = (_Float16)(vs - (float)h);
        const size_t blk = ((size_t)t * k32 + kk) * 2;
        ph[(blk + 0) * 4 * j.np * 8 + off] = h;
        ph[(blk + 1) * 4 * j.np * 8 + off] = l;
    }
}
}  // namespace

extern "C" size_t cpd_pack_batch_table_bytes(int n_jobs) { return n_jobs > 0 ? cpd_align((size_t)n_jobs * sizeof(PackJobDev)) : 0; }

extern "C" int cpd_pack_batch_prepare(const cpd_pack_job *jobs, int n_jobs, void *table, size_t table_bytes, int32_t grid_blocks[3]) {
    if (!jobs || n_jobs <= 0 || !table || !grid_blocks || table_bytes < (size_t)n_jobs * sizeof(PackJobDev)) return CPD_ERR_ARG;
    std::vector<PackJobDev> host((size_t)n_jobs);
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const cpd_pack_job &s = jobs[i];
        if (!s.w || !s.packed || s.kv <= 0 || s.c_in <= 0 || s.c_out <= 0) return CPD_ERR_ARG;
        PackJobDev &d = host[(size_t)i];
        d.w = s.w; d.packed = s.packed; d.kv = s.kv; d.c_in = s.c_in; d.c_out = s.c_out; d.adjoint = s.adjoint != 0; d.flip = s.flip_taps != 0;
        d.ci = d.adjoint ? s.c_out : s.c_in;
        d.co = d.adjoint ? s.c_in : s.c_out;
        d.kc = (d.ci + 15) / 16; d.np = ((d.co + 15) / 16) * 16;
        d.split = d.ci % 32 == 0;
        d.off_bf16 = packed_f32_floats(d.kv, d.ci, d.co);
        d.off_f16 = d.off_bf16 + packed_bf16_floats(d.kv, d.ci, d.co);
        d.off_dsc = d.off_f16 + packed_f16_image_floats(d.kv, d.ci, d.co);
        d.h16 = packed_h16_floats(d.kv, d.ci, d.co) != 0;
        d.off_h16 = d.off_f16 + packed_f16_floats(d.kv, d.ci, d.co);
        d.off_dsc16 = d.off_h16 + packed_h16_image_floats(d.kv, d.ci, d.co);
        d.b_f32 = (unsigned)b0; d.b_scale = (unsigned)b1; d.b_split = (unsigned)b2;
        b0 += (unsigned long long)cpd_div_up((long long)packed_f32_floats(d.kv, d.ci, d.co), 256);
        if (d.split) {
            b1 += (unsigned long long)d.np;
            b2 += (unsigned long long)cpd_div_up((long long)d.kv * (d.ci / 32) * 4 * d.np * 8, 256);
        } else if (d.h16) {
            b1 += (unsigned long long)d.np;
            b2 += (unsigned long long)cpd_div_up((long long)d.kv * 4 * d.np * 4, 256);
        }
        if (b0 >= (1ull << 31) || b2 >= (1ull << 31)) return CPD_ERR_UNSUPPORTED;
    }
    grid_blocks[0] = (int32_t)b0; grid_blocks[1] = (int32_t)b1; grid_blocks[2] = (int32_t)b2;
    if (hipMemcpy(table, host.data(), (size_t)n_jobs * sizeof(PackJobDev), hipMemcpyHostToDevice) != hipSuccess) return CPD_ERR_LAUNCH;
    return CPD_OK;
}

extern "C" int cpd_pack_batch_run(const void *table, int n_jobs, const int32_t grid_blocks[3], int images, cpd_stream_t stream) {
    if (!table || n_jobs <= 0 || !grid_blocks || (images & ~3)) return CPD_ERR_ARG;
    const PackJobDev *jobs = reinterpret_cast<const PackJobDev *>(table);
    hipStream_t s = cpd_s(stream);
    if (grid_blocks[0] > 0) pack_batch_f32_kernel<<<grid_blocks[0], 256, 0, s>>>(jobs, n_jobs);
    if ((images & 2) && grid_blocks[1] > 0) pack_batch_scale_kernel<<<grid_blocks[1], 256, 0, s>>>(jobs, n_jobs);
    if (images && grid_blocks[2] > 0) pack_batch_split_kernel<<<grid_blocks[2], 256, 0, s>>>(jobs, n_jobs, images);
    return cpd_check_launch();
}

extern "C" int cpd_gather_conv_tile(int n_out, int c_in, int c_out, int in_ld, int flags, int *wg, int *a, int *b,
                                    int *vec) {
    if (n_out <= 0 || c_in <= 0 || c_out <= 0 || !wg || !a || !b || !vec) return CPD_ERR_ARG;
    GcPlan pl = plan(n_out, c_in, c_out, in_ld, nullptr, flags);
    *wg = pl.use_wg + ((pl.use_wg >= 2 && pl.math == 2) ? 10 : 0);      // 12 / 13: the f16x2 instantiations of 2 / 3
    *a = pl.a; *b = pl.b; *vec = pl.vec;
    return CPD_OK;
}

// Tap split of a SMALL row-wave launch (one frame, the train step): a workgroup walks its row tile's (tap, 32-channel block) stages
// one after the other -- 108 of them at 128 channels, ~0.9 us each when a CU holds one or two workgroups -- so a layer with fewer
// workgroups than ~2 per CU takes its stage count times a memory round trip whatever its size. With a workspace the taps of a tile
// are dealt to `split` workgroups (blockIdx.y), each writes raw partial sums, and split_finish_kernel adds them in a fixed order
// and runs the epilogue. -> number of parts (1 = no split).
static int rowwave_split(const GcPlan &pl, int n_out, int c_in, int c_out, int kv) {
    if (pl.use_wg != 3 && pl.use_wg != 2) return 1;
    const bool tn = cpd_tuning();
    if (pl.use_wg == 2) {                            // the workgroup (tile) kernel: contiguous shares of its kv * c_in / 32 stages
        int on = 1;
        if (const char *e = cpd_knob(tn, "CPD_GC_SPLIT_TILE")) on = atoi(e);
        const long long wgs = (long long)((n_out + pl.a - 1) / pl.a) * (c_out / pl.b);
        const int stages = kv * (c_in / 32);
        if (!on || wgs >= 600 || stages < 32) return 1;
        long long s = on > 1 ? on : 1200 / (wgs > 0 ? wgs : 1);
        if (s > stages / 16) s = stages / 16;        // at least 16 stages per part
        const int cap = stages >= 256 ? 8 : 4;
        return s < 2 ? 1 : (s > cap ? cap : (int)s);
    }
    if (const char *e = cpd_knob(tn, "CPD_GC_SPLIT")) { const int v = atoi(e); return v < 1 ? 1 : (v > 8 ? 8 : v); }
    const long long wgs = (long long)((n_out + pl.a - 1) / pl.a) * (c_out / pl.b);
    const int stages = kv * (c_in / 32);
    int min_stages = 16;                             // (16 vs 100, same box: one frame 3.23 vs 3.27 ms, train step 9.85 vs 10.2 ms)
    if (const char *e = cpd_knob(tn, "CPD_GC_SPLIT_STAGES")) min_stages = atoi(e);
    if (wgs >= 600 || stages < min_stages || kv < 4) return 1;
    long long s = 1200 / (wgs > 0 ? wgs : 1);
    if (s > kv / 4) s = kv / 4;                      // at least ~4 taps per part
    return s < 2 ? 1 : (s > 4 ? 4 : (int)s);
}

static int rowwave_finish(const GcParams &p, hipStream_t hs) {      // after a row-wave launch: the second half of a tap split
    if (p.split > 1) {
        cpd_launch_log_note("split_finish_kernel");
        const long long threads = (long long)p.n_out * (p.c_out / 4);
        hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)cpd_div_up(threads, 256)), dim3(256), 0, hs, p);
    }
    return cpd_check_launch();
}

static int gather_conv_impl(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                            const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale, const float *shift,
                            const float *residual, int res_ld, int relu, float *out, int out_ld,
                            const int32_t *out_row_map, int out_col_group, int flags, const uint32_t *in_absmax, uint32_t *out_absmax,
                            cpd_stream_t stream, float *part = nullptr, size_t part_bytes = 0) {
    if (n_out == 0 && n_in >= 0 && c_in > 0 && c_out > 0 && kv > 0) return CPD_OK;   // an empty site set is a valid (empty) result
    if (!in || !packed_w || !out || n_in < 0 || n_out < 0 || c_in <= 0 || c_out <= 0 || kv <= 0 || in_ld < c_in ||
        (residual && res_ld < c_out) || (!nbr && kv != 1) || (tapmask && kv > 32) || out_col_group < 0 || (out_col_group > 0 && !out_row_map) ||
        out_ld < (out_col_group > 0 ? (out_col_group < c_out ? out_col_group : c_out) : c_out))
        return CPD_ERR_ARG;
    if (n_out == 0) return CPD_OK;
    // fp16-pair rows: whole 32-channel blocks, f16x2 arithmetic, no pre-scaling (a guarded re-run works on fp32 rows)
    const bool in16 = (flags & CPD_GC_IN_PAIRS) && c_in == 16;      // 16-channel pair rows: the wave kernel's K = 16 form
    if (in16 && (!(flags & CPD_GC_F16X2) || (flags & CPD_GC_DENSE) || (c_out != 16 && c_out != 32) || in_ld % 4 || (((uintptr_t)in) & 15) || in_absmax ||
                 (size_t)n_in * in_ld * sizeof(float) >= 0xfffff000ull))
        return CPD_ERR_UNSUPPORTED;
    if ((flags & CPD_GC_IN_PAIRS) && (flags & CPD_GC_DENSE) && !in16) {
        // DENSE pair rows (round 5: the BEV maps between split-fp16 dense layers): the 128 x 128 pair tile kernel (tile_conv_f16p_kernel) --
        // strided conv through its pixel table, 1 x 1 GEMMs, ConvTranspose(k = s) through the row map / column-group scatter. Pair rows in
        // AND out, no residual, no pre-scaling, no stage split; anything else is refused (nothing else reads dense pair rows)
        if (!(flags & CPD_GC_F16X2) || !(flags & CPD_GC_OUT_PAIRS) || (flags & CPD_GC_RES_PAIRS) || residual || in_absmax || c_in % 32 || c_out % 128 ||
            in_ld % 4 || out_ld % 4 || (((uintptr_t)in) & 15) || (((uintptr_t)out) & 15) || (out_col_group && out_col_group % 32) ||
            (size_t)n_in * in_ld * sizeof(float) >= 0xfffff000ull || split_math(flags, cpd_tuning()) != 2)
            return CPD_ERR_UNSUPPORTED;
        GcParams p;
        memset(&p, 0, sizeof p);
        p.in_pairs = 1; p.out_pairs = 1; p.split = 1;
        p.in = in; p.w = packed_w; p.out_absmax = out_absmax; p.nbr = nbr; p.scale = scale; p.shift = shift; p.out = out; p.out_row_map = out_row_map;
        p.in_ld = in_ld; p.c_in = c_in; p.kc = c_in / 16; p.n_in_rows = n_in;
        p.kv = kv; p.n_out = n_out; p.c_out = c_out; p.ntot = c_out / 16; p.np = c_out;
        p.relu = relu; p.out_ld = out_ld; p.col_group = out_col_group; p.n_sub = (n_out + 15) / 16;
        p.wb = packed_f16_ptr(packed_w, kv, c_in, c_out);
        p.dsc = packed_dsc_ptr(packed_w, kv, c_in, c_out);
        p.n_rb = (n_out + 127) / 128; p.n_cb = c_out / 128; p.items = p.n_rb * p.n_cb;
        cpd_launch_log_note("tile_conv_f16p_kernel<128,128>");
        hipLaunchKernelGGL(tile_conv_f16p_kernel, dim3(p.items), dim3(256), 128 * 128 + 2 * 128 * 64 > 32 * 132 * 4 ? 128 * 128 + 2 * 128 * 64 : 32 * 132 * 4,
                           cpd_s(stream), p);
        return cpd_check_launch();
    }
    if ((flags & CPD_GC_IN_PAIRS) && !in16 &&
        (!(flags & CPD_GC_F16X2) || (flags & CPD_GC_DENSE) || c_in % 32 || c_out % 32 || in_ld % 4 || in_absmax || kv > CPD_RW_TAPS ||
         (size_t)n_in * in_ld * sizeof(float) >= 0xfffff000ull))
        return CPD_ERR_UNSUPPORTED;
    if ((flags & CPD_GC_OUT_PAIRS) && ((c_out % 32 && c_out != 16) || out_col_group > 0)) return CPD_ERR_UNSUPPORTED;
    if ((flags & CPD_GC_RES_PAIRS) && (!residual || (c_out % 32 && c_out != 16))) return CPD_ERR_ARG;
    GcParams p;
    p.in_pairs = (flags & CPD_GC_IN_PAIRS) != 0;
    p.out_pairs = (flags & CPD_GC_OUT_PAIRS) ? (c_out == 16 ? 2 : 1) : 0;       // (2: the 16-channel row format)
    p.res_pairs = (flags & CPD_GC_RES_PAIRS) ? (c_out == 16 ? 2 : 1) : 0;
    p.split = 1; p.part = nullptr;
    p.plan_slots = nullptr; p.plan_ulist = nullptr; p.plan_count = nullptr;
    p.in = in; p.w = packed_w; p.wb = nullptr; p.dsc = nullptr; p.in_absmax = in_absmax; p.out_absmax = out_absmax; p.nbr = nbr; p.tapmask = tapmask; p.n_sub = (n_out + 15) / 16; p.scale = scale; p.shift = shift; p.residual = residual;
    p.out = out; p.out_row_map = out_row_map;
    p.in_ld = in_ld; p.c_in = c_in; p.kc = (c_in + 15) / 16; p.n_in_rows = n_in;
    p.kv = kv; p.n_out = n_out; p.c_out = c_out; p.ntot = (c_out + 15) / 16; p.np = p.ntot * 16;
    p.res_ld = res_ld; p.relu = relu; p.out_ld = out_ld; p.col_group = out_col_group;
    GcPlan pl = plan(n_out, c_in, c_out, in_ld, in, flags);
    if (pl.use_wg == 3 && (kv > CPD_RW_TAPS || (size_t)n_in * in_ld * sizeof(float) >= 0xfffff000ull))
        pl = plan(n_out, c_in, c_out, in_ld, in, flags | CPD_GC_DENSE);   // the row-wave kernel keeps its taps in 32-bit sets and reads its rows through a 4 GB buffer resource
    if (pl.use_wg == 0 && kv > 32) {
        // the wave kernel keeps a tile's taps in 32-bit sets and their rulebook columns in a 32-tap LDS table: a larger kernel goes to
        // the workgroup kernels when its channel counts allow, and is refused -- loudly -- when they do not
        pl = plan(n_out, c_in, c_out, in_ld, in, flags | CPD_GC_DENSE);
        if (pl.use_wg == 0) return CPD_ERR_UNSUPPORTED;
    }
    if (in16 && pl.use_wg != 0) return CPD_ERR_UNSUPPORTED;
    if (p.in_pairs && !in16 && (pl.use_wg != 3 || pl.math != 2)) return CPD_ERR_UNSUPPORTED;
    if ((p.out_pairs || p.res_pairs) && pl.use_wg != 3 && pl.use_wg != 0) return CPD_ERR_UNSUPPORTED;   // (the sparse kernels' epilogues)
    p.taps_inner = 1;       // measured (tools/order_probe.py): -6...-8 % on the 32- and 128-channel SubM layers, neutral at 64
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_TAPS_INNER")) p.taps_inner = atoi(e);
    static const bool trace = getenv("CPD_GC_TRACE") != nullptr;    // one line per launch: which kernel a layer got
    if (trace)
        fprintf(stderr, "cpd_gather_conv n_out=%d kv=%d c_in=%d c_out=%d flags=%d masks=%d -> kind=%d tile=(%d,%d)\n", n_out, kv, c_in, c_out,
                flags, tapmask != nullptr, pl.use_wg, pl.a, pl.b);
    if (pl.use_wg == 2 || pl.use_wg == 3) {
        if (pl.math == 2) { p.wb = packed_f16_ptr(packed_w, kv, c_in, c_out); p.dsc = packed_dsc_ptr(packed_w, kv, c_in, c_out); }
        else p.wb = packed_bf16_ptr(packed_w, kv, c_in, c_out);
        p.n_rb = (n_out + pl.a - 1) / pl.a;
        p.n_cb = c_out / pl.b;
        p.items = p.n_rb * p.n_cb;
    }
    hipStream_t hs = cpd_s(stream);
    if (part && (pl.use_wg == 3 || pl.use_wg == 2) && !out_col_group && c_out % 4 == 0 && out_ld % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
        (!residual || (flags & CPD_GC_RES_PAIRS) || (res_ld % 4 == 0 && ((uintptr_t)residual & 15) == 0)) &&
        (!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0)) {
        const int sp = rowwave_split(pl, n_out, c_in, c_out, kv);
        if (sp > 1 && part_bytes >= (size_t)sp * n_out * c_out * sizeof(float)) { p.split = sp; p.part = part; }
    }
    {   // the f16x2 row-wave kernels' LDS epilogue (rowwave_conv_split_body): rows in place, no tap split; pair-row kernels write pair rows
        // (pair residual), fp32-row kernels fp32 rows (fp32 residual) -- the mixed layers keep the shared epilogue
        int epi = 1;                                 // tuning: 0 = the shared fragment-shaped epilogue
        if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_RW_EPI")) epi = atoi(e);
        p.epi_lds = epi && pl.use_wg == 3 && pl.a <= 128 && pl.math == 2 && !out_row_map &&
                    (p.in_pairs ? (p.out_pairs == 1 && (!residual || p.res_pairs == 1)) : (p.out_pairs == 0 && p.res_pairs == 0)) &&
                    !out_col_group && p.split == 1 && out_ld % 4 == 0 && (((uintptr_t)out) & 15) == 0 &&
                    (!residual || (res_ld % 4 == 0 && (((uintptr_t)residual) & 15) == 0));
    }
    if (pl.use_wg == 2)      // the 128 x 128 split tile kernels' LDS epilogue: 16-byte pieces when the operands allow (element accesses otherwise)
        p.epi_lds = out_ld % 4 == 0 && (((uintptr_t)out) & 15) == 0 && (!out_col_group || out_col_group % 4 == 0) &&
                    (!residual || (res_ld % 4 == 0 && (((uintptr_t)residual) & 15) == 0));
    const dim3 grid(p.items, p.split), block(pl.use_wg == 3 && pl.a > 128 ? pl.a * 2 : 256);
    {   // launch log (cpd_launch_log_*): the instantiation this call runs
        char nm[96];
        const bool e = pl.use_wg == 3 && p.epi_lds;      // (the LDS-epilogue instantiations of the row-wave kernels: f16e / f16se / f16pe)
        const char *sc = (pl.math == 2 && in_absmax) ? (e ? "f16se" : "f16s") : (pl.math == 2 ? (p.in_pairs ? (e ? "f16pe" : "f16p") : (e ? "f16e" : "f16")) : "bf16");
        if (pl.use_wg == 3 && pl.a > 128) snprintf(nm, sizeof nm, "rowwave_conv_f16pw_kernel<%d,%d>", pl.b, pl.a / 32);
        else if (pl.use_wg == 3) snprintf(nm, sizeof nm, "rowwave_conv_%s_kernel<%d,%d>", sc, pl.b, pl.a / 64);
        else if (pl.use_wg == 2) snprintf(nm, sizeof nm, "tile_conv_%s_kernel<%d,%d>", sc, pl.a, pl.b);
        else if (pl.use_wg == 1) snprintf(nm, sizeof nm, "tile_conv_kernel<%d,%d>", pl.a, pl.b);
        else if (in16) snprintf(nm, sizeof nm, "gather_conv_h16_kernel<%d,%d>", pl.a, pl.b);
        else snprintf(nm, sizeof nm, "gather_conv_kernel<%d,%d,%s>", pl.a, pl.b, pl.vec ? "true" : "false");
        cpd_launch_log_note(nm);
    }
#define CPD_LAUNCH(K, LDS) hipLaunchKernelGGL((K), grid, block, (LDS), hs, p)
#define CPD_RW_EPI_LAUNCH(NAME)                                                   \
    do {                                                                          \
        if (pl.a == 64) {                                                         \
            if (pl.b == 32) CPD_LAUNCH((NAME<32, 1>), 0);                         \
            else if (pl.b == 64) CPD_LAUNCH((NAME<64, 1>), 0);                    \
            else CPD_LAUNCH((NAME<128, 1>), 0);                                   \
        } else if (pl.b == 32) CPD_LAUNCH((NAME<32, 2>), 0);                      \
        else if (pl.b == 64) CPD_LAUNCH((NAME<64, 2>), 0);                        \
        else CPD_LAUNCH((NAME<128, 2>), 0);                                       \
        return rowwave_finish(p, hs);                                             \
    } while (0)
    if (pl.use_wg == 3 && pl.math == 2 && p.epi_lds && !p.in_pairs) {
        if (in_absmax) CPD_RW_EPI_LAUNCH(rowwave_conv_f16se_kernel);
        else CPD_RW_EPI_LAUNCH(rowwave_conv_f16e_kernel);
    }
    if (pl.use_wg == 3 && pl.math == 2 && in_absmax) {
        if (pl.a == 64) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16s_kernel<32, 1>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16s_kernel<64, 1>), 0);
            else CPD_LAUNCH((rowwave_conv_f16s_kernel<128, 1>), 0);
        } else if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16s_kernel<32, 2>), 0);
        else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16s_kernel<64, 2>), 0);
        else CPD_LAUNCH((rowwave_conv_f16s_kernel<128, 2>), 0);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 3 && pl.math == 2 && p.in_pairs && p.epi_lds) {
        if (pl.a == 64) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16pe_kernel<32, 1>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16pe_kernel<64, 1>), 0);
            else CPD_LAUNCH((rowwave_conv_f16pe_kernel<128, 1>), 0);
        } else if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16pe_kernel<32, 2>), 0);
        else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16pe_kernel<64, 2>), 0);
        else CPD_LAUNCH((rowwave_conv_f16pe_kernel<128, 2>), 0);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 3 && pl.math == 2 && p.in_pairs) {
        if (pl.a > 128) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16pw_kernel<32>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16pw_kernel<64>), 0);
            else CPD_LAUNCH((rowwave_conv_f16pw_kernel<128>), 0);
        } else if (pl.a == 64) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16p_kernel<32, 1>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16p_kernel<64, 1>), 0);
            else CPD_LAUNCH((rowwave_conv_f16p_kernel<128, 1>), 0);
        } else if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16p_kernel<32, 2>), 0);
        else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16p_kernel<64, 2>), 0);
        else CPD_LAUNCH((rowwave_conv_f16p_kernel<128, 2>), 0);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 3 && pl.math == 2) {
        if (pl.a == 64) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16_kernel<32, 1>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16_kernel<64, 1>), 0);
            else CPD_LAUNCH((rowwave_conv_f16_kernel<128, 1>), 0);
        } else if (pl.b == 32) CPD_LAUNCH((rowwave_conv_f16_kernel<32, 2>), 0);
        else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_f16_kernel<64, 2>), 0);
        else CPD_LAUNCH((rowwave_conv_f16_kernel<128, 2>), 0);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 3) {
        if (pl.a == 64) {
            if (pl.b == 32) CPD_LAUNCH((rowwave_conv_bf16_kernel<32, 1>), 0);
            else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_bf16_kernel<64, 1>), 0);
            else CPD_LAUNCH((rowwave_conv_bf16_kernel<128, 1>), 0);
        } else if (pl.b == 32) CPD_LAUNCH((rowwave_conv_bf16_kernel<32, 2>), 0);
        else if (pl.b == 64) CPD_LAUNCH((rowwave_conv_bf16_kernel<64, 2>), 0);
        else CPD_LAUNCH((rowwave_conv_bf16_kernel<128, 2>), 0);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 2 && pl.math == 2) {
        const size_t lds = 2 * (size_t)(pl.a + pl.b) * 64;
        if (in_absmax) {
            if (pl.a == 64 && pl.b == 128) CPD_LAUNCH((tile_conv_f16s_kernel<64, 128>), lds);
            else if (pl.a == 64) CPD_LAUNCH((tile_conv_f16s_kernel<64, 64>), lds);
            else if (pl.b == 64) CPD_LAUNCH((tile_conv_f16s_kernel<128, 64>), lds);
            else CPD_LAUNCH((tile_conv_f16s_kernel<128, 128>), lds);
            return rowwave_finish(p, hs);
        }
        if (pl.a == 64 && pl.b == 128) CPD_LAUNCH((tile_conv_f16_kernel<64, 128>), lds);
        else if (pl.a == 64) CPD_LAUNCH((tile_conv_f16_kernel<64, 64>), lds);
        else if (pl.b == 64) CPD_LAUNCH((tile_conv_f16_kernel<128, 64>), lds);
        else CPD_LAUNCH((tile_conv_f16_kernel<128, 128>), lds);
        return rowwave_finish(p, hs);
    }
    if (pl.use_wg == 2) {
        int db = 0;
        if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_BF16_DB")) db = atoi(e);
        if (pl.b == 64) db = 0;
        if (pl.a == 64) {                                   // small-problem variant: 64-row tiles double the workgroup count
            const size_t lds64 = 3 * (size_t)(64 + pl.b) * 64;
            if (pl.b == 128) CPD_LAUNCH((tile_conv_bf16_kernel<64, 128, false>), lds64);
            else CPD_LAUNCH((tile_conv_bf16_kernel<64, 64, false>), lds64);
            return rowwave_finish(p, hs);
        }
        const size_t lds = (db ? 2 : 1) * 3 * (size_t)(pl.a + pl.b) * 64;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_conv_bf16_kernel<128, 128, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * (128 + 128) * 64);
            attr_set = true;
        }
        if (pl.b == 64) CPD_LAUNCH((tile_conv_bf16_kernel<128, 64, false>), lds);
        else if (db) CPD_LAUNCH((tile_conv_bf16_kernel<128, 128, true>), lds);
        else CPD_LAUNCH((tile_conv_bf16_kernel<128, 128, false>), lds);
        return rowwave_finish(p, hs);
    }
#undef CPD_LAUNCH
    if (pl.use_wg) {
        gc_kernel_t k = pick_tile(pl.a, pl.b);
        if (!k) return CPD_ERR_UNSUPPORTED;
        p.n_rb = (n_out + pl.a - 1) / pl.a;
        p.n_cb = c_out / pl.b;
        p.items = p.n_rb * p.n_cb;
        size_t lds = 2 * (size_t)(pl.a + pl.b) * 128;
        hipLaunchKernelGGL(k, dim3(p.items), dim3(256), lds, cpd_s(stream), p);
        return cpd_check_launch();
    }
    gc_kernel_t k = in16 ? pick_h16(pl.a, pl.b) : pick(pl.a, pl.b, pl.vec != 0);
    if (!k) return CPD_ERR_UNSUPPORTED;
    if (in16) {
        p.wb = packed_h16_ptr(packed_w, kv, c_in, c_out);
        p.dsc = reinterpret_cast<const float *>(p.wb) + packed_h16_image_floats(kv, c_in, c_out);
        int epi = 1;                                 // the level-1 kernels' LDS epilogue (gather_conv_kernel, H16): pair rows out, in place, one column tile
        if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_H16_EPI")) epi = atoi(e);
        p.epi_lds = epi && pl.b * 16 == c_out && !out_row_map && !out_col_group && out_ld % 4 == 0 && (((uintptr_t)out) & 15) == 0 &&
                    (!residual || (res_ld % 4 == 0 && (((uintptr_t)residual) & 15) == 0)) &&
                    ((c_out == 16 && p.out_pairs == 2 && (!residual || p.res_pairs == 2)) ||
                     (c_out == 32 && p.out_pairs == 1 && (!residual || p.res_pairs == 1)));
    }
    if (!in16 && !pl.vec) {                          // the 5 -> 16 input layer writing 16-channel pair rows: the same LDS epilogue
        int epi = 1;
        if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_H16_EPI")) epi = atoi(e);
        p.epi_lds = epi && pl.b == 1 && c_out == 16 && p.out_pairs == 2 && (!residual || p.res_pairs == 2) && !out_row_map && !out_col_group &&
                    out_ld % 4 == 0 && (((uintptr_t)out) & 15) == 0 && (!residual || (res_ld % 4 == 0 && (((uintptr_t)residual) & 15) == 0));
    }
    p.n_rb = (n_out + 16 * pl.a - 1) / (16 * pl.a);
    p.n_cb = p.ntot / pl.b;
    p.items = p.n_rb * p.n_cb;
    int blocks = (p.items + 3) / 4;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, cpd_s(stream), p);
    return cpd_check_launch();
}

// ---- dense 3x3 / stride 1 / pad 1 over pixel rows, no rulebook (window_conv_bf16_kernel) ----
static int window_bn(int frames, int h, int w, int c_in, int c_out, int flags) {
    const bool tn = cpd_tuning();
    int on = split_math(flags, tn) != 0;
    if (const char *e = cpd_knob(tn, "CPD_GC_WINDOW")) on = on && atoi(e);
    if (!on || frames <= 0 || h < 2 || w < 2 || c_in <= 0 || c_out <= 0 || c_in % 32 || (c_out % 64 && c_out > 16)) return 0;
    const long long rows = (long long)frames * h * w;
    if (rows >= (1ll << 31)) return 0;
    // 16-column tile: the final head convs (c_out <= 16, e.g. 320 -> 11), which are bound by re-reading their wide input
    // once per tap -- the window stages it once per dy
    int bn = c_out <= 16 ? 16 : (c_out % 128 == 0 ? 128 : 64);
    if (const char *e = cpd_knob(tn, "CPD_GC_WINDOW_BN")) { if (atoi(e) == 64 && bn == 128) bn = 64; }      // tuning only
    long long min_wgs = 600;                    // below that the rulebook path's 64-row tiles fill the chip better
    if (const char *e = cpd_knob(tn, "CPD_GC_BF16_MIN")) min_wgs = atoll(e);
    const long long row_tiles = (rows + 127) / 128;
    int small = 1;                              // 0: round 2's single threshold
    if (const char *e = cpd_knob(tn, "CPD_GC_WINDOW_SMALL")) small = atoi(e);
    if (small && !cpd_knob(tn, "CPD_GC_BF16_MIN")) {
        // one- and four-frame batches (round 3, tools/conv_bench.py dense at FRAMES = 1 / 4): a 128-column tile from 500 workgroups
        // (4 x 94 x 94 x 256 channels: 141 vs 161 us on the rulebook path), else 64-column tiles from 250 (188 x 188 x 128 -> 128: 55
        // vs 61 us; 94 x 94 x 256: 76 vs 79), the 16-column head tile from 250 (320 -> 11: 70 vs 86); a layer whose ONLY tiling is
        // 64 columns (512 -> 64) keeps the old bound (140 vs 135 us at 277 workgroups)
        if (bn == 128) {
            if (row_tiles * (c_out / 128) >= 500) return 128;
            long long min64w = 300;
            if (const char *e = cpd_knob(tn, "CPD_GC_WINDOW_MIN64")) min64w = atoll(e);
            return row_tiles * (c_out / 64) >= min64w ? 64 : 0;     // (round 4: 94 x 94 x 256 -> 256 at ONE frame, 280 tiles: 78.6 here vs 60.7 us on the rulebook path's 64-row tiles)
        }
        if (bn == 16) return row_tiles >= 250 ? 16 : 0;
    }
    if (row_tiles * ((c_out + bn - 1) / bn) < min_wgs) return 0;
    return bn;
}
// rows per workgroup of the window kernel: 256 when the layer has a single column tile (c_out = 64 or <= 16), f16x2 arithmetic
// and still >= 4 workgroups per CU that way
static int window_bm(int frames, int h, int w, int c_out, int bn, int flags) {
    const bool tn = cpd_tuning();
    int want = 1;
    if (const char *e = cpd_knob(tn, "CPD_GC_WINDOW_BM256")) want = atoi(e);
    const long long rows = (long long)frames * h * w;
    if (want && split_math(flags, tn) == 2 && (c_out + bn - 1) / bn == 1 && bn <= 64 && (rows + 255) / 256 >= 1024) return 256;
    // several 64-column tiles (the fused 64 -> 320 head convs): 256-row tiles as well (want = 2 turns this off)
    if (want == 1 && split_math(flags, tn) == 2 && bn == 64 && (rows + 255) / 256 >= 1024) return 256;
    return 128;
}
extern "C" int cpd_conv3x3_rows_supported(int frames, int h, int w, int c_in, int c_out, int flags) {
    return window_bn(frames, h, w, c_in, c_out, flags) != 0;
}
extern "C" int cpd_conv3x3_rows_tile(int frames, int h, int w, int c_in, int c_out, int flags, int *bm, int *bn) {
    if (!bm || !bn) return CPD_ERR_ARG;
    *bn = window_bn(frames, h, w, c_in, c_out, flags);
    *bm = *bn ? window_bm(frames, h, w, c_out, *bn, flags) : 0;
    return *bn ? CPD_OK : CPD_ERR_UNSUPPORTED;
}
static int conv3x3_rows_impl(const float *in, int in_ld, int frames, int h, int w, int c_in, const float *packed_w, int c_out,
                             const float *scale, const float *shift, const float *residual, int res_ld, int relu, float *out,
                             int out_ld, int flags, const uint32_t *in_absmax, uint32_t *out_absmax, cpd_stream_t stream) {
    if (!in || !packed_w || !out || frames <= 0 || h <= 0 || w <= 0 || c_in <= 0 || c_out <= 0 || in_ld < c_in || out_ld < c_out ||
        (residual && res_ld < c_out))
        return CPD_ERR_ARG;
    const int bn = window_bn(frames, h, w, c_in, c_out, flags);
    if (!bn || in_ld % 4 || (((uintptr_t)in) & 15)) return CPD_ERR_UNSUPPORTED;
    // 32-bit buffer offsets (the table path takes larger tensors): a row before the tensor's first or after its last must land OUTSIDE
    // num_records and read zeros -- the largest excursion is (w + 1) rows before row 0 / a whole window (<= 258 rows) + w + 1 rows past
    // the last tile's first row, times the row pitch: the tensor plus that distance must stay below 2^32 (size-aware: ADVICE r4)
    if ((size_t)frames * h * w * in_ld * sizeof(float) + ((size_t)w + 1 + 288) * in_ld * sizeof(float) >= 0x100000000ull) return CPD_ERR_UNSUPPORTED;
    GcParams p;
    memset(&p, 0, sizeof(p));
    const int n_out = frames * h * w;
    const int math = split_math(flags, cpd_tuning());
    p.in = in; p.w = packed_w; p.in_absmax = in_absmax; p.out_absmax = out_absmax;
    if (math == 2) { p.wb = packed_f16_ptr(packed_w, 9, c_in, c_out); p.dsc = packed_dsc_ptr(packed_w, 9, c_in, c_out); }
    else p.wb = packed_bf16_ptr(packed_w, 9, c_in, c_out);
    p.scale = scale; p.shift = shift; p.residual = residual; p.out = out;
    p.in_ld = in_ld; p.c_in = c_in; p.kc = (c_in + 15) / 16;
    p.kv = 9; p.n_out = n_out; p.c_out = c_out; p.ntot = (c_out + 15) / 16; p.np = p.ntot * 16;
    p.res_ld = res_ld; p.relu = relu; p.out_ld = out_ld;
    p.n_sub = (n_out + 15) / 16;
    const int bm = window_bm(frames, h, w, c_out, bn, flags);
    p.n_rb = (n_out + bm - 1) / bm; p.n_cb = (c_out + bn - 1) / bn; p.items = p.n_rb * p.n_cb;
    p.img_h = h; p.img_w = w;
    {
        // the 64- / 128-column tiles' LDS epilogue: 16-byte row pieces when `out` (and `residual`) allow them
        p.epi_lds = out_ld % 4 == 0 && (((uintptr_t)out) & 15) == 0 && (!residual || (res_ld % 4 == 0 && (((uintptr_t)residual) & 15) == 0));
        if (bn >= 64 && c_out % bn) return CPD_ERR_UNSUPPORTED;       // (window_bn never picks such a tile)
    }
    if (flags & (CPD_GC_IN_PAIRS | CPD_GC_OUT_PAIRS | CPD_GC_RES_PAIRS)) {
        // fp16-pair rows (round 5): pair rows in; pair rows out on the 64- / 128-column tiles, fp32 rows out on the 16-column head tile
        // (window_conv_f16p_kernel). The batch sizes that take the 128 x 128 / 256 x 64 / 256 x 16 tiles; no residual, no pre-scaling
        const bool pin = (flags & CPD_GC_IN_PAIRS) != 0, pout = (flags & CPD_GC_OUT_PAIRS) != 0;
        if (!pin || (flags & CPD_GC_RES_PAIRS) || residual || in_absmax || math != 2 || (bn >= 64 && !p.epi_lds)) return CPD_ERR_UNSUPPORTED;
        if (!((bn == 128 && bm == 128) || (bn == 64 && (bm == 256 || bm == 128)) || (bn == 16 && (bm == 256 || bm == 128)))) return CPD_ERR_UNSUPPORTED;
        if (pout != (bn >= 64)) return CPD_ERR_UNSUPPORTED;
        p.in_pairs = 1; p.out_pairs = pout ? 1 : 0;
        char nm[96];
        snprintf(nm, sizeof nm, "window_conv_f16p_kernel<%d,%d>", bn, bm);
        cpd_launch_log_note(nm);
        const size_t win = (size_t)(bm + 8) * 128, tile = (size_t)64 * (bn + 4) * 4;
        const size_t wts = bn >= 64 ? 2 * 2 * (size_t)bn * 64 : 2 * (size_t)bn * 64;
        // (round 6, measured and NOT kept: padding this launch's LDS so that only TWO 128 x 128 workgroups fit a CU -- room for a sparse
        // kernel of the other batch in flight on the same CU, matrix pipe and vector-memory pipe side by side -- 1219.9 / 1217.7 frames/s
        // without, 1191.9 / 1176.6 with 26 KB of padding, 1034.1 / 1021.5 with 40 KB: gpurun_out r06_winpad)
        const size_t ldsp = win + wts > tile ? win + wts : tile;
        if (bn == 128) hipLaunchKernelGGL((window_conv_f16p_kernel<128, 128>), dim3(p.items), dim3(256), ldsp, cpd_s(stream), p);
        else if (bn == 64 && bm == 256) hipLaunchKernelGGL((window_conv_f16p_kernel<64, 256>), dim3(p.items), dim3(256), ldsp, cpd_s(stream), p);
        else if (bn == 64) hipLaunchKernelGGL((window_conv_f16p_kernel<64, 128>), dim3(p.items), dim3(256), ldsp, cpd_s(stream), p);      // (4 ... 7-frame batches)
        else if (bm == 128) hipLaunchKernelGGL((window_conv_f16p16_kernel<128>), dim3(p.items), dim3(256), win + 2 * 3 * 2 * 4 * 16 * 16, cpd_s(stream), p);
        else {
            // the 16-column head tile: stage groups of three taps, next group's rows and weights a whole group ahead (window_conv_pairs16_body)
            int grouped = 1;
            if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_P16_GROUPED")) grouped = atoi(e);
            if (grouped) hipLaunchKernelGGL((window_conv_f16p16_kernel<256>), dim3(p.items), dim3(256), win + 2 * 3 * 2 * 4 * 16 * 16, cpd_s(stream), p);
            else hipLaunchKernelGGL((window_conv_f16p_kernel<16, 256>), dim3(p.items), dim3(256), ldsp, cpd_s(stream), p);
        }
        return cpd_check_launch();
    }
    {
        char nm[96];
        snprintf(nm, sizeof nm, "window_conv_%s_kernel<%d,%d>", math == 2 ? (in_absmax ? "f16s" : "f16") : "bf16", bn, bm);
        cpd_launch_log_note(nm);
    }
    const size_t lds = (math == 2 ? 2 : 3) * ((size_t)(128 + 16) * 64 + (size_t)bn * 64);
    if (math == 2 && bm == 256 && in_absmax) {
        const size_t ldsa = 2 * (size_t)(256 + 16) * 64;
        if (bn == 64) hipLaunchKernelGGL((window_conv_f16s_kernel<64, 256>), dim3(p.items), dim3(256), ldsa + 2 * 2 * 64 * 64, cpd_s(stream), p);
        else hipLaunchKernelGGL((window_conv_f16s_kernel<16, 256>), dim3(p.items), dim3(256), ldsa + 2 * 16 * 64, cpd_s(stream), p);
        return cpd_check_launch();
    }
    if (math == 2 && in_absmax) {
        const size_t lds2 = 2 * (size_t)(128 + 16) * 64 + 2 * 2 * (size_t)bn * 64;
        if (bn == 128) hipLaunchKernelGGL((window_conv_f16s_kernel<128>), dim3(p.items), dim3(256), lds2, cpd_s(stream), p);
        else if (bn == 64) hipLaunchKernelGGL((window_conv_f16s_kernel<64>), dim3(p.items), dim3(256), lds2, cpd_s(stream), p);
        else hipLaunchKernelGGL((window_conv_f16s_kernel<16>), dim3(p.items), dim3(256), lds, cpd_s(stream), p);
        return cpd_check_launch();
    }
    if (math == 2 && bm == 256) {
        const size_t ldsa = 2 * (size_t)(256 + 16) * 64;
        if (bn == 64) hipLaunchKernelGGL((window_conv_f16_kernel<64, 256>), dim3(p.items), dim3(256), ldsa + 2 * 2 * 64 * 64, cpd_s(stream), p);
        else hipLaunchKernelGGL((window_conv_f16_kernel<16, 256>), dim3(p.items), dim3(256), ldsa + 2 * 16 * 64, cpd_s(stream), p);
        return cpd_check_launch();
    }
    if (math == 2) {
        const size_t lds2 = 2 * (size_t)(128 + 16) * 64 + 2 * 2 * (size_t)bn * 64;    // window image + two weight buffers
        if (bn == 128) hipLaunchKernelGGL((window_conv_f16_kernel<128>), dim3(p.items), dim3(256), lds2, cpd_s(stream), p);
        else if (bn == 64) hipLaunchKernelGGL((window_conv_f16_kernel<64>), dim3(p.items), dim3(256), lds2, cpd_s(stream), p);
        else hipLaunchKernelGGL((window_conv_f16_kernel<16>), dim3(p.items), dim3(256), lds, cpd_s(stream), p);
        return cpd_check_launch();
    }
    if (bn == 128) hipLaunchKernelGGL((window_conv_bf16_kernel<128>), dim3(p.items), dim3(256), lds, cpd_s(stream), p);
    else if (bn == 64) hipLaunchKernelGGL((window_conv_bf16_kernel<64>), dim3(p.items), dim3(256), lds, cpd_s(stream), p);
    else hipLaunchKernelGGL((window_conv_bf16_kernel<16>), dim3(p.items), dim3(256), lds, cpd_s(stream), p);
    return cpd_check_launch();
}

extern "C" int cpd_gather_conv(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                               const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale, const float *shift,
                               const float *residual, int res_ld, int relu, float *out, int out_ld,
                               const int32_t *out_row_map, int out_col_group, int flags, cpd_stream_t stream) {
    return gather_conv_impl(in, in_ld, n_in, c_in, packed_w, nbr, tapmask, kv, n_out, c_out, scale, shift, residual, res_ld, relu, out,
                            out_ld, out_row_map, out_col_group, flags, nullptr, nullptr, stream);
}
extern "C" int cpd_gather_conv_scaled(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                                      const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale, const float *shift,
                                      const float *residual, int res_ld, int relu, float *out, int out_ld,
                                      const int32_t *out_row_map, int out_col_group, int flags, const uint32_t *in_absmax,
                                      cpd_stream_t stream) {
    return gather_conv_impl(in, in_ld, n_in, c_in, packed_w, nbr, tapmask, kv, n_out, c_out, scale, shift, residual, res_ld, relu, out,
                            out_ld, out_row_map, out_col_group, flags, in_absmax, nullptr, stream);
}
extern "C" int cpd_gather_conv_ranged(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                                      const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale, const float *shift,
                                      const float *residual, int res_ld, int relu, float *out, int out_ld,
                                      const int32_t *out_row_map, int out_col_group, int flags, const uint32_t *in_absmax,
                                      uint32_t *out_absmax, cpd_stream_t stream) {
    return gather_conv_impl(in, in_ld, n_in, c_in, packed_w, nbr, tapmask, kv, n_out, c_out, scale, shift, residual, res_ld, relu, out,
                            out_ld, out_row_map, out_col_group, flags, in_absmax, out_absmax, stream);
}
// The staged row-wave kernel (rowplan_conv_f16p_kernel): a sub-manifold 3 x 3 x 3 layer on fp16-pair rows whose rulebook comes with a
// row plan. Problems it does not take return CPD_ERR_UNSUPPORTED (the caller then runs cpd_gather_conv_ws: same result).
extern "C" int cpd_gather_conv_planned_supported(int n_in, int n_out, int c_in, int c_out, int in_ld, int kv, int flags) {
    const bool tn = cpd_tuning();
    if (const char *e = cpd_knob(tn, "CPD_GC_PLANNED")) { if (!atoi(e)) return 0; }
    long long min_tiles = 512;                 // below two workgroups per CU the row-wave kernel's tap split fills the chip better
    if (const char *e = cpd_knob(tn, "CPD_GC_PLANNED_MIN")) min_tiles = atoll(e);
    return kv == 27 && (flags & CPD_GC_F16X2) && (flags & CPD_GC_IN_PAIRS) && (flags & CPD_GC_OUT_PAIRS) && !(flags & CPD_GC_DENSE) && c_in % 32 == 0 &&
           (c_out == 32 || c_out == 64 || c_out == 128) && in_ld % 4 == 0 && n_in > 0 && (n_out + 127) / 128 >= min_tiles &&
           (size_t)n_in * in_ld * sizeof(float) < 0xfffff000ull;          // (rows are read through a 4 GB buffer resource)
}
extern "C" int cpd_gather_conv_planned(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const uint32_t *tapmask,
                                       const uint16_t *plan_slots, const int32_t *plan_ulist, const int32_t *plan_count, int tile_rows, int kv,
                                       int n_out, int c_out, const float *scale, const float *shift, const float *residual, int res_ld,
                                       int relu, float *out, int out_ld, int flags, uint32_t *out_absmax, cpd_stream_t stream) {
    if (!in || !packed_w || !out || !plan_slots || !plan_ulist || !plan_count || n_in <= 0 || n_out < 0 || in_ld < c_in || out_ld < c_out ||
        (residual && res_ld < c_out))
        return CPD_ERR_ARG;
    if (n_out == 0) return CPD_OK;
    if (tile_rows != 128 && tile_rows != 256) return CPD_ERR_ARG;
    if (!cpd_gather_conv_planned_supported(n_in, n_out, c_in, c_out, in_ld, kv, flags) || (((uintptr_t)in) & 15)) return CPD_ERR_UNSUPPORTED;
    if ((flags & CPD_GC_RES_PAIRS) && !residual) return CPD_ERR_ARG;
    if (residual && !(flags & CPD_GC_RES_PAIRS)) return CPD_ERR_UNSUPPORTED;          // (the kernel's epilogue reads pair rows)
    if ((((uintptr_t)out) & 15) || out_ld % 4 || (residual && ((((uintptr_t)residual) & 15) || res_ld % 4))) return CPD_ERR_UNSUPPORTED;
    GcParams p;
    memset(&p, 0, sizeof p);
    p.in_pairs = 1;
    p.out_pairs = (flags & CPD_GC_OUT_PAIRS) ? 1 : 0;
    p.res_pairs = (flags & CPD_GC_RES_PAIRS) ? 1 : 0;
    p.split = 1;
    p.in = in; p.w = packed_w; p.out_absmax = out_absmax; p.tapmask = tapmask; p.n_sub = (n_out + 15) / 16;
    p.scale = scale; p.shift = shift; p.residual = residual; p.out = out;
    p.in_ld = in_ld; p.c_in = c_in; p.kc = c_in / 16; p.n_in_rows = n_in;
    p.kv = kv; p.n_out = n_out; p.c_out = c_out; p.ntot = c_out / 16; p.np = c_out;
    p.res_ld = res_ld; p.relu = relu; p.out_ld = out_ld;
    p.wb = packed_f16_ptr(packed_w, kv, c_in, c_out);
    p.dsc = packed_dsc_ptr(packed_w, kv, c_in, c_out);
    p.plan_slots = plan_slots; p.plan_ulist = plan_ulist; p.plan_count = plan_count;
    p.n_rb = (n_out + 127) / 128; p.n_cb = 1; p.items = p.n_rb;
    char nm[64];
    snprintf(nm, sizeof nm, tile_rows == 256 ? "rowplan_conv_f16p_kernel<%d,256>" : "rowplan_conv_f16p_kernel<%d>", c_out);
    cpd_launch_log_note(nm);
    const dim3 grid(p.items), block(256);
    hipStream_t hs = cpd_s(stream);
    int tb32 = 3;             // 32 columns: the three dx taps of one dy per weight stage (12 KB, 3 workgroups per CU: 443 us) or all nine (36 KB, 2: 559 us)
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_GC_PLANNED_TB32")) tb32 = atoi(e);
    if (tile_rows == 256) {
        const dim3 grid2((n_out + 255) / 256), block2(512);
        if (c_out == 32) hipLaunchKernelGGL((rowplan_conv_f16p_kernel<32, 3, 256>), grid2, block2, 0, hs, p);
        else if (c_out == 64) hipLaunchKernelGGL((rowplan_conv_f16p_kernel<64, 3, 256>), grid2, block2, 0, hs, p);
        else hipLaunchKernelGGL((rowplan_conv_f16p_kernel<128, 1, 256>), grid2, block2, 0, hs, p);
    } else if (c_out == 32 && tb32 == 3) hipLaunchKernelGGL((rowplan_conv_f16p_kernel<32, 3>), grid, block, 0, hs, p);
    else if (c_out == 32) hipLaunchKernelGGL((rowplan_conv_f16p_kernel<32, 9>), grid, block, 0, hs, p);
    else if (c_out == 64) hipLaunchKernelGGL((rowplan_conv_f16p_kernel<64, 3>), grid, block, 0, hs, p);
    else hipLaunchKernelGGL((rowplan_conv_f16p_kernel<128, 3>), grid, block, 0, hs, p);
    return cpd_check_launch();
}
extern "C" size_t cpd_gather_conv_split_bytes(int n_out, int c_in, int c_out, int in_ld, int kv, int flags) {
    if (n_out <= 0 || c_in <= 0 || c_out <= 0 || kv <= 0) return 0;
    const GcPlan pl = plan(n_out, c_in, c_out, in_ld, nullptr, flags);
    if ((pl.use_wg != 3 || kv > CPD_RW_TAPS) && pl.use_wg != 2) return 0;
    const int sp = rowwave_split(pl, n_out, c_in, c_out, kv);
    return sp > 1 ? (size_t)sp * n_out * c_out * sizeof(float) : 0;
}
extern "C" int cpd_gather_conv_ws(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const int32_t *nbr,
                                  const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale, const float *shift,
                                  const float *residual, int res_ld, int relu, float *out, int out_ld,
                                  const int32_t *out_row_map, int out_col_group, int flags, const uint32_t *in_absmax,
                                  uint32_t *out_absmax, void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    return gather_conv_impl(in, in_ld, n_in, c_in, packed_w, nbr, tapmask, kv, n_out, c_out, scale, shift, residual, res_ld, relu, out,
                            out_ld, out_row_map, out_col_group, flags, in_absmax, out_absmax, stream, (float *)workspace, workspace_bytes);
}
extern "C" int cpd_conv3x3_rows(const float *in, int in_ld, int frames, int h, int w, int c_in, const float *packed_w, int c_out,
                                const float *scale, const float *shift, const float *residual, int res_ld, int relu, float *out,
                                int out_ld, int flags, cpd_stream_t stream) {
    return conv3x3_rows_impl(in, in_ld, frames, h, w, c_in, packed_w, c_out, scale, shift, residual, res_ld, relu, out, out_ld, flags,
                             nullptr, nullptr, stream);
}
extern "C" int cpd_conv3x3_rows_scaled(const float *in, int in_ld, int frames, int h, int w, int c_in, const float *packed_w, int c_out,
                                       const float *scale, const float *shift, const float *residual, int res_ld, int relu, float *out,
                                       int out_ld, int flags, const uint32_t *in_absmax, cpd_stream_t stream) {
    return conv3x3_rows_impl(in, in_ld, frames, h, w, c_in, packed_w, c_out, scale, shift, residual, res_ld, relu, out, out_ld, flags,
                             in_absmax, nullptr, stream);
}
extern "C" int cpd_conv3x3_rows_ranged(const float *in, int in_ld, int frames, int h, int w, int c_in, const float *packed_w, int c_out,
                                       const float *scale, const float *shift, const float *residual, int res_ld, int relu, float *out,
                                       int out_ld, int flags, const uint32_t *in_absmax, uint32_t *out_absmax, cpd_stream_t stream) {
    return conv3x3_rows_impl(in, in_ld, frames, h, w, c_in, packed_w, c_out, scale, shift, residual, res_ld, relu, out, out_ld, flags,
                             in_absmax, out_absmax, stream);
}
// bits of max |x| over n rows x c columns (row pitch ld) raised into an absmax block: the range of a tensor no kernel of this
// library produced (the first split layer of a chain fed from outside)
__global__ void __launch_bounds__(256) absmax_rows_kernel(const float *__restrict__ x, int ld, long long n, int c, uint32_t *__restrict__ block) {
    uint32_t m = 0;
    const long long total = n * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const uint32_t b = __float_as_uint(x[r * ld + (i - r * c)]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(block + (blockIdx.x & (CPD_ABSMAX_SLOTS - 1)) * CPD_ABSMAX_STRIDE, m);
}
extern "C" int cpd_absmax_rows(const float *x, int ld, long long n, int c, uint32_t *absmax_block, cpd_stream_t stream) {
    if (!x || !absmax_block || n < 0 || c <= 0 || ld < c) return CPD_ERR_ARG;
    if (n == 0) return CPD_OK;
    const long long blocks = cpd_div_up(n * c, 256 * 8);
    absmax_rows_kernel<<<(unsigned)(blocks > 2048 ? 2048 : blocks), 256, 0, cpd_s(stream)>>>(x, ld, n, c, absmax_block);
    return cpd_check_launch();
}
