// site_index.hip -- active-site index, rulebooks and densify for the sparse backbone (gfx950).
//
// Replaces the indice-pair ("rulebook") generation and .dense() of [SPCONV] spconv-cu111 2.1.22
// as used by cpd/models/backbones_3d/spconv_backbone.py:414-455,524-529 and
// cpd/models/backbones_2d/map_to_bev/height_compression.py:136-138.
//
// Design (MI355X-first, not a hash-table port): the occupancy of a (batch,z,y,x) grid is a dense
// BITMAP (1 bit/cell) plus a per-64-cell popcount prefix. That gives
//   * coordinate -> row lookup in two coalescable loads (word + prefix), the three x-neighbours
//     of a 3x3x3 stencil share one word;
//   * the canonical ascending (b,z,y,x) order of any site set for free (rank = prefix+popcount),
//     so SparseConv3d outputs need neither sort nor unique -- parity with the reference is defined
//     on that order;
//   * no probing, no collisions, deterministic results.
// Memory scales with grid volume: 17 MB for the 41x1504x1504 Waymo grid, 104 MB for the
// 61x3008x3008 stress grid -- irrelevant next to 288 GB of HBM3E and mostly Infinity-Cache hits.
// Row ids of an arbitrary-order site list (level 0 keeps the voxelizer's first-appearance order)
// go through perm[rank]; canonical lists use rank directly.
#include "common.h"

#include "site_index_layout.h"

namespace {

__global__ void __launch_bounds__(256) index_mark_kernel(const int32_t *__restrict__ idx, int n, Grid g, uint64_t *bitmap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    if ((unsigned)q.x >= (unsigned)g.b || (unsigned)q.y >= (unsigned)g.d || (unsigned)q.z >= (unsigned)g.h ||
        (unsigned)q.w >= (unsigned)g.w)
        return;  // out-of-range rows are ignored (they can never be looked up)
    long long k = g.key(q.x, q.y, q.z, q.w);
    atomicOr((unsigned long long *)&bitmap[k >> 6], 1ull << (k & 63));
}

__global__ void __launch_bounds__(256) index_perm_kernel(const int32_t *__restrict__ idx, int n, Grid g,
                                                         const uint64_t *__restrict__ bitmap,
                                                         const uint32_t *__restrict__ base, int32_t *perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    if ((unsigned)q.x >= (unsigned)g.b || (unsigned)q.y >= (unsigned)g.d || (unsigned)q.z >= (unsigned)g.h ||
        (unsigned)q.w >= (unsigned)g.w)
        return;
    int32_t r = site_lookup(bitmap, base, nullptr, g.key(q.x, q.y, q.z, q.w));
    perm[r] = i;
}

struct PopcFn {
    const uint64_t *bitmap;
    __device__ uint32_t operator()(long long w) const { return (uint32_t)__popcll(bitmap[w]); }
};
struct StoreBaseFn {
    uint32_t *base;
    __device__ void operator()(long long w, uint32_t, uint32_t prefix) const { base[w] = prefix; }
};
struct EmitFn {  // write the coordinates of every set bit in canonical order
    const uint64_t *bitmap;
    const uint32_t *base;
    int32_t *out;
    int n_cap;
    Grid g;
    __device__ void operator()(long long wi, uint32_t, uint32_t) const {
        uint64_t w = bitmap[wi];
        uint32_t r = base[wi];
        while (w) {
            int bpos = __ffsll((unsigned long long)w) - 1;
            w &= w - 1;
            long long k = wi * 64 + bpos;
            if ((int)r < n_cap) {
                int x = (int)(k % g.w);
                long long t = k / g.w;
                int y = (int)(t % g.h);
                t /= g.h;
                int z = (int)(t % g.d);
                int bi = (int)(t / g.d);
                reinterpret_cast<int4 *>(out)[r] = make_int4(bi, z, y, x);
            }
            ++r;
        }
    }
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"      // the unroll requests below only bind in the K3 instantiation
// One thread per output row, all taps: nbr[t][j] coalesced over j. K3: the 3 x 3 x 3 kernel of every SubM / strided layer of
// the backbone with its tap loops unrolled (tap numbers, mask bits and table rows become constants); else runtime extents.
template <bool K3>
__global__ void __launch_bounds__(256)
rulebook_kernel(const int32_t *__restrict__ out_idx, int n_out, Grid gin, int kd_, int kh_, int kw_, int sd, int sh, int sw,
                int pd, int ph, int pw, const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                const int32_t *__restrict__ perm_in, const int32_t *__restrict__ flags, int32_t *__restrict__ nbr,
                uint32_t *__restrict__ tapmask, uint32_t *__restrict__ row_pattern) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int kd = K3 ? 3 : kd_, kh = K3 ? 3 : kh_, kw = K3 ? 3 : kw_;
    const bool live = j < n_out;
    if (!live) j = n_out - 1;  // keep whole waves alive for the ballots below
    // canonical list: row id = rank; otherwise row id = map[rank] (the index's own map or the caller's: index_order)
    // nbr == nullptr: pattern pass -- only which taps exist is wanted (row_pattern), no row ids, no table
    const int32_t *perm = nbr ? index_order(flags, perm_in) : nullptr;
    int4 q = reinterpret_cast<const int4 *>(out_idx)[j];
    int t = 0;
    uint32_t my_mask = 0;  // lanes 0..3 of a wave collect the tap masks of its four 16-row sub-tiles
    uint32_t pattern = 0;
    const int lane = threadIdx.x & 63;
    const int x0 = q.w * sw - pw;
    const int xlo = x0 < 0 ? 0 : x0, xhi = x0 + kw - 1 < gin.w ? x0 + kw - 1 : gin.w - 1;
#pragma unroll
    for (int tz = 0; tz < kd; ++tz) {
        int z = q.y * sd - pd + tz;
#pragma unroll
        for (int ty = 0; ty < kh; ++ty) {
            int y = q.z * sh - ph + ty;
            // the kw taps of one (z, y) row are consecutive cells: they share one bitmap word and its prefix (two when the
            // run crosses a 64-cell boundary), fetched once per row instead of once per tap
            const bool row_ok = (unsigned)z < (unsigned)gin.d && (unsigned)y < (unsigned)gin.h && (unsigned)q.x < (unsigned)gin.b &&
                                xlo <= xhi;
            long long rowkey = 0;
            uint64_t wa = 0, wb = 0;
            uint32_t ba = 0, bb = 0;
            long long wia = 0;
            if (row_ok) {
                rowkey = gin.key(q.x, z, y, 0);
                wia = (rowkey + xlo) >> 6;
                const long long wib = (rowkey + xhi) >> 6;
                wa = bitmap[wia]; ba = base[wia];
                wb = wa; bb = ba;
                if (wib != wia) { wb = bitmap[wib]; bb = base[wib]; }
            }
#pragma unroll
            for (int tx = 0; tx < kw; ++tx, ++t) {
                int x = x0 + tx;
                int32_t r = -1;
                if (row_ok && (unsigned)x < (unsigned)gin.w) {
                    const long long k = rowkey + x;
                    const bool first = (k >> 6) == wia;
                    const uint64_t w = first ? wa : wb;
                    const uint64_t bit = 1ull << (k & 63);
                    if (w & bit) {
                        r = (int32_t)((first ? ba : bb) + __popcll(w & (bit - 1ull)));
                        if (perm) r = perm[r];
                    }
                }
                if (live && nbr) nbr[(size_t)t * n_out + j] = r;
                if (r >= 0 && t < 32) pattern |= 1u << t;
                const unsigned long long hit = __ballot(live && r >= 0);
                if (lane < 4 && t < 32 && ((hit >> (16 * lane)) & 0xffffull)) my_mask |= 1u << t;
            }
        }
    }
    if (tapmask && lane < 4) {
        const int sub = ((blockIdx.x * blockDim.x + (threadIdx.x & ~63)) >> 4) + lane;
        if (sub < (n_out + 15) / 16) tapmask[sub] = my_mask;
    }
    if (row_pattern && live) row_pattern[j] = pattern;
}

#pragma clang diagnostic pop

// ---- rulebook of a CHUNK-ORDERED level (round 4) ---------------------------------------------------------------------------------
// rulebook_kernel above walks the output rows in THEIR order, one row per lane. For a level in tap-pattern order the rows of a wave are
// not neighbours in space: its nine (z, y) bitmap-word / prefix fetches and its 27 rank -> row fetches go to 64 different cache
// lines per instruction -- one L1 tag lookup per lane, ~2900 per 64 rows: the kernel runs at the L1's lookup rate (a tap-ordered
// level cost 2.2x a canonical one, profiles/README.md). But a chunk-ordered level (cpd_order_rows_by_taps) permutes rows only INSIDE
// chunks of `chunk` canonical rows. So: one workgroup per chunk walks the chunk's rows in CANONICAL order -- lanes are x-neighbours,
// their words, prefixes and neighbour ranks coincide or are adjacent: a few lines per instruction --, drops each row's three dx taps
// of one (dz, dy) into an LDS stage at the row's NEW position (old_to_new - chunk start), and the stage goes out as whole lines:
// nbr[t][chunk start ...]. Nine such passes; the tap masks come from the staged values (16 new rows = 16 consecutive lanes).
// Same table, bit for bit. 3 x 3 x 3 kernels, any stride / padding.
#define CPD_RBC_CHUNK 4096
__global__ void __launch_bounds__(1024)
rulebook_chunk_kernel(const int32_t *__restrict__ out_canon, const int32_t *__restrict__ out_o2n, int n_out, Grid gin, int sd, int sh, int sw,
                      int pd, int ph, int pw, const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                      const int32_t *__restrict__ perm_in, const int32_t *__restrict__ flags, int32_t *__restrict__ nbr,
                      uint32_t *__restrict__ tapmask) {
    __shared__ int32_t stage[3][CPD_RBC_CHUNK];
    constexpr int RPT = CPD_RBC_CHUNK / 1024;                       // rows per thread
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * CPD_RBC_CHUNK;
    const int32_t *perm = index_order(flags, perm_in);
    int4 q[RPT];
    int dst[RPT];                                                   // new position inside the chunk (-1: no such row)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int j = c0 + k * 1024 + tid;                          // canonical row
        dst[k] = -1;
        q[k] = make_int4(0, 0, 0, 0);
        if (j < n_out) {
            q[k] = reinterpret_cast<const int4 *>(out_canon)[j];
            dst[k] = (out_o2n ? out_o2n[j] : j) - c0;
        }
    }
    uint32_t pattern[RPT];                                          // per NEW row c0 + k * 1024 + tid: which taps exist
#pragma unroll
    for (int k = 0; k < RPT; ++k) pattern[k] = 0;
    for (int tzy = 0; tzy < 9; ++tzy) {
        const int tz = tzy / 3, ty = tzy - 3 * tz;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (dst[k] < 0) continue;
            const int z = q[k].y * sd - pd + tz, y = q[k].z * sh - ph + ty;
            const int x0 = q[k].w * sw - pw;
            int32_t r3[3] = {-1, -1, -1};
            if ((unsigned)z < (unsigned)gin.d && (unsigned)y < (unsigned)gin.h && (unsigned)q[k].x < (unsigned)gin.b) {
                const long long rowkey = gin.key(q[k].x, z, y, 0);
                const int xlo = x0 < 0 ? 0 : x0, xhi = x0 + 2 < gin.w ? x0 + 2 : gin.w - 1;
                if (xlo <= xhi) {
                    const long long wia = (rowkey + xlo) >> 6, wib = (rowkey + xhi) >> 6;
                    const uint64_t wa = bitmap[wia];
                    const uint32_t ba = base[wia];
                    uint64_t wb = wa;
                    uint32_t bb = ba;
                    if (wib != wia) { wb = bitmap[wib]; bb = base[wib]; }
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const int x = x0 + tx;
                        if ((unsigned)x >= (unsigned)gin.w) continue;
                        const long long kk = rowkey + x;
                        const bool first = (kk >> 6) == wia;
                        const uint64_t w = first ? wa : wb;
                        const uint64_t bit = 1ull << (kk & 63);
                        if (w & bit) {
                            int32_t r = (int32_t)((first ? ba : bb) + __popcll(w & (bit - 1ull)));
                            if (perm) r = perm[r];
                            r3[tx] = r;
                        }
                    }
                }
            }
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) stage[tx][dst[k]] = r3[tx];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = k * 1024 + tid, row = c0 + i;
            if (row < n_out) {
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int32_t v = stage[tx][i];
                    nbr[(size_t)(3 * tzy + tx) * n_out + row] = v;
                    if (v >= 0) pattern[k] |= 1u << (3 * tzy + tx);
                }
            }
        }
        __syncthreads();
    }
    if (tapmask) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {                             // 16 consecutive new rows = 16 consecutive lanes: OR over each 16-lane group
            uint32_t m = pattern[k];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o, 16);
            const int row = c0 + k * 1024 + tid;
            if ((tid & 15) == 0 && row < n_out) tapmask[row >> 4] = m;
        }
    }
}

// Output sites of a regular (strided) sparse conv: every input site marks the output cells whose receptive field holds it.
// One thread per input site; for each (z', y') output row it reaches, the x' cells it reaches form a mask inside ONE 64-cell
// bitmap word (two at a word boundary: the second part goes out on its own). Neighbouring sites of a row -- consecutive lanes
// in canonical order -- hit the same word, so the masks of a run of lanes with equal word index are OR-ed together in the
// wave (log-step shuffles) and only the run's last lane touches memory: one test (+ one atomic when a bit is new) per run
// instead of one per site and tap.
__device__ __forceinline__ void outset_flush(uint64_t *bitmap, long long word, uint64_t mask, bool valid) {
    const int lane = threadIdx.x & 63;
    long long key = valid ? word : -1 - lane;             // invalid lanes get unique keys: they never join a run
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long k2 = __shfl_up(key, d, 64);
        const uint64_t m2 = __shfl_up(mask, d, 64);
        if (lane >= d && k2 == key) mask |= m2;             // runs are contiguous: equal keys d apart lie in one run
    }
    const long long kn = __shfl_down(key, 1, 64);
    const bool tail = lane == 63 || kn != key;
    if (valid && tail && (bitmap[word] & mask) != mask) atomicOr((unsigned long long *)&bitmap[word], (unsigned long long)mask);
}

__global__ void __launch_bounds__(256)
outset_mark_kernel(const int32_t *__restrict__ in_idx, int n_in, Grid gout, int kd, int kh, int kw, int sd, int sh, int sw,
                   int pd, int ph, int pw, uint64_t *bitmap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_in;
    int4 q = make_int4(0, 0, 0, 0);
    if (live) q = reinterpret_cast<const int4 *>(in_idx)[i];
    const bool ok = live && (unsigned)q.x < (unsigned)gout.b;
    // x' cells reached: nx = (q.w + pw - tx) / sw for the taps tx with the right parity: a short ascending run of cells
    int xlo = 0x7fffffff, xhi = -1;
    for (int tx = 0; tx < kw; ++tx) {
        int nx = q.w + pw - tx;
        if (nx < 0 || nx % sw) continue;
        nx /= sw;
        if (nx >= gout.w) continue;
        xlo = nx < xlo ? nx : xlo;
        xhi = nx > xhi ? nx : xhi;
    }
    // (all lanes of a wave walk the same (tz, ty) loop; a lane without a valid cell for a pair joins the shuffles as invalid)
    for (int tz = 0; tz < kd; ++tz) {
        int nz = q.y + pd - tz;
        const bool zok = nz >= 0 && nz % sd == 0 && nz / sd < gout.d;
        nz /= sd;
        for (int ty = 0; ty < kh; ++ty) {
            int ny = q.z + ph - ty;
            const bool yok = ny >= 0 && ny % sh == 0 && ny / sh < gout.h;
            ny /= sh;
            const bool v = ok && zok && yok && xlo <= xhi;
            if (!__any(v)) continue;
            long long k0 = 0;
            uint64_t m0 = 0, m1 = 0;
            if (v) {
                k0 = gout.key(q.x, nz, ny, xlo);
                const int b0 = (int)(k0 & 63), cnt = xhi - xlo + 1;      // cells xlo..xhi are consecutive for sw <= 2 ... (else per cell below)
                if (sw <= 2 || cnt == 1) {
                    const uint64_t run = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
                    m0 = run << b0;
                    m1 = b0 + cnt > 64 ? run >> (64 - b0) : 0ull;
                } else {                                                  // sparse x' set (stride > 2): mark cell by cell
                    for (int tx = 0; tx < kw; ++tx) {
                        int nx = q.w + pw - tx;
                        if (nx < 0 || nx % sw) continue;
                        nx /= sw;
                        if (nx >= gout.w) continue;
                        const int b = b0 + (nx - xlo);
                        if (b < 64) m0 |= 1ull << b; else m1 |= 1ull << (b - 64);
                    }
                }
            }
            outset_flush(bitmap, k0 >> 6, m0, v);
            if (__any(v && m1)) outset_flush(bitmap, (k0 >> 6) + 1, m1, v && m1);
        }
    }
}

__global__ void __launch_bounds__(256) emit_words_kernel(long long words, EmitFn fn) {
    long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < words) fn(w, 0u, 0u);
}

__global__ void __launch_bounds__(256) densify_nhwc_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                           int n, int c4, Grid g, float *__restrict__ out) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int i = (int)(tid / c4), k = (int)(tid % c4);
    if (i >= n) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    float4 v = reinterpret_cast<const float4 *>(feat)[(size_t)i * c4 + k];
    size_t pix = ((size_t)q.x * g.h + q.z) * g.w + q.w;
    reinterpret_cast<float4 *>(out)[(pix * g.d + q.y) * c4 + k] = v;
}

// the same rows set back to zero: the second half of a densify into a PERSISTENT pre-zeroed map (cpd_densify_nhwc_rows / _clear)
__global__ void __launch_bounds__(256) densify_clear_kernel(const int32_t *__restrict__ idx, int n, int c4, Grid g, float *__restrict__ out) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int i = (int)(tid / c4), k = (int)(tid % c4);
    if (i >= n) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    size_t pix = ((size_t)q.x * g.h + q.z) * g.w + q.w;
    reinterpret_cast<float4 *>(out)[(pix * g.d + q.y) * c4 + k] = float4{0.f, 0.f, 0.f, 0.f};
}

// channels-last pixels with the REFERENCE's channel order c*D + z (what view(N, C*D, H, W) of spconv's dense() gives): lanes run over
// the channels of a site, so a wave writes D-strided floats of one pixel
__global__ void __launch_bounds__(256) densify_nhwc_cd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                              int n, int c, Grid g, float *__restrict__ out) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int i = (int)(tid / c), k = (int)(tid % c);
    if (i >= n) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    size_t pix = ((size_t)q.x * g.h + q.z) * g.w + q.w;
    out[(pix * c + k) * g.d + q.y] = feat[(size_t)i * c + k];
}

__global__ void __launch_bounds__(256) densify_nchw_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                           int n, int c, Grid g, float *__restrict__ out) {
    // lanes run over sites (canonical order => consecutive x), channels looped through LDS-free
    // strided reads: out[b][ch*D+z][y][x]
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int ch = (int)(tid / n), i = (int)(tid % n);
    if (ch >= c) return;
    int4 q = reinterpret_cast<const int4 *>(idx)[i];
    size_t plane = (size_t)g.h * g.w;
    out[(((size_t)q.x * c + ch) * g.d + q.y) * plane + (size_t)q.z * g.w + q.w] = feat[(size_t)i * c + ch];
}

__global__ void __launch_bounds__(256) rulebook_conv2d_kernel(int batch, int h, int w, int ho, int wo, int kh, int kw,
                                                              int stride, int pad, int32_t *__restrict__ nbr) {
    long long n_out = (long long)batch * ho * wo;
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    int x = (int)(j % wo);
    long long t2 = j / wo;
    int y = (int)(t2 % ho), b = (int)(t2 / ho);
    int t = 0;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx, ++t) {
            int iy = y * stride - pad + ky, ix = x * stride - pad + kx;
            int32_t r = -1;
            if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) r = (int32_t)(((long long)b * h + iy) * w + ix);
            nbr[(size_t)t * n_out + j] = r;
        }
}

static int valid_shape(int batch, const int32_t s[3]) {
    if (batch <= 0 || !s || s[0] <= 0 || s[1] <= 0 || s[2] <= 0) return 0;
    long long cells = (long long)batch * s[0] * s[1] * s[2];
    return cells < (1ll << 40);
}

static int scan_bitmap(const IndexView &v, int32_t *total_out, int32_t cap, hipStream_t s) {
    return device_scan(v.words, PopcFn{v.bitmap}, StoreBaseFn{v.base}, v.bsum, total_out, cap, s);
}

}  // namespace

extern "C" size_t cpd_index_bytes(int batch, const int32_t shape_zyx[3], int n_capacity) {
    if (!valid_shape(batch, shape_zyx) || n_capacity < 0) return 0;
    return index_carve(nullptr, batch, shape_zyx, n_capacity).bytes;
}

extern "C" int cpd_index_build(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3], void *index,
                               size_t index_bytes, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || !index || (n > 0 && !indices)) return CPD_ERR_ARG;
    IndexView v = index_carve(index, batch, shape_zyx, n);
    if (index_bytes < v.bytes) return CPD_ERR_WORKSPACE;
    hipStream_t s = cpd_s(stream);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    if (cpd_zero_fill(v.bitmap, (size_t)v.words * 8, s)) return CPD_ERR_LAUNCH;
    CPD_HIP_TRY(hipMemsetAsync(v.flags, 0, 4, s));
    CPD_HIP_TRY(hipMemsetAsync(v.flags, 1, 1, s));  // flags[0] = 1 (little endian): perm in use
    if (n > 0) index_mark_kernel<<<cpd_div_up(n, 256), 256, 0, s>>>(indices, n, g, v.bitmap);
    int rc = scan_bitmap(v, nullptr, -1, s);
    if (rc) return rc;
    if (n > 0) index_perm_kernel<<<cpd_div_up(n, 256), 256, 0, s>>>(indices, n, g, v.bitmap, v.base, v.perm);
    return cpd_check_launch();
}

// ---- row order of a level ------------------------------------------------------------------------------------------------
// The conv kernels skip a (16-row group, tap) pair when no row of the group has a neighbour at the tap; in canonical (b,z,y,x)
// order a group mixes rows with different neighbour patterns and 30-55 % of the executed MFMAs multiply zeros. Sorting the rows
// of every CHUNK of consecutive canonical rows by their kv-bit neighbour pattern makes the groups nearly uniform (executed /
// useful 1.33-1.55 -> 1.06-1.12 on the Waymo-shape levels) while a chunk's rows stay within the same few thousand rows, so the
// gathers keep their cache locality.
// One workgroup per chunk: a stable LSD radix sort (BlockSort27 below) of the 27-bit
// patterns with the local row as payload -- equal patterns keep their canonical order, the result is deterministic. (A bitonic
// network does log^2 work: 1800 lane-operations per row at 8192 rows, 100+ us on the one CU a chunk lives on; the radix passes
// need ~ 250.)
// Block-wide STABLE sort of T * E (27-bit key, payload) pairs, hand-written (round 4; rocPRIM's block_radix_sort did this until
// round 3 -- the one vendor primitive of the hot path). LSD radix, four passes of 7 bits. Items live in registers in "wave-striped"
// order: wave w owns positions [w * 64 E, (w + 1) * 64 E), slot e of lane l is position (w * E + e) * 64 + l -- so walking the slots
// of a wave in order IS walking its positions in order. One pass: per slot the lanes of a wave with the same digit find each other
// with seven ballots (peer mask), a lane's rank among them is a popcount, the wave's running count per digit sits in LDS
// (hist[digit][wave]: only this wave touches its column -- no atomics); an exclusive scan over (digit major, wave minor) turns the
// counts into bases; position = base + count before this slot + rank among peers: equal digits keep their order. Keys and payloads
// go through one LDS buffer (scatter, barrier, read back in wave-striped order).
template <int T, int E>
struct BlockSort27 {
    static constexpr int W = T / 64, N = T * E;
    struct Storage {
        uint32_t key[N], val[N];
        uint32_t hist[128 * W + 1];
        uint32_t scan_tmp[17];
    };
    __device__ static void sort(uint32_t (&key)[E], uint32_t (&val)[E], Storage &sm) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int shift = 0; shift < 27; shift += 7) {
            for (int i = tid; i < 128 * W; i += T) sm.hist[i] = 0;
            __syncthreads();
            uint32_t rank[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t d = (key[e] >> shift) & 127u;
                unsigned long long peers = ~0ull;
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const unsigned long long m = __ballot((d >> b) & 1u);
                    peers &= ((d >> b) & 1u) ? m : ~m;
                }
                const uint32_t before = sm.hist[d * W + wave];                  // this wave's earlier slots with digit d
                rank[e] = before + (uint32_t)__popcll(peers & lt);
                __builtin_amdgcn_wave_barrier();
                if ((peers & lt) == 0ull) sm.hist[d * W + wave] = before + (uint32_t)__popcll(peers);     // the peers' lowest lane
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            // exclusive scan of hist[128 * W] (digit major, wave minor): T threads, (128 W) / T entries each
            constexpr int PER = (128 * W + T - 1) / T;
            uint32_t loc[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = tid * PER + k;
                loc[k] = i < 128 * W ? sm.hist[i] : 0u;
                sum += loc[k];
            }
            uint32_t tot;
            uint32_t ex = block_excl_scan(sum, sm.scan_tmp, &tot);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = tid * PER + k;
                if (i < 128 * W) sm.hist[i] = ex;
                ex += loc[k];
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t d = (key[e] >> shift) & 127u;
                const uint32_t pos = sm.hist[d * W + wave] + rank[e];
                sm.key[pos] = key[e];
                sm.val[pos] = val[e];
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = (wave * E + e) * 64 + lane;
                key[e] = sm.key[i];
                val[e] = sm.val[i];
            }
            __syncthreads();
        }
    }
};

// `pre_n2o` (optional): the list being sorted is itself a re-ordering of the canonical list (cpd_order_rows_bricks): position ->
// canonical row; new_to_old / old_to_new are then written against the canonical rows.
template <int T, int E>
__global__ void __launch_bounds__(T) order_rows_kernel(const uint32_t *__restrict__ pattern, const int32_t *__restrict__ coords, int n,
                                                       int32_t *__restrict__ new_to_old, int32_t *__restrict__ old_to_new,
                                                       int32_t *__restrict__ coords_out, const int32_t *__restrict__ pre_n2o = nullptr) {
    using sort_t = BlockSort27<T, E>;
    __shared__ typename sort_t::Storage storage;
    const int c0 = blockIdx.x * (T * E), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t key[E], val[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = (wave * E + e) * 64 + lane, row = c0 + i;          // wave-striped: position i of the chunk
        key[e] = row < n ? (pattern[row] & 0x7ffffffu) : 0x7ffffffu;      // padding rows: largest key, and last among equals
        val[e] = (unsigned)i;
    }
    sort_t::sort(key, val, storage);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int row = c0 + (wave * E + e) * 64 + lane;
        if (row < n) {
            const int old = c0 + (int)val[e];
            const int canon = pre_n2o ? pre_n2o[old] : old;
            new_to_old[row] = canon;
            old_to_new[canon] = row;
            if (coords_out) reinterpret_cast<int4 *>(coords_out)[row] = reinterpret_cast<const int4 *>(coords)[old];
        }
    }
}

// ---- brick order (round 4) ---------------------------------------------------------------------------------------------------
// The staged row-wave kernel fetches a tile's DISTINCT input rows once; how many there are depends on how close in space the tile's
// 128 rows are. Canonical (b, z, y, x) order makes a tile a run of whole x-lines -- in a dense plane ONE line segment, whose 3 x 3
// neighbourhood in the plane is three segments: 3 distinct rows per row and plane. Rows ordered by (b, z, y / BY, x / BX, y, x) --
// BY x BX bricks of one z-plane -- give a tile a compact 2-D footprint: 2.9 distinct rows per output row instead of 4.5 on the
// Waymo-shape levels (tools/unique_probe2.py; 8 x 8 bricks), never more than 205 per dz group.
// No sort: a row's position is a sum of RANK queries on the level's canonical bitmap index --
//   rows before its band (the BY lines y0 .. y0 + BY - 1 of its plane)
// + rows of the band left of its brick                       (per line: rank(line, xb0) - rank(line, 0))
// + rows of its brick in earlier lines                       (per line: rank(line, xb1) - rank(line, xb0))
// + rows of its own line inside the brick before it          (its own rank - rank(line, xb0)).
__device__ __forceinline__ uint32_t rank_before(const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base, long long key,
                                                long long cells, uint32_t n_total) {
    if (key >= cells) return n_total;
    const uint64_t w = bitmap[key >> 6];
    return base[key >> 6] + (uint32_t)__popcll(w & ((1ull << (key & 63)) - 1ull));
}

__global__ void __launch_bounds__(256) brick_position_kernel(const int32_t *__restrict__ idx, int n, Grid g, int by, int bx,
                                                             const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                                                             int32_t *__restrict__ pos_to_row, int32_t *__restrict__ coords_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;            // canonical row = its rank
    if (j >= n) return;
    const int4 q = reinterpret_cast<const int4 *>(idx)[j];         // (b, z, y, x)
    const long long cells = (long long)g.b * g.d * g.h * g.w;
    const int y0 = (q.z / by) * by, y1 = y0 + by < g.h ? y0 + by : g.h;
    const int xb0 = (q.w / bx) * bx, xb1 = xb0 + bx;
    uint32_t line0 = rank_before(bitmap, base, g.key(q.x, q.y, y0, 0), cells, (uint32_t)n);
    uint32_t pos = line0;
    for (int yy = y0; yy < y1; ++yy) {
        const uint32_t next0 = rank_before(bitmap, base, g.key(q.x, q.y, yy, 0) + g.w, cells, (uint32_t)n);     // start of the next line
        const uint32_t a = xb0 > 0 ? rank_before(bitmap, base, g.key(q.x, q.y, yy, xb0), cells, (uint32_t)n) : line0;
        pos += a - line0;
        if (yy < q.z) pos += (xb1 < g.w ? rank_before(bitmap, base, g.key(q.x, q.y, yy, xb1), cells, (uint32_t)n) : next0) - a;
        else if (yy == q.z) pos += (uint32_t)j - a;
        line0 = next0;
    }
    pos_to_row[pos] = j;
    reinterpret_cast<int4 *>(coords_out)[pos] = q;
}

__global__ void index_set_order_kernel(int32_t *flags, const int32_t *rank_to_row) {
    *reinterpret_cast<const int32_t **>(flags + 2) = rank_to_row;
    flags[0] = rank_to_row ? 2 : 0;
}

extern "C" int cpd_order_rows_by_taps(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3], const int32_t ksize[3],
                                      const void *index, int chunk_rows, int32_t *new_to_old, int32_t *old_to_new,
                                      int32_t *indices_out, void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || !ksize || !index || (n > 0 && (!indices || !new_to_old || !old_to_new || !workspace)))
        return CPD_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (ksize[d] <= 0 || !(ksize[d] & 1)) return CPD_ERR_ARG;
    if (ksize[0] * ksize[1] * ksize[2] > 32 || ksize[2] > 32) return CPD_ERR_UNSUPPORTED;
    if (chunk_rows != 1024 && chunk_rows != 4096 && chunk_rows != 8192 && chunk_rows != 16384) return CPD_ERR_UNSUPPORTED;
    if (workspace_bytes < (size_t)(n > 0 ? n : 1) * 4) return CPD_ERR_WORKSPACE;
    if (n == 0) return CPD_OK;
    hipStream_t s = cpd_s(stream);
    uint32_t *pattern = static_cast<uint32_t *>(workspace);
    IndexView v = index_carve(const_cast<void *>(index), batch, shape_zyx, 1);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    // pattern pass: the rulebook walk without the table (ranks are not even resolved to rows)
    if (ksize[0] == 3 && ksize[1] == 3 && ksize[2] == 3)
        rulebook_kernel<true><<<cpd_div_up(n, 256), 256, 0, s>>>(indices, n, g, 3, 3, 3, 1, 1, 1, 1, 1, 1, v.bitmap, v.base, v.perm, v.flags,
                                                                 nullptr, nullptr, pattern);
    else
        rulebook_kernel<false><<<cpd_div_up(n, 256), 256, 0, s>>>(indices, n, g, ksize[0], ksize[1], ksize[2], 1, 1, 1, ksize[0] / 2,
                                                                  ksize[1] / 2, ksize[2] / 2, v.bitmap, v.base, v.perm, v.flags, nullptr,
                                                                  nullptr, pattern);
    const int blocks = cpd_div_up(n, chunk_rows);
    if (chunk_rows == 1024) order_rows_kernel<256, 4><<<blocks, 256, 0, s>>>(pattern, indices, n, new_to_old, old_to_new, indices_out);
    else if (chunk_rows == 4096) order_rows_kernel<1024, 4><<<blocks, 1024, 0, s>>>(pattern, indices, n, new_to_old, old_to_new, indices_out);
    else if (chunk_rows == 8192) order_rows_kernel<1024, 8><<<blocks, 1024, 0, s>>>(pattern, indices, n, new_to_old, old_to_new, indices_out);
    else order_rows_kernel<1024, 16><<<blocks, 1024, 0, s>>>(pattern, indices, n, new_to_old, old_to_new, indices_out);
    return cpd_check_launch();
}


extern "C" int cpd_order_rows_bricks(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3], const void *index,
                                     int brick_y, int brick_x, int tile_rows, int32_t *new_to_old, int32_t *old_to_new, int32_t *indices_out,
                                     void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    if (tile_rows != 128 && tile_rows != 256) return CPD_ERR_UNSUPPORTED;
    if (!valid_shape(batch, shape_zyx) || n < 0 || !index || brick_y <= 0 || brick_x <= 0 ||
        (n > 0 && (!indices || !new_to_old || !old_to_new || !indices_out || !workspace)))
        return CPD_ERR_ARG;
    const size_t n1 = (size_t)(n > 0 ? n : 1);
    if (workspace_bytes < cpd_align(n1 * 4) * 2 + cpd_align(n1 * 16)) return CPD_ERR_WORKSPACE;
    if (n == 0) return CPD_OK;
    hipStream_t s = cpd_s(stream);
    char *ws = static_cast<char *>(workspace);
    uint32_t *pattern = reinterpret_cast<uint32_t *>(ws);
    int32_t *pos_to_row = reinterpret_cast<int32_t *>(ws + cpd_align(n1 * 4));
    int32_t *coords_b = reinterpret_cast<int32_t *>(ws + 2 * cpd_align(n1 * 4));
    IndexView v = index_carve(const_cast<void *>(index), batch, shape_zyx, 1);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    // (the index must be the CANONICAL one of `indices`: row = rank)
    brick_position_kernel<<<cpd_div_up(n, 256), 256, 0, s>>>(indices, n, g, brick_y, brick_x, v.bitmap, v.base, pos_to_row, coords_b);
    // then, inside every tile of 128 rows of that order, rows sorted by their 27-bit neighbour pattern (what cpd_order_rows_by_taps
    // does per 4096-row chunk): the kernels' 16-row tap skipping gets nearly uniform groups, the tile keeps its footprint
    rulebook_kernel<true><<<cpd_div_up(n, 256), 256, 0, s>>>(coords_b, n, g, 3, 3, 3, 1, 1, 1, 1, 1, 1, v.bitmap, v.base, v.perm, v.flags,
                                                             nullptr, nullptr, pattern);
    if (tile_rows == 128) order_rows_kernel<64, 2><<<cpd_div_up(n, 128), 64, 0, s>>>(pattern, coords_b, n, new_to_old, old_to_new, indices_out, pos_to_row);
    else order_rows_kernel<64, 4><<<cpd_div_up(n, 256), 64, 0, s>>>(pattern, coords_b, n, new_to_old, old_to_new, indices_out, pos_to_row);
    return cpd_check_launch();
}

extern "C" int cpd_index_set_order(void *index, const int32_t *rank_to_row, cpd_stream_t stream) {
    if (!index) return CPD_ERR_ARG;
    index_set_order_kernel<<<1, 1, 0, cpd_s(stream)>>>(static_cast<int32_t *>(index), rank_to_row);     // flags lead the buffer
    return cpd_check_launch();
}

extern "C" int cpd_conv_out_shape(const int32_t in_shape[3], const int32_t ksize[3], const int32_t stride[3],
                                  const int32_t pad[3], int32_t out_shape[3]) {
    if (!in_shape || !ksize || !stride || !pad || !out_shape) return CPD_ERR_ARG;
    for (int d = 0; d < 3; ++d) {
        if (ksize[d] <= 0 || stride[d] <= 0 || pad[d] < 0) return CPD_ERR_ARG;
        out_shape[d] = (in_shape[d] + 2 * pad[d] - ksize[d]) / stride[d] + 1;
        if (out_shape[d] <= 0) return CPD_ERR_ARG;
    }
    return CPD_OK;
}

static int rulebook_launch(const int32_t *out_idx, int n_out, int batch, const int32_t in_shape[3], const int32_t k[3],
                           const int32_t st[3], const int32_t pd[3], const void *index, int32_t *nbr, uint32_t *tapmask,
                           hipStream_t s) {
    if (k[2] > 32) return CPD_ERR_UNSUPPORTED;       // a row's kw taps must fit two bitmap words (rulebook_kernel's per-row word cache)
    // the index was carved with some capacity; pointers before perm do not depend on it
    IndexView v = index_carve(const_cast<void *>(index), batch, in_shape, 1);
    Grid g{batch, in_shape[0], in_shape[1], in_shape[2]};
    if (n_out > 0 && k[0] == 3 && k[1] == 3 && k[2] == 3)
        rulebook_kernel<true><<<cpd_div_up(n_out, 256), 256, 0, s>>>(out_idx, n_out, g, 3, 3, 3, st[0], st[1], st[2], pd[0], pd[1], pd[2],
                                                                     v.bitmap, v.base, v.perm, v.flags, nbr, tapmask, nullptr);
    else if (n_out > 0)
        rulebook_kernel<false><<<cpd_div_up(n_out, 256), 256, 0, s>>>(out_idx, n_out, g, k[0], k[1], k[2], st[0], st[1], st[2],
                                                                      pd[0], pd[1], pd[2], v.bitmap, v.base,
                                                                      v.perm, v.flags, nbr,
                                                                      (k[0] * k[1] * k[2] <= 32) ? tapmask : nullptr, nullptr);
    return cpd_check_launch();
}

extern "C" int cpd_rulebook_subm(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3],
                                 const int32_t ksize[3], const void *index, int32_t *nbr, uint32_t *tapmask,
                                 cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || !ksize || !index || (n > 0 && (!indices || !nbr))) return CPD_ERR_ARG;
    for (int d = 0; d < 3; ++d)
        if (ksize[d] <= 0 || !(ksize[d] & 1)) return CPD_ERR_ARG;
    const int32_t one[3] = {1, 1, 1};
    const int32_t pad[3] = {ksize[0] / 2, ksize[1] / 2, ksize[2] / 2};
    return rulebook_launch(indices, n, batch, shape_zyx, ksize, one, pad, index, nbr, tapmask, cpd_s(stream));
}

extern "C" int cpd_rulebook_conv(const int32_t *out_indices, int n_out, int batch, const int32_t in_shape[3],
                                 const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                                 const void *in_index, int32_t *nbr, uint32_t *tapmask, cpd_stream_t stream) {
    int32_t os[3];
    if (!valid_shape(batch, in_shape) || n_out < 0 || !in_index || (n_out > 0 && (!out_indices || !nbr)))
        return CPD_ERR_ARG;
    int rc = cpd_conv_out_shape(in_shape, ksize, stride, pad, os);
    if (rc) return rc;
    return rulebook_launch(out_indices, n_out, batch, in_shape, ksize, stride, pad, in_index, nbr, tapmask, cpd_s(stream));
}


// The same tables as cpd_rulebook_subm / cpd_rulebook_conv for a CHUNK-ORDERED output level: `out_canonical` [n_out][4] is the level's
// canonical site list and `out_old_to_new` [n_out] its order (canonical row -> row; rows move only inside chunks of 4096 canonical
// rows: cpd_order_rows_by_taps with chunk_rows = 4096; NULL = canonical). Rows of nbr / tapmask are in the NEW order, bit for bit what
// the plain builders give over the re-ordered list -- built by walking each chunk in canonical order (coalesced index reads) and
// re-ordering it in LDS (rulebook_chunk_kernel). 3 x 3 x 3 kernels only (else CPD_ERR_UNSUPPORTED).
extern "C" int cpd_rulebook_chunk_ordered(const int32_t *out_canonical, const int32_t *out_old_to_new, int n_out, int batch,
                                          const int32_t in_shape[3], const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                                          const void *in_index, int chunk_rows, int32_t *nbr, uint32_t *tapmask, cpd_stream_t stream) {
    if (!valid_shape(batch, in_shape) || n_out < 0 || !in_index || !ksize || !stride || !pad || (n_out > 0 && (!out_canonical || !nbr)))
        return CPD_ERR_ARG;
    if (ksize[0] != 3 || ksize[1] != 3 || ksize[2] != 3 || chunk_rows != CPD_RBC_CHUNK) return CPD_ERR_UNSUPPORTED;
    if (n_out == 0) return CPD_OK;
    IndexView v = index_carve(const_cast<void *>(in_index), batch, in_shape, 1);
    Grid g{batch, in_shape[0], in_shape[1], in_shape[2]};
    rulebook_chunk_kernel<<<cpd_div_up(n_out, CPD_RBC_CHUNK), 1024, 0, cpd_s(stream)>>>(out_canonical, out_old_to_new, n_out, g, stride[0], stride[1],
                                                                                         stride[2], pad[0], pad[1], pad[2], v.bitmap, v.base, v.perm,
                                                                                         v.flags, nbr, tapmask);
    return cpd_check_launch();
}

extern "C" int cpd_conv_outset(const int32_t *in_indices, int n_in, int batch, const int32_t in_shape[3],
                               const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3], void *out_index,
                               size_t out_index_bytes, int32_t *n_out, cpd_stream_t stream) {
    int32_t os[3];
    if (!valid_shape(batch, in_shape) || n_in < 0 || !out_index || !n_out || (n_in > 0 && !in_indices)) return CPD_ERR_ARG;
    int rc = cpd_conv_out_shape(in_shape, ksize, stride, pad, os);
    if (rc) return rc;
    if (!valid_shape(batch, os)) return CPD_ERR_UNSUPPORTED;
    IndexView v = index_carve(out_index, batch, os, 0);
    if (out_index_bytes < v.bytes) return CPD_ERR_WORKSPACE;
    hipStream_t s = cpd_s(stream);
    Grid g{batch, os[0], os[1], os[2]};
    if (cpd_zero_fill(v.bitmap, (size_t)v.words * 8, s)) return CPD_ERR_LAUNCH;
    CPD_HIP_TRY(hipMemsetAsync(v.flags, 0, 4, s));  // canonical: rank == row id, perm unused
    if (ksize[2] > 32) return CPD_ERR_UNSUPPORTED;            // (a site's x' cells must fit two bitmap words)
    if (n_in > 0)
        outset_mark_kernel<<<cpd_div_up(n_in, 256), 256, 0, s>>>(in_indices, n_in, g, ksize[0], ksize[1], ksize[2],
                                                                 stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
                                                                 v.bitmap);
    return scan_bitmap(v, n_out, -1, s);
}

extern "C" int cpd_index_emit(const void *index, int batch, const int32_t shape_zyx[3], int32_t *indices, int n_capacity,
                              cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || !index || n_capacity < 0 || (n_capacity > 0 && !indices)) return CPD_ERR_ARG;
    IndexView v = index_carve(const_cast<void *>(index), batch, shape_zyx, 0);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    hipStream_t s = cpd_s(stream);
    long long nthreads = v.words;
    auto fn = EmitFn{v.bitmap, v.base, indices, n_capacity, g};
    emit_words_kernel<<<cpd_div_up(nthreads, 256), 256, 0, s>>>(nthreads, fn);
    return cpd_check_launch();
}

extern "C" int cpd_densify_nhwc(const float *feat, const int32_t *indices, int n, int c, int batch,
                                const int32_t shape_zyx[3], float *out, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || c <= 0 || (c & 3) || !out || (n > 0 && (!feat || !indices)))
        return CPD_ERR_ARG;
    hipStream_t s = cpd_s(stream);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    size_t total = (size_t)batch * shape_zyx[0] * shape_zyx[1] * shape_zyx[2] * c;
    if (cpd_zero_fill(out, total * sizeof(float), s)) return CPD_ERR_LAUNCH;      // (a grid-stride loop of 16-byte stores: the memory system's write rate)
    long long threads = (long long)n * (c / 4);
    if (threads > 0) densify_nhwc_kernel<<<cpd_div_up(threads, 256), 256, 0, s>>>(feat, indices, n, c / 4, g, out);
    return cpd_check_launch();
}

// cpd_densify_nhwc without the clear: `out` is a caller-owned map that IS all zero (a persistent buffer); the occupied rows are
// scattered into it, and cpd_densify_nhwc_clear(indices ...) -- queued after the map's last reader -- puts the zeros back by writing the
// same rows: 2 x (sites x C) floats moved instead of the whole (B, H, W, D * C) map (12 % occupied at the stride-8 level of a Waymo frame).
extern "C" int cpd_densify_nhwc_rows(const float *feat, const int32_t *indices, int n, int c, int batch,
                                     const int32_t shape_zyx[3], float *out, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || c <= 0 || (c & 3) || !out || (n > 0 && (!feat || !indices)))
        return CPD_ERR_ARG;
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    long long threads = (long long)n * (c / 4);
    if (threads > 0) densify_nhwc_kernel<<<cpd_div_up(threads, 256), 256, 0, cpd_s(stream)>>>(feat, indices, n, c / 4, g, out);
    return cpd_check_launch();
}
extern "C" int cpd_densify_nhwc_clear(const int32_t *indices, int n, int c, int batch, const int32_t shape_zyx[3], float *out,
                                      cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || c <= 0 || (c & 3) || !out || (n > 0 && !indices)) return CPD_ERR_ARG;
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    long long threads = (long long)n * (c / 4);
    if (threads > 0) densify_clear_kernel<<<cpd_div_up(threads, 256), 256, 0, cpd_s(stream)>>>(indices, n, c / 4, g, out);
    return cpd_check_launch();
}

extern "C" int cpd_densify_nhwc_cd(const float *feat, const int32_t *indices, int n, int c, int batch,
                                   const int32_t shape_zyx[3], float *out, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || c <= 0 || !out || (n > 0 && (!feat || !indices))) return CPD_ERR_ARG;
    hipStream_t s = cpd_s(stream);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    size_t total = (size_t)batch * shape_zyx[0] * shape_zyx[1] * shape_zyx[2] * c;
    if (cpd_zero_fill(out, total * sizeof(float), s)) return CPD_ERR_LAUNCH;      // (a grid-stride loop of 16-byte stores: the memory system's write rate)
    long long threads = (long long)n * c;
    if (threads > 0) densify_nhwc_cd_kernel<<<cpd_div_up(threads, 256), 256, 0, s>>>(feat, indices, n, c, g, out);
    return cpd_check_launch();
}

extern "C" int cpd_densify_nchw(const float *feat, const int32_t *indices, int n, int c, int batch,
                                const int32_t shape_zyx[3], float *out, cpd_stream_t stream) {
    if (!valid_shape(batch, shape_zyx) || n < 0 || c <= 0 || !out || (n > 0 && (!feat || !indices))) return CPD_ERR_ARG;
    hipStream_t s = cpd_s(stream);
    Grid g{batch, shape_zyx[0], shape_zyx[1], shape_zyx[2]};
    size_t total = (size_t)batch * shape_zyx[0] * shape_zyx[1] * shape_zyx[2] * c;
    if (cpd_zero_fill(out, total * sizeof(float), s)) return CPD_ERR_LAUNCH;      // (a grid-stride loop of 16-byte stores: the memory system's write rate)
    long long threads = (long long)n * c;
    if (threads > 0) densify_nchw_kernel<<<cpd_div_up(threads, 256), 256, 0, s>>>(feat, indices, n, c, g, out);
    return cpd_check_launch();
}

extern "C" int cpd_rulebook_conv2d(int batch, int h, int w, int kh, int kw, int stride, int pad, int32_t *nbr,
                                   cpd_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || !nbr) return CPD_ERR_ARG;
    int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (ho <= 0 || wo <= 0) return CPD_ERR_ARG;
    long long n_out = (long long)batch * ho * wo;
    if (n_out >= (1ll << 31)) return CPD_ERR_UNSUPPORTED;
    rulebook_conv2d_kernel<<<cpd_div_up(n_out, 256), 256, 0, cpd_s(stream)>>>(batch, h, w, ho, wo, kh, kw, stride, pad, nbr);
    return cpd_check_launch();
}
