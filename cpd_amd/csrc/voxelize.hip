// voxelize.hip -- point cloud -> voxels (+ fused MeanVFE) on gfx950.
//
// Replaces VoxelGeneratorWrapper.generate -> [SPCONV] Point2VoxelCPU3d.point_to_voxel
// (cpd/datasets/processor/data_processor.py:14-59) and MeanVFE (mean_vfe.py:41-43).
//
// The reference is a SERIAL scan over points with a dense int32 lookup volume; voxel ids follow
// first appearance and each voxel keeps its first P points in point order. The same result is
// produced here data-parallel, with no sort and no hash probing:
//   1 keys      : cell key per point (fp32 floor((p-lo)/vs), IEEE division), occupancy bit set
//                 in a dense bitmap over the grid (1 bit per cell; 92.7 M cells = 11.6 MB for the
//                 Waymo grid -- trivially resident in 288 GB HBM / 256 MB Infinity Cache)
//   2 scan      : popcount prefix over the bitmap  -> dense rank r of every occupied cell
//   3 first     : first[r] = min point index (atomicMin)
//   4 order     : flag(i) = (first[rank_i] == i); prefix sum over POINT order of the flags gives
//                 the serial first-appearance voxel id; ids >= max_voxels are dropped exactly as
//                 the serial scan would
//   5 insert    : per voxel, the P smallest point indices via an atomicMin insertion cascade
//   6 gather    : coalesced copy of the kept points into voxels[M,P,C], count, mean
// HBM traffic is ~ the points read 3x (12 B/pt keys + rows) + outputs; see DESIGN.md.
#include "common.h"
#include <string.h>
#include "site_index_layout.h"

thread_local int g_cpd_last_hip_error = 0;

extern "C" const char *cpd_version(void) { return "cpd_hip 0.1 (gfx950)"; }
extern "C" int cpd_last_hip_error(void) { return g_cpd_last_hip_error; }

// ---- launch log (diagnostics: which kernel instantiation served a call; tests assert the names, nothing else reads it) ----
#include <map>
#include <mutex>
#include <string>
static std::mutex g_log_mutex;
static std::map<std::string, int> g_log;
static bool g_log_on = false;
extern "C" void cpd_launch_log_enable(int on) {
    std::lock_guard<std::mutex> lk(g_log_mutex);
    g_log_on = on != 0;
    g_log.clear();
}
extern "C" void cpd_launch_log_note(const char *kernel) {
    if (!g_log_on || !kernel) return;
    std::lock_guard<std::mutex> lk(g_log_mutex);
    ++g_log[kernel];
}
extern "C" size_t cpd_launch_log_dump(char *buf, size_t cap) {      // "name count\n" lines; returns the bytes needed (incl. the NUL)
    std::lock_guard<std::mutex> lk(g_log_mutex);
    std::string out;
    for (const auto &kv : g_log) out += kv.first + " " + std::to_string(kv.second) + "\n";
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return out.size() + 1;
}

extern "C" int cpd_voxel_grid_size(const float vsize_xyz[3], const float range_xyz[6], int32_t grid_zyx[3]) {
    if (!vsize_xyz || !range_xyz || !grid_zyx) return CPD_ERR_ARG;
    for (int a = 0; a < 3; ++a) {
        float g = (range_xyz[3 + a] - range_xyz[a]) / vsize_xyz[a];
        grid_zyx[2 - a] = (int32_t)roundf(g);
        if (grid_zyx[2 - a] <= 0) return CPD_ERR_ARG;
    }
    return CPD_OK;
}

namespace {

struct VoxGeom {
    float lo[3];   // x,y,z lower bounds
    float vs[3];   // x,y,z voxel size
    int32_t g[3];  // z,y,x grid
};

struct VoxWs {
    uint64_t *bitmap;   // [words]
    uint32_t *base;     // [words] popcount prefix
    uint32_t *bsum_bm;  // scan block sums (bitmap)
    uint32_t *bsum_pt;  // scan block sums (points)
    int32_t *pkey;      // [n] cell key or -1
    int32_t *prank;     // [n] dense rank of the point's cell
    int32_t *first;     // [n] min point index per rank
    int32_t *vid;       // [n] voxel id per rank
    int32_t *slots;     // [cap*P] kept point indices per voxel (sorted ascending)
    int32_t *counts;    // [cap] points seen per voxel
    int32_t *nocc;      // scalar: occupied cells
    long long words;
    size_t bytes;
};

static VoxWs carve(void *ws, int n, int P, int cap, long long cells) {
    VoxWs w;
    size_t off = 0;
    char *b = (char *)ws;
    auto take = [&](size_t bytes) {
        void *p = b ? (void *)(b + off) : nullptr;
        off += cpd_align(bytes);
        return p;
    };
    w.words = (cells + 63) / 64;
    w.bitmap = (uint64_t *)take((size_t)w.words * 8);
    w.base = (uint32_t *)take((size_t)w.words * 4);
    w.bsum_bm = (uint32_t *)take((size_t)scan_num_blocks(w.words) * 4);
    w.bsum_pt = (uint32_t *)take((size_t)scan_num_blocks(n) * 4);
    size_t nn = (size_t)(n > 0 ? n : 1);
    w.pkey = (int32_t *)take(nn * 4);
    w.prank = (int32_t *)take(nn * 4);
    w.first = (int32_t *)take(nn * 4);
    w.vid = (int32_t *)take(nn * 4);
    w.slots = (int32_t *)take((size_t)(cap > 0 ? cap : 1) * P * 4);
    w.counts = (int32_t *)take((size_t)(cap > 0 ? cap : 1) * 4);
    w.nocc = (int32_t *)take(4);
    w.bytes = off;
    return w;
}

__global__ void __launch_bounds__(256) vox_keys_kernel(const float *__restrict__ pts, int n, int c, VoxGeom geo,
                                                       int32_t *__restrict__ pkey, uint64_t *bitmap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pts + (size_t)i * c;
    int32_t cz[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // j over z,y,x ; coordinate axis 2-j
        // fp32 subtract then IEEE-correct fp32 divide then floor: bit-identical to the serial CPU
        // voxelizer at voxel boundaries (no reciprocal multiply, no contraction possible here).
        float f = floorf(__fdiv_rn(__fsub_rn(p[2 - j], geo.lo[2 - j]), geo.vs[2 - j]));
        if (!(f >= 0.0f) || !(f < (float)geo.g[j])) ok = false;
        cz[j] = ok ? (int32_t)f : 0;
    }
    int32_t key = -1;
    if (ok) {
        key = (cz[0] * geo.g[1] + cz[1]) * geo.g[2] + cz[2];
        atomicOr((unsigned long long *)&bitmap[key >> 6], 1ull << (key & 63));
    }
    pkey[i] = key;
}

__global__ void __launch_bounds__(256) vox_first_kernel(int n, const int32_t *__restrict__ pkey,
                                                        const uint64_t *__restrict__ bitmap,
                                                        const uint32_t *__restrict__ base, int32_t *__restrict__ prank,
                                                        int32_t *first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t key = pkey[i];
    int32_t r = -1;
    const int32_t prev = __shfl_up(key, 1);     // (all lanes of a wave reach this: the early return above is per block tail only)
    if (key >= 0) {
        uint64_t w = bitmap[key >> 6];
        r = (int32_t)(base[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1ull)));
        // a lane whose left neighbour has the same key cannot hold the voxel's first point
        if ((threadIdx.x & 63) == 0 || prev != key) atomicMin(&first[r], i);
    }
    prank[i] = r;
}

struct FlagFn {  // 1 iff point i is the first point of its voxel
    const int32_t *prank;
    const int32_t *first;
    __device__ uint32_t operator()(long long i) const {
        int32_t r = prank[i];
        return (r >= 0 && first[r] == (int32_t)i) ? 1u : 0u;
    }
};
struct AssignVoxelFn {  // flagged point i starts voxel id = prefix (serial first-appearance order)
    const int32_t *prank;
    const int32_t *pkey;
    int32_t *vid;
    int32_t *coords;
    int coord_cols, batch_idx, max_voxels;
    int32_t gy, gx;
    __device__ void operator()(long long i, uint32_t flag, uint32_t prefix) const {
        if (!flag) return;
        int32_t v = (int32_t)prefix;
        vid[prank[i]] = v;
        if (v < max_voxels) {
            int32_t key = pkey[i];
            int32_t x = key % gx, y = (key / gx) % gy, z = key / (gx * gy);
            int32_t *o = coords + (size_t)v * coord_cols;
            if (coord_cols == 4) { o[0] = batch_idx; o[1] = z; o[2] = y; o[3] = x; }
            else { o[0] = z; o[1] = y; o[2] = x; }
        }
    }
};

struct PopcFn {
    const uint64_t *bitmap;
    __device__ uint32_t operator()(long long w) const { return (uint32_t)__popcll(bitmap[w]); }
};
struct StoreBaseFn {
    uint32_t *base;
    __device__ void operator()(long long w, uint32_t, uint32_t prefix) const { base[w] = prefix; }
};

__global__ void __launch_bounds__(256) vox_insert_kernel(int n, int P, int max_voxels,
                                                         const int32_t *__restrict__ prank,
                                                         const int32_t *__restrict__ vid, const int32_t *__restrict__ first,
                                                         int32_t *slots, int32_t *counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t r = prank[i];
    if (r < 0) return;
    int32_t v = vid[r];
    if (v >= max_voxels) return;
    atomicAdd(&counts[v], 1);
    // Insertion cascade: slot p ends up holding the (p+1)-th smallest point index of the voxel. Slot 0 is the voxel's first
    // point (known from vox_first: a plain store); every other value enters at slot 1, whatever loses an atomicMin moves on.
    int32_t x = i;
    int32_t *s = slots + (size_t)v * P;
    if (first[r] == i) { s[0] = i; return; }
    for (int p = 1; p < P; ++p) {
        int32_t old = atomicMin(&s[p], x);
        if (old == 0x7f7f7f7f) break;  // slot was empty: nothing displaced
        x = old > x ? old : x;
    }
}

__global__ void __launch_bounds__(256) vox_gather_kernel(const float *__restrict__ pts, int c, int P, int cap,
                                                         const int32_t *__restrict__ n_vox,
                                                         const int32_t *__restrict__ slots,
                                                         const int32_t *__restrict__ counts, float *voxels,
                                                         int32_t *num_points, float *mean) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int v = (int)(tid / c), ch = (int)(tid % c);
    if (v >= cap || v >= *n_vox) return;
    int cnt = counts[v];
    if (cnt > P) cnt = P;
    float s = 0.f;
    for (int p = 0; p < P; ++p) {  // sum order p = 0..P-1, zeros included (mean_vfe.py:41)
        float val = 0.f;
        if (p < cnt) val = pts[(size_t)slots[(size_t)v * P + p] * c + ch];
        if (voxels) voxels[((size_t)v * P + p) * c + ch] = val;
        s += val;
    }
    if (mean) mean[(size_t)v * c + ch] = __fdiv_rn(s, (float)(cnt < 1 ? 1 : cnt));
    if (ch == 0) num_points[v] = cnt;
}


// ------------------------------------------------------------------------------------------
// Batched voxelizer: all frames of a batch in ONE set of launches. Points are concatenated in frame
// order, a cell key carries the frame (key = frame * cells + cell) and the occupancy bitmap spans
// batch x grid; the first-appearance scan then runs over the concatenation, which keeps every
// frame's serial order, and a per-frame base turns global ids into the per-frame voxel ids the
// max_voxels cap applies to. Output rows are the frames' voxels back to back.
// ------------------------------------------------------------------------------------------
#define CPD_VOX_MAX_FRAMES 64
struct FrameOffsets {
    int32_t nf;
    int32_t off[CPD_VOX_MAX_FRAMES + 1];
    __device__ __forceinline__ int frame_of(int i) const {        // the last frame that starts at or before point i
        int lo = 0, hi = nf - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (i >= off[mid]) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    // the same for lane `lane` of a wave whose lanes hold consecutive points: the wave's first and last point are looked up on the
    // scalar unit; only a wave that straddles a frame boundary searches per lane
    // ... and for a point index that is NOT consecutive across the lanes but usually in one frame wave-wide (a voxel's first point, the
    // voxels of a wave being neighbours in row order): the first valid lane's frame, found on the scalar unit, is tried first
    __device__ __forceinline__ int frame_of_near(int i) const {
        const unsigned long long m = __ballot(i >= 0);
        int hint = 0;
        if (m) hint = frame_of(__builtin_amdgcn_readlane(i, __builtin_ctzll(m)));
        return (i >= off[hint] && i < off[hint + 1]) ? hint : (i >= 0 ? frame_of(i) : 0);
    }
    __device__ __forceinline__ int frame_of_lane(int i, int lane) const {
        const int i0 = __builtin_amdgcn_readfirstlane(i - lane);
        const int f0 = frame_of(i0), f1 = frame_of(i0 + 63);
        return f0 == f1 ? f0 : frame_of(i);
    }
};

// Where the rows of the batch's (virtually concatenated) point list live: one base pointer per frame. A caller with ONE contiguous
// buffer gives base + off[f] * c (cpd_voxelize_batch*); a caller whose frames sit in separate allocations -- the usual case: clouds
// arrive one by one -- gives them as they are (cpd_voxelize_batch_frames) and no concatenation pass (153 MB at 48 frames) is needed.
struct FramePts {
    const float *p[CPD_VOX_MAX_FRAMES];
    __device__ __forceinline__ const float *row(const FrameOffsets &fo, int f, int i, int c) const { return p[f] + (size_t)(i - fo.off[f]) * c; }
};

// (the batched gather: slot entries are indices into the batch's virtual concatenation; a voxel's points all sit in its frame)
__global__ void __launch_bounds__(256) vox_gather_frames_kernel(FramePts pts, FrameOffsets fo, int c, int P, int cap,
                                                                const int32_t *__restrict__ n_vox, const int32_t *__restrict__ slots,
                                                                const int32_t *__restrict__ counts, float *voxels, int32_t *num_points,
                                                                float *mean) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int v = (int)(tid / c), ch = (int)(tid % c);
    if (v >= cap || v >= *n_vox) return;
    int cnt = counts[v];
    if (cnt > P) cnt = P;
    const int frame = fo.frame_of_near(cnt > 0 ? slots[(size_t)v * P] : -1);
    float s = 0.f;
    for (int p = 0; p < P; ++p) {  // sum order p = 0..P-1, zeros included (mean_vfe.py:41)
        float val = 0.f;
        if (p < cnt) val = pts.row(fo, frame, slots[(size_t)v * P + p], c)[ch];
        if (voxels) voxels[((size_t)v * P + p) * c + ch] = val;
        s += val;
    }
    if (mean) mean[(size_t)v * c + ch] = __fdiv_rn(s, (float)(cnt < 1 ? 1 : cnt));
    if (ch == 0) num_points[v] = cnt;
}

// Batched form: pkey holds the point's cell INSIDE its frame (< cells, 31 bits); the bitmap position frame * cells + cell is
// formed in 64 bits where it is needed (the frame of a point follows from its index), so a batch is bounded by the 64 frames
// of FrameOffsets, not by 2^31 cells.
__global__ void __launch_bounds__(256) vox_keys_batch_kernel(FramePts pts, int n, int c, VoxGeom geo,
                                                             long long cells, FrameOffsets fo, int32_t *__restrict__ pkey,
                                                             uint64_t *bitmap, uint32_t *__restrict__ touched) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int frame = fo.frame_of_lane(i, threadIdx.x & 63);
    const float *p = pts.row(fo, frame, i, c);
    int32_t cz[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float f = floorf(__fdiv_rn(__fsub_rn(p[2 - j], geo.lo[2 - j]), geo.vs[2 - j]));
        if (!(f >= 0.0f) || !(f < (float)geo.g[j])) ok = false;
        cz[j] = ok ? (int32_t)f : 0;
    }
    long long key = -1;
    const int32_t local = ok ? (cz[0] * geo.g[1] + cz[1]) * geo.g[2] + cz[2] : -1;
    if (ok) key = (long long)frame * cells + local;
    // consecutive returns of a beam often share a voxel: the lane after an equal key leaves the bit to its neighbour
    const long long prev = __shfl_up(key, 1);
    if (ok && ((threadIdx.x & 63) == 0 || prev != key)) {
        atomicOr((unsigned long long *)&bitmap[key >> 6], 1ull << (key & 63));
        touched[(key >> 6) / SCAN_TILE] = 1u;                        // (the touched-block scan's mark: any non-zero word, plain store)
    }
    pkey[i] = local;
}

__global__ void __launch_bounds__(256) vox_first_batch_kernel(int n, long long cells, FrameOffsets fo, const int32_t *__restrict__ pkey,
                                                              const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                                                              int32_t *__restrict__ prank, int32_t *first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t local = pkey[i];
    const long long key = local >= 0 ? (long long)fo.frame_of(i) * cells + local : -1;
    int32_t r = -1;
    const long long prev = __shfl_up(key, 1);
    if (key >= 0) {
        const uint64_t w = bitmap[key >> 6];
        r = (int32_t)(base[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1ull)));
        if ((threadIdx.x & 63) == 0 || prev != key) atomicMin(&first[r], i);
    }
    prank[i] = r;
}

struct AssignGlobalFn {  // flagged point i starts the voxel with GLOBAL id = prefix; frame starts record their base
    const int32_t *prank;
    int32_t *gid;
    int32_t *frame_base;
    FrameOffsets fo;
    __device__ void operator()(long long i, uint32_t flag, uint32_t prefix) const {
        // the point that starts a frame records its base; empty frames share an offset with the next non-empty one: all get it
        for (int f = fo.frame_of((int)i); f >= 0 && fo.off[f] == (int32_t)i; --f) frame_base[f] = (int32_t)prefix;
        if (flag) gid[prank[i]] = (int32_t)prefix;
    }
};

__global__ void vox_frames_kernel(FrameOffsets fo, int n, int max_voxels, int32_t *frame_base, const int32_t *total,
                                  int32_t *out_base, int32_t *n_voxels) {
    if (threadIdx.x != 0) return;
    int32_t acc = 0;
    for (int f = 0; f < fo.nf; ++f) {
        if (fo.off[f] >= n) frame_base[f] = *total;               // frames that start past the last point
        const int32_t next = (f + 1 < fo.nf && fo.off[f + 1] < n) ? frame_base[f + 1] : *total;
        int32_t cnt = next - frame_base[f];
        if (cnt > max_voxels) cnt = max_voxels;
        out_base[f] = acc;
        n_voxels[f] = cnt;
        acc += cnt;
    }
    out_base[fo.nf] = acc;
    n_voxels[fo.nf] = acc;
}

__global__ void __launch_bounds__(256) vox_assign_batch_kernel(int n, FrameOffsets fo, int max_voxels,
                                                               const int32_t *__restrict__ pkey,
                                                               const int32_t *__restrict__ prank,
                                                               const int32_t *__restrict__ first,
                                                               const int32_t *__restrict__ frame_base,
                                                               const int32_t *__restrict__ out_base, int32_t *vid,
                                                               int32_t *coords, int32_t gy, int32_t gx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = prank[i];
    if (r < 0 || first[r] != i) return;
    const int f = fo.frame_of(i);
    const int32_t local = vid[r] - frame_base[f];               // vid holds the global id until here
    if (local >= max_voxels) { vid[r] = -1; return; }
    const int32_t row = out_base[f] + local;
    vid[r] = row;
    const int32_t key = pkey[i];                                // the cell inside the frame
    int32_t *o = coords + (size_t)row * 4;
    o[0] = f; o[1] = key / (gx * gy); o[2] = (key / gx) % gy; o[3] = key % gx;
}

// Canonical row order (engine-internal; the boundary's order is first appearance): row = rank of the voxel's cell in the occupancy
// bitmap = ascending (frame, z, y, x). No first-appearance scan over the points, no rank -> row map: the site index is canonical.
// Frame f owns the ranks between the prefix at its first cell and at the next frame's.
__global__ void vox_frames_canonical_kernel(int nf, long long cells, const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                                            const uint32_t *__restrict__ block_sums, const int32_t *__restrict__ total, int32_t *n_voxels) {
    // lane f: the ranks at the first cells of frames f and f + 1 (0 at frame 0, the total at frame nf) -- all frames' lookups in flight
    // together instead of one dependent pair of loads after the other on a single thread (23 -> ~3 us at 48 frames)
    const int f = threadIdx.x;                        // (nf <= CPD_VOX_MAX_FRAMES = 64 = the block)
    auto rank_at = [&](int fr) -> int32_t {
        if (fr <= 0) return 0;
        if (fr >= nf) return *total;
        const long long key = (long long)fr * cells;
        // (the prefix entries of a scan block without a bit are not written by the touched-block scan: the rank anywhere inside such a
        // block is the count at its start)
        if (scan_block_empty(block_sums, key >> 6)) return (int32_t)scan_block_start(block_sums, key >> 6);
        return (int32_t)(base[key >> 6] + __popcll(bitmap[key >> 6] & ((1ull << (key & 63)) - 1ull)));
    };
    if (f < nf) n_voxels[f] = rank_at(f + 1) - rank_at(f);
    if (f == 0) n_voxels[nf] = *total;
}
__global__ void __launch_bounds__(256) vox_assign_canonical_kernel(int n, FrameOffsets fo, const int32_t *__restrict__ pkey,
                                                                   const int32_t *__restrict__ prank, const int32_t *__restrict__ first,
                                                                   int32_t *coords, int32_t gy, int32_t gx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = prank[i];
    if (r < 0 || first[r] != i) return;
    const int32_t key = pkey[i];
    int32_t *o = coords + (size_t)r * 4;
    o[0] = fo.frame_of(i); o[1] = key / (gx * gy); o[2] = (key / gx) % gy; o[3] = key % gx;
}

// Consecutive returns of a beam often fall into one voxel: lanes of a wave with the same voxel form a RUN (in point order).
// The run's last lane adds the run length to the voxel's count in one atomic, and a lane that has max_points earlier lanes of
// its own run can never hold one of the voxel's max_points smallest indices, so it skips the cascade altogether.
__global__ void __launch_bounds__(256) vox_insert_batch_kernel(int n, int P, const int32_t *__restrict__ prank,
                                                               const int32_t *__restrict__ vid, const int32_t *__restrict__ first,
                                                               int32_t *slots, int32_t *counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int32_t v = -1;
    bool is_first = false;
    if (i < n) {
        const int32_t r = prank[i];
        if (r >= 0) {
            v = vid ? vid[r] : r;                                // < 0: voxel beyond its frame's max_voxels; no map: row = rank
            is_first = first[r] == i;
        }
    }
    const int32_t key = v >= 0 ? v : -1 - lane;                  // invalid lanes never join a run
    int pos = 0;                                                 // lanes before this one in its run
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t k2 = __shfl_up(key, d, 64);
        const int p2 = __shfl_up(pos, d, 64);
        if (lane >= d && k2 == key && pos == d - 1) pos += p2 + 1;   // the run reaches back at least d lanes: append the earlier part
    }
    const int32_t kn = __shfl_down(key, 1, 64);
    if (v < 0) return;
    if (lane == 63 || kn != key) atomicAdd(&counts[v], pos + 1);
    if (pos >= P) return;
    int32_t x = i;
    int32_t *s = slots + (size_t)v * P;
    // slot 0 is already known: the voxel's first point (vox_first) owns it with a plain store; everybody else starts the
    // cascade at slot 1 -- one returning atomic less per point, none at all for the single-point voxels
    if (is_first) { s[0] = i; return; }
    for (int p = 1; p < P; ++p) {
        int32_t old = atomicMin(&s[p], x);
        if (old == 0x7f7f7f7f) break;
        x = old > x ? old : x;
    }
}

// ---- canonical rows, round 3: per-voxel point lists by counting sort instead of the atomicMin cascade ----
// The cascade (vox_insert_batch_kernel) costs a returning atomic per point and slot on a [voxels][max_points] array that is
// 150 MB at 48 frames: every atomic is an HBM line in and out. Here a point takes an ARRIVAL slot in its voxel with one
// atomicAdd on a dense counter array (one per run of equal voxels in a wave), a scan of the counters gives every voxel its
// segment of one compact point list, and the kernel that builds the voxel picks the max_points SMALLEST indices of its segment in
// ascending order -- the points the serial voxelizer keeps, in its summation order -- whatever order they arrived in. No `first`
// array, no slot fill; the voxel's coordinates come from its first kept point.
__global__ void __launch_bounds__(256) vox_count_kernel(int n, long long cells, FrameOffsets fo, const int32_t *__restrict__ pkey,
                                                        const uint64_t *__restrict__ bitmap, const uint32_t *__restrict__ base,
                                                        int32_t *__restrict__ prank, int32_t *__restrict__ ppos, int32_t *counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int32_t r = -1;
    if (i < n) {
        const int32_t local = pkey[i];
        if (local >= 0) {
            const long long key = (long long)fo.frame_of_lane(i, lane) * cells + local;
            const uint64_t w = bitmap[key >> 6];
            r = (int32_t)(base[key >> 6] + __popcll(w & ((1ull << (key & 63)) - 1ull)));
        }
    }
    // runs of equal voxels among consecutive lanes (consecutive returns of a beam): one atomic per run
    const int32_t rr = r >= 0 ? r : -1 - lane;                   // invalid lanes never join a run
    const int32_t prev = __shfl_up(rr, 1, 64);
    const unsigned long long starts = __ballot(lane == 0 || prev != rr);
    const int my_start = 63 - __builtin_clzll(starts & (~0ull >> (63 - lane)));       // lane 0 always starts a run
    const unsigned long long above = lane == 63 ? 0ull : (starts >> (lane + 1)) << (lane + 1);
    const int next = above ? __builtin_ctzll(above) : 64;
    int32_t slot0 = 0;
    if (r >= 0 && lane == my_start) slot0 = atomicAdd(&counts[r], next - my_start);
    slot0 = __shfl(slot0, my_start, 64);
    if (i < n) {
        prank[i] = r;
        ppos[i] = slot0 + (lane - my_start);
    }
}

struct CountFn {
    const int32_t *counts;
    __device__ uint32_t operator()(long long v) const { return (uint32_t)counts[v]; }
};
struct StoreOffsetFn {
    int32_t *offsets;
    __device__ void operator()(long long v, uint32_t, uint32_t prefix) const { offsets[v] = (int32_t)prefix; }
};

__global__ void __launch_bounds__(256) vox_scatter_kernel(int n, const int32_t *__restrict__ prank, const int32_t *__restrict__ ppos,
                                                          const int32_t *__restrict__ offsets, int32_t *__restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = prank[i];
    if (r >= 0) order[offsets[r] + ppos[i]] = i;
}

// thread (voxel v, channel ch): the voxel's kept points = the max_points smallest indices of its segment, ascending
__global__ void __launch_bounds__(256) vox_build_kernel(FramePts pts, int c, int P, int cap, FrameOffsets fo,
                                                        const int32_t *__restrict__ n_vox, const int32_t *__restrict__ pkey,
                                                        const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets,
                                                        const int32_t *__restrict__ order, float *voxels, int32_t *coords,
                                                        int32_t *num_points, float *mean, int32_t gy, int32_t gx) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int v = (int)(tid / c), ch = (int)(tid % c);
    if (v >= cap || v >= *n_vox) return;
    const int cnt = counts[v];
    const int32_t *seg = order + offsets[v];
    const int kept = cnt < P ? cnt : P;
    if (cnt <= 8 && P <= 8) {
        // the common case in registers: the segment's (<= 8) entries in one round of independent loads, the selection on the
        // vector ALU, then the kept points' values in a second round -- two memory round trips instead of one per candidate
        int32_t e[8], sel[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = j < cnt ? seg[j] : 0x7fffffff;
        int32_t last = -1;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            sel[p] = 0x7fffffff;
            if (p < kept) {                 // (skipped by the whole wave once none of its voxels keeps that many: most keep 1-3)
                int32_t best = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 8; ++j) best = (e[j] > last && e[j] < best) ? e[j] : best;
                sel[p] = best;
                last = best;
            }
        }
        const int frame = fo.frame_of_near(kept > 0 ? sel[0] : -1);      // (a voxel's points all belong to its frame)
        float val[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) val[p] = p < kept ? pts.row(fo, frame, sel[p], c)[ch] : 0.f;
        if (ch == 0 && kept > 0) {
            const int32_t key = pkey[sel[0]];                      // the cell inside the frame
            int32_t *o = coords + (size_t)v * 4;
            o[0] = frame; o[1] = key / (gx * gy); o[2] = (key / gx) % gy; o[3] = key % gx;
        }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) {       // sum order p = 0..P-1, zeros included (mean_vfe.py:41)
            if (p < P) {
                if (voxels) voxels[((size_t)v * P + p) * c + ch] = val[p];
                s += val[p];
            }
        }
        if (mean) mean[(size_t)v * c + ch] = __fdiv_rn(s, (float)(kept < 1 ? 1 : kept));
        if (ch == 0) num_points[v] = kept;
        return;
    }
    int32_t last = -1;
    int frame = 0;
    float s = 0.f;
    for (int p = 0; p < P; ++p) {      // sum order p = 0..P-1, zeros included (mean_vfe.py:41)
        float val = 0.f;
        if (p < kept) {
            int32_t best = 0x7fffffff;
            for (int j = 0; j < cnt; ++j) {
                const int32_t x = seg[j];
                best = (x > last && x < best) ? x : best;
            }
            last = best;
            if (p == 0) frame = fo.frame_of(best);
            val = pts.row(fo, frame, best, c)[ch];
            if (p == 0 && ch == 0) {
                const int32_t key = pkey[best];                    // the cell inside the frame
                int32_t *o = coords + (size_t)v * 4;
                o[0] = frame; o[1] = key / (gx * gy); o[2] = (key / gx) % gy; o[3] = key % gx;
            }
        }
        if (voxels) voxels[((size_t)v * P + p) * c + ch] = val;
        s += val;
    }
    if (mean) mean[(size_t)v * c + ch] = __fdiv_rn(s, (float)(kept < 1 ? 1 : kept));
    if (ch == 0) num_points[v] = kept;
}

}  // namespace

static int vox_geom(const float vs[3], const float rg[6], VoxGeom *g, long long *cells) {
    int32_t grid[3];
    int rc = cpd_voxel_grid_size(vs, rg, grid);
    if (rc) return rc;
    for (int a = 0; a < 3; ++a) { g->lo[a] = rg[a]; g->vs[a] = vs[a]; g->g[a] = grid[a]; }
    *cells = (long long)grid[0] * grid[1] * grid[2];
    if (*cells >= (1ll << 31)) return CPD_ERR_UNSUPPORTED;
    return CPD_OK;
}

extern "C" size_t cpd_voxelize_workspace_bytes(int n_points, int max_points, int max_voxels, const float vsize_xyz[3],
                                               const float range_xyz[6]) {
    VoxGeom g;
    long long cells;
    if (n_points < 0 || max_points <= 0 || max_voxels <= 0) return 0;
    if (vox_geom(vsize_xyz, range_xyz, &g, &cells)) return 0;
    int cap = max_voxels < n_points ? max_voxels : n_points;
    return carve(nullptr, n_points, max_points, cap, cells).bytes;
}

extern "C" int cpd_voxelize(const float *points, int n_points, int c, const float vsize_xyz[3],
                            const float range_xyz[6], int max_points, int max_voxels, int batch_idx, int coord_cols,
                            float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                            int32_t *n_voxels, void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    if (n_points < 0 || c < 3 || max_points <= 0 || max_voxels <= 0 || !coords || !num_points || !n_voxels ||
        !workspace || (coord_cols != 3 && coord_cols != 4) || (n_points > 0 && !points))
        return CPD_ERR_ARG;
    VoxGeom geo;
    long long cells;
    int rc = vox_geom(vsize_xyz, range_xyz, &geo, &cells);
    if (rc) return rc;
    hipStream_t s = cpd_s(stream);
    const int n = n_points;
    const int cap = max_voxels < n ? max_voxels : n;
    VoxWs w = carve(workspace, n, max_points, cap, cells);
    if (workspace_bytes < w.bytes) return CPD_ERR_WORKSPACE;
    if (n == 0) {
        CPD_HIP_TRY(hipMemsetAsync(n_voxels, 0, 4, s));
        return CPD_OK;
    }
    CPD_HIP_TRY(hipMemsetAsync(w.bitmap, 0, (size_t)w.words * 8, s));
    CPD_HIP_TRY(hipMemsetAsync(w.first, 0x7f, (size_t)n * 4, s));
    CPD_HIP_TRY(hipMemsetAsync(w.slots, 0x7f, (size_t)cap * max_points * 4, s));
    CPD_HIP_TRY(hipMemsetAsync(w.counts, 0, (size_t)cap * 4, s));
    const int nb = cpd_div_up(n, 256);
    vox_keys_kernel<<<nb, 256, 0, s>>>(points, n, c, geo, w.pkey, w.bitmap);
    rc = device_scan(w.words, PopcFn{w.bitmap}, StoreBaseFn{w.base}, w.bsum_bm, w.nocc, -1, s);
    if (rc) return rc;
    vox_first_kernel<<<nb, 256, 0, s>>>(n, w.pkey, w.bitmap, w.base, w.prank, w.first);
    rc = device_scan(n, FlagFn{w.prank, w.first},
                     AssignVoxelFn{w.prank, w.pkey, w.vid, coords, coord_cols, batch_idx, max_voxels, geo.g[1], geo.g[2]},
                     w.bsum_pt, n_voxels, max_voxels, s);
    if (rc) return rc;
    vox_insert_kernel<<<nb, 256, 0, s>>>(n, max_points, max_voxels, w.prank, w.vid, w.first, w.slots, w.counts);
    const long long threads = (long long)cap * c;
    vox_gather_kernel<<<cpd_div_up(threads, 256), 256, 0, s>>>(points, c, max_points, cap, n_voxels, w.slots, w.counts,
                                                               voxels, num_points, mean_features);
    return cpd_check_launch();
}

// ---- batched entry points ---------------------------------------------------------------------
static int batch_caps(int n_total, int n_frames, int max_voxels, long long cells, int *cap) {
    if (n_frames <= 0 || n_frames > CPD_VOX_MAX_FRAMES) return CPD_ERR_UNSUPPORTED;
    if (cells >= (1ll << 31) || (long long)n_frames * cells >= (1ll << 40)) return CPD_ERR_UNSUPPORTED;
    long long c = (long long)n_frames * max_voxels;
    *cap = (int)(c < n_total ? c : n_total);
    return CPD_OK;
}

extern "C" size_t cpd_voxelize_batch_workspace_bytes(int n_total, int n_frames, int max_points, int max_voxels,
                                                     const float vsize_xyz[3], const float range_xyz[6]) {
    VoxGeom g;
    long long cells;
    int cap;
    if (n_total < 0 || max_points <= 0 || max_voxels <= 0) return 0;
    if (vox_geom(vsize_xyz, range_xyz, &g, &cells)) return 0;
    if (batch_caps(n_total, n_frames, max_voxels, cells, &cap)) return 0;
    return carve(nullptr, n_total, max_points, cap, (long long)n_frames * cells).bytes + cpd_align(2 * (CPD_VOX_MAX_FRAMES + 1) * 4 + 16);
}

// `index` != NULL: the occupancy bitmap, its popcount prefix and the rank -> row map are built INSIDE that site index (of
// the grid with z_extra more z-levels -- the backbone's sparse_shape = grid + [1,0,0], spconv_backbone.py:412), which is
// then the level-0 index of the sparse tensor as it stands: no second bitmap, mark, scan and permutation pass.
static int voxelize_batch_impl(const float *points, const float *const *frame_points, const int32_t *frame_offsets, int n_frames, int c,
                               const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                               float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                               int32_t *n_voxels, void *workspace, size_t workspace_bytes, void *index, size_t index_bytes,
                               int z_extra, int canonical, cpd_stream_t stream) {
    if (!frame_offsets || c < 3 || max_points <= 0 || max_voxels <= 0 || !coords || !num_points || !n_voxels || !workspace ||
        z_extra < 0)
        return CPD_ERR_ARG;
    VoxGeom geo;
    long long cells;
    int rc = vox_geom(vsize_xyz, range_xyz, &geo, &cells);
    if (rc) return rc;
    if (index) cells = (long long)(geo.g[0] + z_extra) * geo.g[1] * geo.g[2];   // keys run over the index's (deeper) grid
    if (n_frames <= 0 || n_frames > CPD_VOX_MAX_FRAMES) return CPD_ERR_UNSUPPORTED;
    FrameOffsets fo;
    fo.nf = n_frames;
    for (int f = 0; f <= n_frames; ++f) {
        fo.off[f] = frame_offsets[f];
        if (frame_offsets[f] < 0 || (f > 0 && frame_offsets[f] < frame_offsets[f - 1])) return CPD_ERR_ARG;
    }
    if (frame_offsets[0] != 0) return CPD_ERR_ARG;
    const int n = frame_offsets[n_frames];
    if (n > 0 && !points && !frame_points) return CPD_ERR_ARG;
    FramePts fp;
    for (int f = 0; f < CPD_VOX_MAX_FRAMES; ++f) fp.p[f] = nullptr;
    for (int f = 0; f < n_frames; ++f) {
        fp.p[f] = frame_points ? frame_points[f] : points + (size_t)frame_offsets[f] * c;
        if (!fp.p[f] && frame_offsets[f + 1] > frame_offsets[f]) return CPD_ERR_ARG;
    }
    int cap;
    rc = batch_caps(n, n_frames, max_voxels, cells, &cap);
    if (rc) return rc;
    if (canonical && cap < n) return CPD_ERR_UNSUPPORTED;     // rows are ranks (< n): the row capacity must not be cap-limited
    hipStream_t s = cpd_s(stream);
    // with an index the bitmap / prefix / scan spine live there: the workspace (sized by cpd_voxelize_batch_workspace_bytes for
    // the plain grid) then carries no bitmap at all
    VoxWs w = carve(workspace, n, max_points, cap, index ? 0 : (long long)n_frames * cells);
    if (workspace_bytes < w.bytes + cpd_align(2 * (CPD_VOX_MAX_FRAMES + 1) * 4 + 16)) return CPD_ERR_WORKSPACE;
    int32_t *const vid_ws = w.vid;          // the workspace's own [n] words (below, w.vid may become the index's rank -> row map)
    if (index) {
        const int32_t shape[3] = {geo.g[0] + z_extra, geo.g[1], geo.g[2]};
        IndexView v = index_carve(index, n_frames, shape, n > 0 ? n : 1);       // ranks <= occupied cells <= points
        if (index_bytes < v.bytes) return CPD_ERR_WORKSPACE;
        w.words = v.words;
        w.bitmap = v.bitmap; w.base = v.base; w.bsum_bm = v.bsum; w.vid = v.perm;
        CPD_HIP_TRY(hipMemsetAsync(v.flags, 0, 4, s));
        if (!canonical) CPD_HIP_TRY(hipMemsetAsync(v.flags, 1, 1, s));          // flags[0] = 1: row id = perm[rank]; canonical: rank
    }
    int32_t *frame_base = (int32_t *)((char *)workspace + w.bytes);
    int32_t *out_base = frame_base + CPD_VOX_MAX_FRAMES + 1;
    int32_t *total = out_base + CPD_VOX_MAX_FRAMES + 1;
    if (n == 0) {
        CPD_HIP_TRY(hipMemsetAsync(n_voxels, 0, (size_t)(n_frames + 1) * 4, s));
        if (index) {                                          // an empty but valid index
            CPD_HIP_TRY(hipMemsetAsync(w.bitmap, 0, (size_t)w.words * 8, s));
            CPD_HIP_TRY(hipMemsetAsync(w.base, 0, (size_t)w.words * 4, s));
        }
        return CPD_OK;
    }
    if (cpd_zero_fill(w.bitmap, (size_t)w.words * 8, s)) return CPD_ERR_LAUNCH;
    if (cpd_zero_fill(w.counts, (size_t)cap * 4, s)) return CPD_ERR_LAUNCH;
    // the bitmap's scan visits only the 32 KB blocks that hold a bit (device_scan_touched): its spine starts out zero and the key
    // kernel marks the blocks it sets bits in
    CPD_HIP_TRY(hipMemsetAsync(w.bsum_bm, 0, (size_t)scan_num_blocks(w.words) * 4, s));
    const int nb = cpd_div_up(n, 256);
    bool cascade = false;                    // tuning only (CPD_TUNE=1 CPD_VOX_CASCADE=1): round 2's atomicMin cascade, for A/B timing
    if (const char *e = cpd_knob(cpd_tuning(), "CPD_VOX_CASCADE")) cascade = atoi(e) != 0;
    if (canonical && !cascade) {
        // rows = ranks: the max_voxels cap (defined on first-appearance order) is NOT applied -- the caller checks the per-frame
        // counts and falls back to the exact path if a frame exceeds it. Point lists by counting sort (above): ppos in w.first's
        // words, the compact list in w.slots' (cap * max_points >= n of them), the segment offsets in the workspace's vid words.
        int32_t *const ppos = w.first, *const order = w.slots, *const offsets = vid_ws;
        vox_keys_batch_kernel<<<nb, 256, 0, s>>>(fp, n, c, geo, cells, fo, w.pkey, w.bitmap, w.bsum_bm);
        rc = device_scan_touched(w.words, PopcFn{w.bitmap}, StoreBaseFn{w.base}, w.bsum_bm, w.nocc, -1, s);
        if (rc) return rc;
        vox_count_kernel<<<nb, 256, 0, s>>>(n, cells, fo, w.pkey, w.bitmap, w.base, w.prank, ppos, w.counts);
        rc = device_scan(n, CountFn{w.counts}, StoreOffsetFn{offsets}, w.bsum_pt, nullptr, -1, s);   // (ranks < occupied cells <= n <= cap)
        if (rc) return rc;
        vox_frames_canonical_kernel<<<1, 64, 0, s>>>(n_frames, cells, w.bitmap, w.base, w.bsum_bm, w.nocc, n_voxels);
        vox_scatter_kernel<<<nb, 256, 0, s>>>(n, w.prank, ppos, offsets, order);
        const long long threads_c = (long long)cap * c;
        vox_build_kernel<<<cpd_div_up(threads_c, 256), 256, 0, s>>>(fp, c, max_points, cap, fo, n_voxels + n_frames, w.pkey, w.counts, offsets,
                                                                     order, voxels, coords, num_points, mean_features, geo.g[1], geo.g[2]);
        return cpd_check_launch();
    }
    if (cpd_fill_bytes(w.first, 0x7f, (size_t)n * 4, s)) return CPD_ERR_LAUNCH;
    if (cpd_fill_bytes(w.slots, 0x7f, (size_t)cap * max_points * 4, s)) return CPD_ERR_LAUNCH;
    CPD_HIP_TRY(hipMemsetAsync(frame_base, 0, (CPD_VOX_MAX_FRAMES + 1) * 4, s));
    vox_keys_batch_kernel<<<nb, 256, 0, s>>>(fp, n, c, geo, cells, fo, w.pkey, w.bitmap, w.bsum_bm);
    rc = device_scan_touched(w.words, PopcFn{w.bitmap}, StoreBaseFn{w.base}, w.bsum_bm, w.nocc, -1, s);
    if (rc) return rc;
    vox_first_batch_kernel<<<nb, 256, 0, s>>>(n, cells, fo, w.pkey, w.bitmap, w.base, w.prank, w.first);
    if (canonical) {
        // rows = ranks: the max_voxels cap (defined on first-appearance order) is NOT applied -- the caller checks the per-frame
        // counts and falls back to the exact path if a frame exceeds it
        vox_frames_canonical_kernel<<<1, 64, 0, s>>>(n_frames, cells, w.bitmap, w.base, w.bsum_bm, w.nocc, n_voxels);
        vox_assign_canonical_kernel<<<nb, 256, 0, s>>>(n, fo, w.pkey, w.prank, w.first, coords, geo.g[1], geo.g[2]);
        vox_insert_batch_kernel<<<nb, 256, 0, s>>>(n, max_points, w.prank, nullptr, w.first, w.slots, w.counts);
    } else {
        rc = device_scan(n, FlagFn{w.prank, w.first}, AssignGlobalFn{w.prank, w.vid, frame_base, fo}, w.bsum_pt, total, -1, s);
        if (rc) return rc;
        vox_frames_kernel<<<1, 64, 0, s>>>(fo, n, max_voxels, frame_base, total, out_base, n_voxels);
        vox_assign_batch_kernel<<<nb, 256, 0, s>>>(n, fo, max_voxels, w.pkey, w.prank, w.first, frame_base, out_base,
                                                   w.vid, coords, geo.g[1], geo.g[2]);
        vox_insert_batch_kernel<<<nb, 256, 0, s>>>(n, max_points, w.prank, w.vid, w.first, w.slots, w.counts);
    }
    const long long threads = (long long)cap * c;
    vox_gather_frames_kernel<<<cpd_div_up(threads, 256), 256, 0, s>>>(fp, fo, c, max_points, cap, n_voxels + n_frames, w.slots, w.counts,
                                                                      voxels, num_points, mean_features);
    return cpd_check_launch();
}

extern "C" int cpd_voxelize_batch(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                                  const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                                  float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                                  int32_t *n_voxels, void *workspace, size_t workspace_bytes, cpd_stream_t stream) {
    return voxelize_batch_impl(points, nullptr, frame_offsets, n_frames, c, vsize_xyz, range_xyz, max_points, max_voxels, voxels, coords,
                               num_points, mean_features, n_voxels, workspace, workspace_bytes, nullptr, 0, 0, 0, stream);
}

extern "C" int cpd_voxelize_batch_index(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                                        const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                                        float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                                        int32_t *n_voxels, void *workspace, size_t workspace_bytes, void *index,
                                        size_t index_bytes, int z_extra, cpd_stream_t stream) {
    if (!index) return CPD_ERR_ARG;
    return voxelize_batch_impl(points, nullptr, frame_offsets, n_frames, c, vsize_xyz, range_xyz, max_points, max_voxels, voxels, coords,
                               num_points, mean_features, n_voxels, workspace, workspace_bytes, index, index_bytes, z_extra, 0, stream);
}

extern "C" int cpd_voxelize_batch_canonical(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                                            const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                                            float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                                            int32_t *n_voxels, void *workspace, size_t workspace_bytes, void *index,
                                            size_t index_bytes, int z_extra, cpd_stream_t stream) {
    if (!index) return CPD_ERR_ARG;
    return voxelize_batch_impl(points, nullptr, frame_offsets, n_frames, c, vsize_xyz, range_xyz, max_points, max_voxels, voxels, coords,
                               num_points, mean_features, n_voxels, workspace, workspace_bytes, index, index_bytes, z_extra, 1, stream);
}

// The frames' point lists as they lie -- one device pointer per frame (a HOST array of n_frames pointers; frame f holds
// frame_offsets[f + 1] - frame_offsets[f] rows of c floats) -- instead of one concatenated buffer: everything else is
// cpd_voxelize_batch (index = NULL), cpd_voxelize_batch_index (canonical = 0) or cpd_voxelize_batch_canonical (canonical = 1).
extern "C" int cpd_voxelize_batch_frames(const float *const *frame_points, const int32_t *frame_offsets, int n_frames, int c,
                                         const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                                         float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                                         int32_t *n_voxels, void *workspace, size_t workspace_bytes, void *index,
                                         size_t index_bytes, int z_extra, int canonical, cpd_stream_t stream) {
    if (!frame_points || (canonical && !index)) return CPD_ERR_ARG;
    return voxelize_batch_impl(nullptr, frame_points, frame_offsets, n_frames, c, vsize_xyz, range_xyz, max_points, max_voxels, voxels, coords,
                               num_points, mean_features, n_voxels, workspace, workspace_bytes, index, index ? index_bytes : 0,
                               index ? z_extra : 0, canonical, stream);
}
