// row_plan.hip -- the ROW PLAN of a sub-manifold rulebook (round 4): what the staged row-wave kernel (gather_conv.hip,
// rowplan_conv_f16p_kernel) needs to fetch every input row of a tile ONCE instead of once per (row, tap) pair.
//
// [SPCONV] spconv's indice pairs (spconv_backbone.py:17-21 SubMConv3d, indice_key 'subm*' / 'res*') list, per kernel offset, the
// (input row, output row) pairs; cpd_rulebook_subm keeps them output-stationary as nbr[27][n]. A SubMConv3d's 27 taps fall into three
// dz groups of nine (tap = (dz * 3 + dy) * 3 + dx): the groups read three different z-planes, so they share no input row. For every
// tile of 128 consecutive output rows and every group the plan holds
//   ulist [tile][group][0 .. count)   the DISTINCT input rows the group's nine taps touch, ascending;
//   slots [tile][tap][row]            u16: position of nbr[tap][row] in its group's list (0xffff: no neighbour);
//   count [tile][0..2]                list lengths, [3] their sum.
// Rows that are neighbours in space share most of their inputs: 2.9 distinct rows per output row in brick order
// (cpd_order_rows_bricks) against 13.5 pairs (tools/unique_probe2.py). The plan is built once per level and read by its four convs.
//
// One workgroup per tile: the group's <= 1152 ids are de-duplicated in an LDS hash table (open addressing; atomicCAS), ranked by
// counting (rank = number of smaller ids: all pairs, ~200^2 / 256 threads), and every (tap, row) finds its slot by binary search.
// No ordering assumption on the ids: any row order is planned correctly, only the list lengths depend on it.
#include "common.h"

#define CPD_PLAN_TILE 128
#define CPD_PLAN_LIST 1152

namespace {

__global__ void __launch_bounds__(256) rulebook_plan_kernel(const int32_t *__restrict__ nbr, int n, uint16_t *__restrict__ slots,
                                                            int32_t *__restrict__ ulist, int32_t *__restrict__ count) {
    __shared__ int32_t htab[2048];
    __shared__ int32_t list[CPD_PLAN_LIST];
    __shared__ int32_t sorted[CPD_PLAN_LIST];
    __shared__ int cnt;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int row0 = tile * CPD_PLAN_TILE;
    int total = 0;
    for (int grp = 0; grp < 3; ++grp) {
        for (int i = tid; i < 2048; i += 256) htab[i] = -1;
        if (tid == 0) cnt = 0;
        __syncthreads();
        int32_t my[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int e = tid + 256 * k;
            int32_t id = -1;
            if (e < 9 * CPD_PLAN_TILE) {
                const int t = 9 * grp + (e >> 7), r = e & 127;
                if (row0 + r < n) id = nbr[(size_t)t * n + row0 + r];
            }
            my[k] = id;
            if (id >= 0) {
                uint32_t h = ((uint32_t)id * 2654435761u) >> 21;           // 11 bits
                while (true) {
                    const int32_t old = atomicCAS(&htab[h], -1, id);
                    if (old == -1 || old == id) break;
                    h = (h + 1) & 2047u;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 2048; i += 256) {
            const int32_t v = htab[i];
            if (v >= 0) list[atomicAdd(&cnt, 1)] = v;
        }
        __syncthreads();
        const int u = cnt;
        for (int i = tid; i < u; i += 256) {
            const int32_t v = list[i];
            int rnk = 0;
            for (int j = 0; j < u; ++j) rnk += list[j] < v ? 1 : 0;
            sorted[rnk] = v;
        }
        __syncthreads();
        int32_t *ul = ulist + ((size_t)tile * 3 + grp) * CPD_PLAN_LIST;
        for (int i = tid; i < u; i += 256) ul[i] = sorted[i];
        if (tid == 0) count[tile * 4 + grp] = u;
        total += u;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int e = tid + 256 * k;
            if (e >= 9 * CPD_PLAN_TILE) continue;
            const int t = 9 * grp + (e >> 7), r = e & 127;
            uint32_t sl = 0xffffu;
            const int32_t id = my[k];
            if (id >= 0) {
                int lo = 0, hi = u - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sorted[mid] < id) lo = mid + 1; else hi = mid;
                }
                sl = (uint32_t)lo;
            }
            slots[((size_t)tile * 27 + t) * CPD_PLAN_TILE + r] = (uint16_t)sl;
        }
        __syncthreads();
    }
    if (tid == 0) count[tile * 4 + 3] = total;
}

}  // namespace

extern "C" size_t cpd_rulebook_plan_bytes(int n_out, int which) {
    if (n_out < 0) return 0;
    const size_t tiles = (size_t)cpd_div_up(n_out > 0 ? n_out : 1, CPD_PLAN_TILE);
    if (which == 0) return tiles * 27 * CPD_PLAN_TILE * sizeof(uint16_t);   // slots
    if (which == 1) return tiles * 3 * CPD_PLAN_LIST * sizeof(int32_t);     // ulist
    if (which == 2) return tiles * 4 * sizeof(int32_t);                     // count
    return 0;
}

extern "C" int cpd_rulebook_plan(const int32_t *nbr, int kv, int n_out, uint16_t *slots, int32_t *ulist, int32_t *count,
                                 cpd_stream_t stream) {
    if (n_out < 0 || (n_out > 0 && (!nbr || !slots || !ulist || !count))) return CPD_ERR_ARG;
    if (kv != 27) return CPD_ERR_UNSUPPORTED;
    if (n_out == 0) return CPD_OK;
    rulebook_plan_kernel<<<cpd_div_up(n_out, CPD_PLAN_TILE), 256, 0, cpd_s(stream)>>>(nbr, n_out, slots, ulist, count);
    return cpd_check_launch();
}
