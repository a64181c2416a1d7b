// row_plan.hip -- the ROW PLAN of a sub-manifold rulebook (round 4): what the staged row-wave kernel (gather_conv.hip,
// rowplan_conv_f16p_kernel) needs to fetch every input row of a tile ONCE instead of once per (row, tap) pair.
//
// [SPCONV] spconv's indice pairs (spconv_backbone.py:17-21 SubMConv3d, indice_key 'subm*' / 'res*') list, per kernel offset, the
// (input row, output row) pairs; cpd_rulebook_subm keeps them output-stationary as nbr[27][n]. A SubMConv3d's 27 taps fall into three
// dz groups of nine (tap = (dz * 3 + dy) * 3 + dx): the groups read three different z-planes, so they share no input row. For every
// tile of 128 consecutive output rows and every group the plan holds
//   ulist [tile][group][0 .. count)   the DISTINCT input rows the group's nine taps touch (in order of first occurrence);
//   slots [tile][tap][row]            u16: position of nbr[tap][row] in its group's list (0xffff: no neighbour);
//   count [tile][0..2]                list lengths, [3] their sum.
// Rows that are neighbours in space share most of their inputs: 2.9 distinct rows per output row in brick order
// (cpd_order_rows_bricks) against 13.5 pairs (tools/unique_probe2.py). The plan is built once per level and read by its four convs.
//
// One workgroup per tile: the group's <= 9 * tile ids are de-duplicated in an LDS hash table (open addressing; atomicCAS); the FIRST
// entry (tap-major, row-minor) of an id is found with an atomicMin, and a block scan over the "first entry" flags gives the id its list
// position -- the order of first occurrence: deterministic, no sort; every other (tap, row) with that id reads it back from the table. (Round 4's first version ranked the ids by counting and searched them: 475 us per 1.65 M rows at 256-row tiles; this
// one is bounded by reading the table it plans.) No ordering assumption on the ids: any row order is planned correctly.
#include "common.h"


namespace {

template <int TILE>
__global__ void __launch_bounds__(256) rulebook_plan_kernel(const int32_t *__restrict__ nbr, int n, uint16_t *__restrict__ slots,
                                                            int32_t *__restrict__ ulist, int32_t *__restrict__ count) {
    constexpr int LIST = 9 * TILE, HT = 16 * TILE, PER = (9 * TILE + 255) / 256;
    __shared__ int32_t hkey[HT];               // open-addressing table: row id (-1 = free)
    __shared__ int32_t hmin[HT];               // ... the smallest entry number (tap-major, row-minor) that carries the id
    __shared__ uint16_t hslot[HT];             // ... and the list position of the id
    __shared__ uint32_t scan_tmp[17];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int row0 = tile * TILE;
    int total = 0;
    for (int grp = 0; grp < 3; ++grp) {
        for (int i = tid; i < HT; i += 256) { hkey[i] = -1; hmin[i] = 0x7fffffff; }
        __syncthreads();
        int32_t *ul = ulist + ((size_t)tile * 3 + grp) * LIST;
        // a thread owns PER CONSECUTIVE entries e = tid * PER + k of the group (entry e = tap 9 grp + e / TILE, row e % TILE)
        uint32_t where[PER];                   // table position of the entry's id (HT: no neighbour)
        int32_t ids[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            int32_t id = -1;
            if (e < 9 * TILE) {
                const int t = 9 * grp + e / TILE, r = e % TILE;
                if (row0 + r < n) id = nbr[(size_t)t * n + row0 + r];
            }
            ids[k] = id;
            where[k] = HT;
            if (id >= 0) {
                uint32_t h = (((uint32_t)id * 2654435761u) >> 16) & (uint32_t)(HT - 1);
                while (true) {
                    const int32_t old = atomicCAS(&hkey[h], -1, id);
                    if (old == -1 || old == id) break;
                    h = (h + 1) & (uint32_t)(HT - 1);
                }
                where[k] = h;
                atomicMin(&hmin[h], e);
            }
        }
        __syncthreads();
        // list position of an id = how many ids have their first entry before its first entry: a block scan over the "I am my id's first
        // entry" flags in entry order -- deterministic (arrival order is not: a group longer than the kernel's window is walked in passes,
        // and which rows fall into which pass decides the order a row's taps are accumulated in; bench.py's results_digest caught it)
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) mine += (where[k] < (uint32_t)HT && hmin[where[k]] == tid * PER + k) ? 1u : 0u;
        uint32_t u;
        uint32_t pos = block_excl_scan(mine, scan_tmp, &u);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (where[k] < (uint32_t)HT && hmin[where[k]] == tid * PER + k) {
                hslot[where[k]] = (uint16_t)pos;
                ul[pos] = ids[k];
                ++pos;
            }
        }
        __syncthreads();
        if (tid == 0) count[tile * 4 + grp] = (int32_t)u;
        total += (int)u;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid * PER + k;
            if (e >= 9 * TILE) continue;
            const int t = 9 * grp + e / TILE, r = e % TILE;
            slots[((size_t)tile * 27 + t) * TILE + r] = where[k] < (uint32_t)HT ? hslot[where[k]] : (uint16_t)0xffffu;
        }
        __syncthreads();
    }
    if (tid == 0) count[tile * 4 + 3] = total;
}

}  // namespace

extern "C" size_t cpd_rulebook_plan_bytes(int n_out, int tile_rows, int which) {
    if (n_out < 0 || (tile_rows != 128 && tile_rows != 256)) return 0;
    const size_t tiles = (size_t)cpd_div_up(n_out > 0 ? n_out : 1, tile_rows);
    if (which == 0) return tiles * 27 * tile_rows * sizeof(uint16_t);       // slots
    if (which == 1) return tiles * 3 * 9 * tile_rows * sizeof(int32_t);     // ulist
    if (which == 2) return tiles * 4 * sizeof(int32_t);                     // count
    return 0;
}

extern "C" int cpd_rulebook_plan(const int32_t *nbr, int kv, int n_out, int tile_rows, uint16_t *slots, int32_t *ulist, int32_t *count,
                                 cpd_stream_t stream) {
    if (n_out < 0 || (n_out > 0 && (!nbr || !slots || !ulist || !count))) return CPD_ERR_ARG;
    if (kv != 27 || (tile_rows != 128 && tile_rows != 256)) return CPD_ERR_UNSUPPORTED;
    if (n_out == 0) return CPD_OK;
    if (tile_rows == 128) rulebook_plan_kernel<128><<<cpd_div_up(n_out, 128), 256, 0, cpd_s(stream)>>>(nbr, n_out, slots, ulist, count);
    else rulebook_plan_kernel<256><<<cpd_div_up(n_out, 256), 256, 0, cpd_s(stream)>>>(nbr, n_out, slots, ulist, count);
    return cpd_check_launch();
}
