// What does the vector-memory path (TA -> TCP -> L2) charge for the SHAPE of a gather? The sparse row-wave kernels fetch, per
// (16-row sub-tile, 32-channel block), 16 rows x 128 B with two 64-lane dwordx4 loads; rocprofv3 (profiles/r03_rowwave_pmc.json)
// counts ~64 TCP accesses per such instruction and puts the TCP at 0.64-0.84 accesses per cycle per CU. This probe issues the SAME
// bytes (16 rows x 128 B per pair of instructions, rows drawn from an L2-resident table) in different lane -> address shapes:
//   frag   lane l -> row l & 15, 16-byte piece 2*(l >> 4) + i   (the MFMA fragment shape the kernels use: a quad = 4 rows)
//   quad   lane l -> row l >> 2, piece (l & 3) + 4*i            (a quad = 64 contiguous bytes of one row)
//   line   lane l -> row (l >> 3) + 8*i, piece l & 7            (8 lanes = one 128-byte row)
//   quad + transpose: `quad` loads followed by the 4 x 16 lane transpose (8 ds_bpermute_b32) that turns them into fragment registers
//   planar frag: `frag` lanes, but the table is stored [piece][row][16 B] so that consecutive rows of one piece are contiguous
// and reports bytes / cycle / CU for each. build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/probe/gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows: int32 [n_groups][16] row ids; every wave walks groups wave, wave + n_waves, ...
template <int SHAPE>
__global__ void __launch_bounds__(256) probe(const float *__restrict__ table, const int32_t *__restrict__ rows, int n_groups, int n_rows,
                                             int iters, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * 256) >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        for (int gi = wave; gi < n_groups; gi += n_waves) {
            const int32_t *grp = rows + (size_t)gi * 16;
            f32x4 a, b;
            if (SHAPE == 0) {                                   // frag
                const int r = grp[lane & 15], g = lane >> 4;
                const float *p = table + (size_t)r * 32 + g * 8;
                a = *reinterpret_cast<const f32x4 *>(p);
                b = *reinterpret_cast<const f32x4 *>(p + 4);
            } else if (SHAPE == 1 || SHAPE == 3) {              // quad (+ perm)
                const int r = grp[lane >> 2], q = lane & 3;
                const float *p = table + (size_t)r * 32 + q * 4;
                a = *reinterpret_cast<const f32x4 *>(p);
                b = *reinterpret_cast<const f32x4 *>(p + 16);
                if (SHAPE == 3) {
                    // 4 x 16 lane transpose: fragment lane (row tr, k-group tg) takes pieces tg and tg + 4 of its row, i.e. registers
                    // a and b of quad lane 4 tr + tg (the k order inside a 32-channel block is free: the weights are packed to match)
                    const int src = (4 * (lane & 15) + (lane >> 4)) << 2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a[k])));
                        b[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b[k])));
                    }
                }
            } else if (SHAPE == 2) {                            // line
                const int q = lane & 7;
                const int r0 = grp[lane >> 3], r1 = grp[8 + (lane >> 3)];
                a = *reinterpret_cast<const f32x4 *>(table + (size_t)r0 * 32 + q * 4);
                b = *reinterpret_cast<const f32x4 *>(table + (size_t)r1 * 32 + q * 4);
            } else if (SHAPE == 5) {                            // line-shaped gather DIRECT TO LDS (global_load_lds_dwordx4): no VGPR write-back
                extern __shared__ __attribute__((aligned(16))) char lds[];
                const int q = lane & 7;
                const int r0 = grp[lane >> 3], r1 = grp[8 + (lane >> 3)];
                char *base = lds + (threadIdx.x >> 6) * 2048;                       // 2 KB per wave: two 1 KB instructions
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + (size_t)r0 * 32 + q * 4),
                                                 (__attribute__((address_space(3))) void *)base, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + (size_t)r1 * 32 + q * 4),
                                                 (__attribute__((address_space(3))) void *)(base + 1024), 16, 0, 0);
                a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a;
            } else {                                            // planar frag: table [8 pieces][n_rows][4 floats]
                const int r = grp[lane & 15], g = lane >> 4;
                a = *reinterpret_cast<const f32x4 *>(table + ((size_t)(2 * g) * n_rows + r) * 4);
                b = *reinterpret_cast<const f32x4 *>(table + ((size_t)(2 * g + 1) * n_rows + r) * 4);
            }
            acc += a * b;
        }
    }
    if (SHAPE == 5) {
        extern __shared__ __attribute__((aligned(16))) char lds[];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] += *reinterpret_cast<const float *>(lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 2048);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[threadIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    const int n_rows = argc > 1 ? atoi(argv[1]) : 1 << 16;            // 64 K rows x 128 B = 8 MB: L2 / MALL resident
    const int n_groups = 1 << 18;                                     // 4 M row fetches per pass = 512 MB of gathers
    const int iters = 4;
    float *table, *sink;
    int32_t *rows;
    hipMalloc(&table, (size_t)n_rows * 128); hipMalloc(&sink, 4096); hipMalloc(&rows, (size_t)n_groups * 16 * 4);
    hipMemset(table, 0, (size_t)n_rows * 128);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"frag (16 rows x 4 pieces per instr)", "quad (64 B contiguous per quad)", "line (128 B per 8 lanes)", "quad + lane transpose to frag",
                           "planar frag", "line, direct to LDS (global_load_lds)"};
    // row patterns: random rows | runs of R consecutive rows at random places (what x-runs of a sparse level look like)
    for (int run : {1, 4}) {
        std::vector<int32_t> h((size_t)n_groups * 16);
        srand(7);
        for (size_t g = 0; g < (size_t)n_groups; ++g)
            for (int s = 0; s < 16; s += run) {
                // a neighbourhood: groups that follow each other draw from the same 4096-row window (the kernels' gathers are local)
                const int base = (int)((g * 37) % (size_t)(n_rows - 4096 - 16)) + rand() % 4096;
                for (int k = 0; k < run; ++k) h[g * 16 + s + k] = base + k;
            }
        hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int wgs_per_cu : {2, 6, 8}) {
            for (int shape = 0; shape < 6; ++shape) {
                const dim3 grid(cus * wgs_per_cu), block(256);
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0, 0);
                    switch (shape) {
                        case 0: hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink); break;
                        case 1: hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink); break;
                        case 2: hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink); break;
                        case 3: hipLaunchKernelGGL(probe<3>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink); break;
                        case 4: hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink); break;
                        default: hipLaunchKernelGGL(probe<5>, grid, block, 8192, 0, table, rows, n_groups, n_rows, iters, sink); break;
                    }
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                const double bytes = (double)n_groups * 16 * 128 * iters;
                printf("run %2d  %d WG/CU  %-38s %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  %5.2f instr-pairs/us/CU\n", run, wgs_per_cu, names[shape],
                       best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / cus / 2.4e9, (double)n_groups * iters / (best * 1e3) / cus);
            }
        }
    }
    return 0;
}
