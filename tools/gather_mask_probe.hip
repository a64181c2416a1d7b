// What does a PARTLY active gather cost? (VERDICT r5 #1b: an x-line form of the row-wave kernels would replace the dx = +-1 gathers of a
// 16-row tile by lane shifts of the dx = 0 rows plus PATCH gathers of the ~1.5 rows per tile whose x-neighbour is not the next row.)
// The kernels' gather shape: quad-shaped buffer loads, a quad of lanes = 64 contiguous bytes of one row, 16 rows x 128 B per pair of
// raw_buffer_load_b128. Here K of the 16 rows of every instruction pair are real and 16 - K are
//   oob   : addressed beyond num_records of the buffer resource (the hardware returns zeros without a memory access), or
//   exec  : switched off in EXEC (a divergent branch around the load)
// and the probe reports instruction pairs / us / CU and the bytes actually fetched per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_mask_probe.hip -o tools/probe/gather_mask_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: oob rows, 1: exec-masked rows
__global__ void __launch_bounds__(256) probe(const float *__restrict__ table, const int32_t *__restrict__ rows, int n_groups, int n_rows, int iters,
                                             float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * 256) >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, (int)((uint32_t)n_rows * 128u), 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        for (int gi = wave; gi < n_groups; gi += n_waves) {
            const int r = rows[(size_t)gi * 16 + (lane >> 2)];            // -1: not a real row
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (MODE == 0) {
                const uint32_t ru = (uint32_t)r < (uint32_t)n_rows ? (uint32_t)r : (uint32_t)n_rows;
                const uint32_t off = ru * 128u + (uint32_t)(lane & 3) * 16u;
                a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 64, 0));
            } else if (r >= 0) {
                const uint32_t off = (uint32_t)r * 128u + (uint32_t)(lane & 3) * 16u;
                a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 64, 0));
            }
            acc += a * b;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[threadIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    const int n_rows = argc > 1 ? atoi(argv[1]) : 1 << 14;            // 16 K rows x 128 B = 2 MB: fits one XCD's L2
    // argv[2] = W > 0: every row id is drawn from the FIXED window [0, W) -- W = 64 / 128 rows = 8 / 16 KB: what every CU's 32 KB vector L1
    // retains (is an L1 hit any cheaper than an L2 hit for this shape?); 0: the moving 4096-row neighbourhoods of the default
    const int fixed_window = argc > 2 ? atoi(argv[2]) : 0;
    const int n_groups = 1 << 18;
    const int iters = 4;
    float *table, *sink;
    int32_t *rows;
    hipMalloc(&table, (size_t)n_rows * 128); hipMalloc(&sink, 4096); hipMalloc(&rows, (size_t)n_groups * 16 * 4);
    hipMemset(table, 0, (size_t)n_rows * 128);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("table %d rows (%.1f MB), fixed window %d rows\n", n_rows, n_rows * 128 / 1048576.0, fixed_window);
    for (int keep : {16, 8, 4, 2, 1, 0}) {
        if (fixed_window > 0 && keep != 16 && keep != 1) continue;
        std::vector<int32_t> h((size_t)n_groups * 16);
        srand(7);
        for (size_t g = 0; g < (size_t)n_groups; ++g) {
            const int first = rand() % 16;                             // which rows of the group are real: `keep` consecutive ones from a random start
            for (int s = 0; s < 16; ++s) {
                const int base = fixed_window > 0 ? rand() % fixed_window : (int)((g * 37) % (size_t)(n_rows - 4096 - 16)) + rand() % 4096;
                h[g * 16 + s] = ((s - first + 16) % 16) < keep ? base : -1;
            }
        }
        hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int wgs_per_cu : {6, 8}) {
            for (int mode = 0; mode < 2; ++mode) {
                const dim3 grid(cus * wgs_per_cu), block(256);
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0, 0);
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink);
                    else hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, table, rows, n_groups, n_rows, iters, sink);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                const double bytes = (double)n_groups * keep * 128 * iters;
                printf("rows/instr-pair %2d  %d WG/CU  %-5s %8.1f us  %6.2f instr-pairs/us/CU  %5.1f fetched B/clk/CU (2.4 GHz)\n", keep, wgs_per_cu,
                       mode == 0 ? "oob" : "exec", best * 1e3, (double)n_groups * iters / (best * 1e3) / cus, bytes / (best * 1e-3) / cus / 2.4e9);
            }
        }
    }
    return 0;
}
