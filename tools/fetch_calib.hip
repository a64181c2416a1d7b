// What does rocprofv3's FETCH_SIZE report for a GATHER? (VERDICT r4 weak #5.) MI355X_MICROARCH.md calibrates "gfx950 reports half of the
// bytes" on wide coalesced streams; the row-wave kernels' traffic (profiles/r0x_pmc_summary.json: 2 x FETCH_SIZE + WRITE_SIZE) is made
// of 128-byte row gathers. This probe reads a KNOWN number of bytes, every 128-byte line of a 2 GB table exactly once (no reuse: the
// table is 8 x the Infinity Cache), in four shapes, so that FETCH_SIZE / known bytes can be read per kernel from
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/probe/fetch_calib
//   stream   : lanes read consecutive 16-byte pieces (a wave instruction = 1 KB contiguous)
//   rows8    : 8 lanes = one 128-byte row, rows in a random permutation (the window / tile kernels' staging shape)
//   quad2    : a quad = 64 contiguous bytes of one row, two instructions per row half (the row-wave kernels' gather shape)
//   half64   : only the first 64 bytes of every row are read (does a half-line gather fetch 64 or 128 bytes?)
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/probe/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_stream(const f32x4 *__restrict__ t, size_t n16, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += t[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
__global__ void __launch_bounds__(256) k_rows8(const float *__restrict__ t, const int32_t *__restrict__ perm, size_t n_rows, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    const int q = threadIdx.x & 7;
    for (size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 3; g < n_rows; g += ((size_t)gridDim.x * 256) >> 3)
        acc += *reinterpret_cast<const f32x4 *>(t + (size_t)perm[g] * 32 + q * 4);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
__global__ void __launch_bounds__(256) k_quad2(const float *__restrict__ t, const int32_t *__restrict__ perm, size_t n_rows, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    const int q = threadIdx.x & 3;
    for (size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 2; g < n_rows; g += ((size_t)gridDim.x * 256) >> 2) {
        const float *p = t + (size_t)perm[g] * 32 + q * 4;
        acc += *reinterpret_cast<const f32x4 *>(p);
        acc += *reinterpret_cast<const f32x4 *>(p + 16);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
__global__ void __launch_bounds__(256) k_half64(const float *__restrict__ t, const int32_t *__restrict__ perm, size_t n_rows, float *sink) {
    f32x4 acc = {0, 0, 0, 0};
    const int q = threadIdx.x & 3;
    for (size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 2; g < n_rows; g += ((size_t)gridDim.x * 256) >> 2)
        acc += *reinterpret_cast<const f32x4 *>(t + (size_t)perm[g] * 32 + q * 4);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

int main() {
    const size_t n_rows = (size_t)1 << 24;                          // 16 M rows x 128 B = 2 GB
    float *table, *sink;
    int32_t *perm;
    hipMalloc(&table, n_rows * 128); hipMalloc(&sink, 256); hipMalloc(&perm, n_rows * 4);
    hipMemset(table, 0, n_rows * 128);
    std::vector<int32_t> h(n_rows);
    for (size_t i = 0; i < n_rows; ++i) h[i] = (int32_t)i;
    uint64_t s = 88172645463325252ull;                              // xorshift Fisher-Yates: every row exactly once, in random order
    for (size_t i = n_rows - 1; i > 0; --i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const size_t j = s % (i + 1);
        const int32_t tmp = h[i]; h[i] = h[j]; h[j] = tmp;
    }
    hipMemcpy(perm, h.data(), n_rows * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const dim3 grid(prop.multiProcessorCount * 8), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        for (int k = 0; k < 4; ++k) {
            hipEventRecord(e0, 0);
            if (k == 0) hipLaunchKernelGGL(k_stream, grid, block, 0, 0, reinterpret_cast<const f32x4 *>(table), n_rows * 8, sink);
            else if (k == 1) hipLaunchKernelGGL(k_rows8, grid, block, 0, 0, table, perm, n_rows, sink);
            else if (k == 2) hipLaunchKernelGGL(k_quad2, grid, block, 0, 0, table, perm, n_rows, sink);
            else hipLaunchKernelGGL(k_half64, grid, block, 0, 0, table, perm, n_rows, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const char *nm[] = {"k_stream", "k_rows8", "k_quad2", "k_half64"};
            const double bytes = (k == 3 ? 0.5 : 1.0) * (double)n_rows * 128 + (k ? (double)n_rows * 4 : 0.0);
            printf("%-9s known bytes %.0f (table %s + perm)  %.1f us  %.2f TB/s\n", nm[k], bytes, k == 3 ? "half" : "all", ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
