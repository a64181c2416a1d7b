// How expensive are N workgroups' atomicMax calls on S device words spaced `stride` bytes apart? (the max |dz| word of the
// BatchNorm backward: 4418 workgroups, one guarded atomic each.) build: hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/probe/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) k(uint32_t *w, int slots, int stride_words, int guard, float *sink) {
    // a little streaming work so that the launch looks like the real kernel
    float v = sink[(size_t)blockIdx.x * 256 + threadIdx.x];
    uint32_t m = __float_as_uint(fabsf(v));
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)m, o); m = t > m ? t : m; }
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (slots == 0) return;
        uint32_t *p = w + (size_t)(blockIdx.x % slots) * stride_words;
        if (!guard || m > *reinterpret_cast<volatile uint32_t *>(p)) atomicMax(p, m);
    }
}
int main() {
    const int blocks = 4418;
    uint32_t *w; float *sink;
    hipMalloc(&w, 64 << 20); hipMalloc(&sink, (size_t)blocks * 256 * 4);
    float *h = (float *)malloc((size_t)blocks * 256 * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)blocks * 256; ++i) { float a = 0; for (int k = 0; k < 12; ++k) a += rand() / (float)RAND_MAX - 0.5f; h[i] = a * 1e-5f; }
    hipMemcpy(sink, h, (size_t)blocks * 256 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[][2] = {{1, 1}, {16, 1}, {64, 1}, {16, 32}, {16, 64}, {64, 64}, {16, 1024}, {64, 1024}, {16, 1088}, {64, 1088}, {256, 64}};
    for (int guard = 0; guard < 2; ++guard)
        for (auto &c : cfg) {
            float tot = 0;
            const int reps = 30;
            for (int r = 0; r < reps + 3; ++r) {
                hipMemsetAsync(w, 0, 64 << 20, 0);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, w, c[0], c[1], guard, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 3) tot += ms;
            }
            printf("guard %d slots %3d stride %5d B: %6.1f us\n", guard, c[0], c[1] * 4, tot / reps * 1e3);
        }
    {   // no atomics at all
        float tot = 0;
        for (int r = 0; r < 33; ++r) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, w, 0, 1, 0, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 3) tot += ms;
        }
        printf("no atomics: %6.1f us\n", tot / 30 * 1e3);
    }
    return 0;
}
