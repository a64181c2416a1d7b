// Diagnostics of the part itself (not on the product path): what a kernel made of NOTHING but split-fp16's MFMAs sustains under this
// socket's power cap. bench.py prints it next to roofline.frac (`power_capped_peak`), so that one line separates "the dominant kernel's
// schedule" from "the matrix pipe's clock under 1400 W" (VERDICT r5 #3a). Same instruction as the window / tile / row-wave kernels
// (v_mfma_f32_16x16x32_f16), operands held in registers, no memory traffic inside the loop -- tools/mfma_power_probe.hip as a C-ABI entry.
#include "common.h"

typedef _Float16 diag_h8 __attribute__((ext_vector_type(8)));
typedef float diag_f4 __attribute__((ext_vector_type(4)));

// a wave = a 64 x 64 tile per k32 step: 4 A + 4 B fragments, 16 MFMAs of 8192 multiply-adds
__global__ void __launch_bounds__(256) mfma_burn_kernel(const diag_h8 *__restrict__ a_src, const diag_h8 *__restrict__ b_src, float *__restrict__ sink,
                                                        int iters) {
    const int lane = threadIdx.x & 63;
    diag_h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = a_src[(i * 64 + lane + 7 * blockIdx.x) & 511];
        b[i] = b_src[(i * 64 + lane + 13 * blockIdx.x) & 511];
    }
    diag_f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = diag_f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    diag_f4 s = diag_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

extern "C" double cpd_mfma_burn_flops(int blocks, int iters) {
    return 2.0 * (double)blocks * 4.0 * (double)iters * 64.0 * 64.0 * 32.0;
}

extern "C" int cpd_mfma_burn(const void *a_operands, const void *b_operands, float *sink, int blocks, int iters, cpd_stream_t stream) {
    if (!a_operands || !b_operands || !sink || blocks <= 0 || iters <= 0) return CPD_ERR_ARG;
    hipLaunchKernelGGL(mfma_burn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const diag_h8 *>(a_operands),
                       static_cast<const diag_h8 *>(b_operands), sink, iters);
    return cpd_check_launch();
}
