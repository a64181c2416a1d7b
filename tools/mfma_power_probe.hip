// mfma_power_probe.hip -- sustained MFMA rate and board power of the two f16 MFMA shapes on gfx950, operands held in registers
// (no memory traffic), random vs zero operand data. Tells how much of the split-fp16 conv kernels' gap to the 2.4 GHz peak is
// the power cap, and whether the 32x32x16 shape (half the operand-register reads per MAC) is cheaper than 16x16x32.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <atomic>
#include <string>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 0: 16x16x32, a 64x64 wave tile per k32 step = 4 A + 4 B fragments, 16 MFMAs (8192 MACs each)
// SHAPE 1: 32x32x16, the same 64x64 tile per k32 = two k16 steps of 2 A + 2 B fragments, 4 MFMAs each (16384 MACs each)
template <int SHAPE>
__global__ void __launch_bounds__(256) burn(const h8 *src, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane) % 512]; b[i] = src[((i + 4) * 64 + lane) % 512]; }
    if constexpr (SHAPE == 0) {
        f4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
        f4 s = f4{0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
        out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
        f16v acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * k + i], b[2 * k + j], acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(a[0]), "+v"(b[0]));
        }
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

static std::atomic<bool> stop_flag{false};
static std::string last_smi;
static void sampler() {
    while (!stop_flag) {
        FILE *f = popen("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | head -2 | tr '\\n' ' '", "r");
        if (f) { char buf[512]; std::string s; while (fgets(buf, sizeof buf, f)) s += buf; pclose(f); last_smi = s; }
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
    }
}

template <int SHAPE>
static void run(const char *name, const h8 *src, float *out, double secs) {
    const int blocks = 256 * 8, iters = 4000;
    burn<SHAPE><<<blocks, 256>>>(src, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    int reps = 0; float ms_total = 0.f;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) burn<SHAPE><<<blocks, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms; reps += 5;
    }
    const double macs = (double)blocks * 4 * iters * 64.0 * 64.0 * 32.0;     // per launch: waves x iterations x 64x64x32
    printf("%-28s %8.1f TFLOP/s (f16 issue)   smi: %s\n", name, 2.0 * macs * reps / (ms_total * 1e-3) / 1e12, last_smi.c_str());
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    h8 *src_r, *src_z, *src_p; float *out;
    hipMalloc(&src_r, 512 * 16); hipMalloc(&src_z, 512 * 16); hipMalloc(&src_p, 512 * 16); hipMalloc(&out, 256 * 8 * 256 * 4);
    _Float16 hr[512 * 8], hp[512 * 8];
    srand(1);
    for (int i = 0; i < 512 * 8; ++i) {
        float u = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
        hr[i] = (_Float16)u;
        hp[i] = (_Float16)(u > 0.f ? u : 0.f);       // post-ReLU-like: half zeros
    }
    hipMemcpy(src_r, hr, sizeof hr, hipMemcpyHostToDevice);
    hipMemcpy(src_p, hp, sizeof hp, hipMemcpyHostToDevice);
    hipMemset(src_z, 0, 512 * 16);
    std::thread th(sampler);
    run<0>("16x16x32 random", src_r, out, secs);
    run<1>("32x32x16 random", src_r, out, secs);
    run<0>("16x16x32 relu-like", src_p, out, secs);
    run<1>("32x32x16 relu-like", src_p, out, secs);
    run<0>("16x16x32 zeros", src_z, out, secs);
    run<1>("32x32x16 zeros", src_z, out, secs);
    stop_flag = true; th.join();
    return 0;
}
