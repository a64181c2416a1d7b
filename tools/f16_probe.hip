// f16_probe.hip -- does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs on gfx950, and what do the
// f32 -> f16 conversions the split uses produce (rounding, overflow)?   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float *a_vals, const float *b_vals, float *out, float *cv) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // row r = lane & 15 of A gets a_vals[r] at k = 0 (lanes with g = 0); column n of B gets b_vals[n] at k = 0
    if ((lane >> 4) == 0) { a[0] = (_Float16)a_vals[lane & 15]; b[0] = (_Float16)b_vals[lane & 15]; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = c[i];
    if (lane < 16) {
        float x = a_vals[lane];
        _Float16 h = (_Float16)x;
        float r = x - (float)h;
        _Float16 l = (_Float16)r;
        cv[lane * 4 + 0] = (float)h; cv[lane * 4 + 1] = (float)l; cv[lane * 4 + 2] = x - ((float)h + (float)l);
        f2 xx = {x, x * 1.0001f};
        h2 hh = __builtin_convertvector(xx, h2);
        cv[lane * 4 + 3] = (float)hh[1];
    }
}
int main() {
    float ha[16], hb[16], ho[256], hc[64];
    // subnormal fp16 range is < 6.1e-5 (2^-14); smallest 2^-24 = 5.96e-8
    const float av[16] = {1.f, 3.0e-5f, 1.0e-5f, 1.0e-6f, 1.2e-7f, 6.0e-8f, 65504.f, 70000.f, 1e-3f, 0.1f, 0.3333333f, 1234.567f, -2.5e-5f, 2.0e-4f, 3.14159265f, 1e-8f};
    for (int i = 0; i < 16; ++i) { ha[i] = av[i]; hb[i] = (i % 2) ? 1024.f : 1.f; }
    float *da, *db, *dout, *dc;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 1024); hipMalloc(&dc, 256);
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout, dc);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 256, hipMemcpyDeviceToHost);
    printf("MFMA f16 subnormal handling: out[r][n] = a[r]*b[n] (k=0 only)\n");
    for (int r = 0; r < 16; ++r)
        printf("a=% .6e  fp16(a)=% .6e  mfma*1=% .6e  mfma*1024/1024=% .6e  lo=% .6e resid=% .3e cvtvec=% .6e\n", ha[r], hc[r * 4], ho[r * 16 + 0],
               ho[r * 16 + 1] / 1024.f, hc[r * 4 + 1], hc[r * 4 + 2], hc[r * 4 + 3]);
    return 0;
}
