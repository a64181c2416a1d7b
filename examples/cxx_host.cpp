// examples/cxx_host.cpp -- a host with no Python and no torch driving the C-ABI (include/cpd_hip.h):
//   points (binary f32 file) -> cpd_voxelize (+ fused MeanVFE) -> cpd_index_build -> cpd_rulebook_subm
//   -> cpd_pack_weight -> cpd_gather_conv (SubMConv3d 5 -> 16, weights from a binary f32 file)
// and prints the voxel count and two checksums that tests/test_gpu_cxx_host.py compares with the Python
// host path on the same inputs. Round 4: the launch chain BETWEEN the count read-backs (index build -> rulebook -> conv; every entry
// point takes a stream, allocates nothing and never synchronises: include/cpd_hip.h) is also captured into a hipGraph and replayed --
// same bits out, and the per-replay time printed next to the direct launches' (fourth to sixth output tokens).
// Build: hipcc examples/cxx_host.cpp -Iinclude -Lcpd_amd/csrc -lcpd_hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cpd_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CPD(x) do { int rc_ = (x); if (rc_ != CPD_OK) { fprintf(stderr, "%s failed: %d\n", #x, rc_); return 3; } } while (0)

static std::vector<float> read_f32(const char *path) {
    std::vector<float> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s points.f32 weights.f32\n", argv[0]); return 1; }
    const int C = 5, P = 5, MAXV = 1000000, COUT = 16, KV = 27;
    const float vs[3] = {0.1f, 0.1f, 0.15f}, rg[6] = {-75.2f, -75.2f, -2.0f, 75.2f, 75.2f, 4.0f};
    std::vector<float> pts = read_f32(argv[1]), w = read_f32(argv[2]);
    const int n = (int)(pts.size() / C);
    if (n <= 0 || (int)w.size() != KV * C * COUT) { fprintf(stderr, "bad inputs\n"); return 1; }

    float *d_pts, *d_mean, *d_w, *d_pw, *d_out;
    int32_t *d_coords, *d_num, *d_nvox, *d_nbr;
    uint32_t *d_mask;
    void *d_ws, *d_index;
    CHECK_HIP(hipMalloc(&d_pts, pts.size() * 4));
    CHECK_HIP(hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
    const int cap = n < MAXV ? n : MAXV;
    CHECK_HIP(hipMalloc(&d_mean, (size_t)cap * C * 4));
    CHECK_HIP(hipMalloc(&d_coords, (size_t)cap * 16));
    CHECK_HIP(hipMalloc(&d_num, (size_t)cap * 4));
    CHECK_HIP(hipMalloc(&d_nvox, 4));
    const size_t ws_bytes = cpd_voxelize_workspace_bytes(n, P, MAXV, vs, rg);
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_CPD(cpd_voxelize(d_pts, n, C, vs, rg, P, MAXV, /*batch_idx=*/0, /*coord_cols=*/4, /*voxels=*/nullptr, d_coords, d_num, d_mean,
                           d_nvox, d_ws, ws_bytes, nullptr));
    int32_t m = 0;
    CHECK_HIP(hipMemcpy(&m, d_nvox, 4, hipMemcpyDeviceToHost));

    int32_t grid[3];
    CHECK_CPD(cpd_voxel_grid_size(vs, rg, grid));
    const int32_t shape[3] = {grid[0] + 1, grid[1], grid[2]}, k3[3] = {3, 3, 3};       // spconv_backbone.py:412
    const size_t index_bytes = cpd_index_bytes(1, shape, m);
    CHECK_HIP(hipMalloc(&d_index, index_bytes));
    CHECK_CPD(cpd_index_build(d_coords, m, 1, shape, d_index, index_bytes, nullptr));
    CHECK_HIP(hipMalloc(&d_nbr, (size_t)KV * m * 4));
    CHECK_HIP(hipMalloc(&d_mask, (size_t)((m + 15) / 16) * 4));
    CHECK_CPD(cpd_rulebook_subm(d_coords, m, 1, shape, k3, d_index, d_nbr, d_mask, nullptr));

    CHECK_HIP(hipMalloc(&d_w, w.size() * 4));
    CHECK_HIP(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&d_pw, cpd_packed_weight_floats(KV, C, COUT) * 4));
    CHECK_CPD(cpd_pack_weight(d_w, KV, C, COUT, d_pw, nullptr));
    CHECK_HIP(hipMalloc(&d_out, (size_t)m * COUT * 4));
    CHECK_CPD(cpd_gather_conv(d_mean, C, m, C, d_pw, d_nbr, d_mask, KV, m, COUT, nullptr, nullptr, nullptr, 0, /*relu=*/1, d_out, COUT,
                              nullptr, 0, 0, nullptr));
    CHECK_HIP(hipDeviceSynchronize());

    std::vector<float> mean((size_t)m * C), out((size_t)m * COUT);
    CHECK_HIP(hipMemcpy(mean.data(), d_mean, mean.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    double s_mean = 0, s_out = 0;
    for (float v : mean) s_mean += v;
    for (float v : out) s_out += v;

    // ---- the same chain as a hipGraph: capture once, replay
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    auto chain = [&](hipStream_t q) -> int {
        CHECK_CPD(cpd_index_build(d_coords, m, 1, shape, d_index, index_bytes, q));
        CHECK_CPD(cpd_rulebook_subm(d_coords, m, 1, shape, k3, d_index, d_nbr, d_mask, q));
        CHECK_CPD(cpd_gather_conv(d_mean, C, m, C, d_pw, d_nbr, d_mask, KV, m, COUT, nullptr, nullptr, nullptr, 0, 1, d_out, COUT, nullptr, 0, 0, q));
        return 0;
    };
    CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)m * COUT * 4, st));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    if (int rc = chain(st)) return rc;
    CHECK_HIP(hipStreamEndCapture(st, &graph));
    CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK_HIP(hipGraphLaunch(exec, st));
    CHECK_HIP(hipStreamSynchronize(st));
    std::vector<float> out_g((size_t)m * COUT);
    CHECK_HIP(hipMemcpy(out_g.data(), d_out, out_g.size() * 4, hipMemcpyDeviceToHost));
    double s_graph = 0;
    bool same = true;
    for (size_t i = 0; i < out_g.size(); ++i) { s_graph += out_g[i]; same = same && out_g[i] == out[i]; }
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    const int reps = 50;
    float ms_graph = 0, ms_direct = 0;
    CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CHECK_HIP(hipGraphLaunch(exec, st));
    CHECK_HIP(hipEventRecord(e1, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipEventElapsedTime(&ms_graph, e0, e1));
    CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i)
        if (int rc = chain(st)) return rc;
    CHECK_HIP(hipEventRecord(e1, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipEventElapsedTime(&ms_direct, e0, e1));
    printf("%d %.9e %.9e %s %.1f %.1f\n", m, s_mean, s_out, same ? "graph_bitwise_equal" : "graph_DIFFERS", 1e3 * ms_graph / reps, 1e3 * ms_direct / reps);
    (void)s_graph;
    return same ? 0 : 4;
}
