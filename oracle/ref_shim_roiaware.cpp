// ref_shim_roiaware.cpp -- TEST INFRASTRUCTURE ONLY.
// C-ABI doorway into the REFERENCE's own CPU points-in-boxes test
// (cpd/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:143-168), compiled from where it lies under
// /root/reference by oracle/Makefile into oracle/_ref/. No reference code here: the entry point is
// declared and raw host pointers are wrapped into at::Tensor views. The reference file also
// declares three CUDA launchers that only its GPU entry points call; they stay unresolved (the
// library is linked with --unresolved-symbols=ignore-all and loaded lazily), nothing stands in for them.
#include <torch/torch.h>

int points_in_boxes_cpu(at::Tensor boxes_tensor, at::Tensor pts_tensor, at::Tensor pts_indices_tensor);

extern "C" int ref_points_in_boxes_cpu(const float* boxes, int n, const float* pts, int m, int* out) {
    at::Tensor tb = at::from_blob(const_cast<float*>(boxes), {n, 7}, at::TensorOptions().dtype(at::kFloat));
    at::Tensor tp = at::from_blob(const_cast<float*>(pts), {m, 3}, at::TensorOptions().dtype(at::kFloat));
    at::Tensor to = at::from_blob(out, {n, m}, at::TensorOptions().dtype(at::kInt));
    return points_in_boxes_cpu(tb, tp, to);
}
