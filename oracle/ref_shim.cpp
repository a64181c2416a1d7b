// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// C-ABI doorway into the REFERENCE's own CPU BEV-IoU (cpd/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252),
// compiled from where it lies under /root/reference by oracle/Makefile into oracle/_ref/. This
// file contains no reference code: it only declares the reference entry point (as its header
// iou3d_cpu.h:9 does) and wraps raw host pointers into at::Tensor views.
#include <torch/torch.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor,
                      at::Tensor ans_iou_tensor);  // iou3d_cpu.h:9

extern "C" int ref_boxes_iou_bev_cpu(const float* a, int n, const float* b, int m, float* out) {
    auto opt = at::TensorOptions().dtype(at::kFloat);
    at::Tensor ta = at::from_blob(const_cast<float*>(a), {n, 7}, opt);
    at::Tensor tb = at::from_blob(const_cast<float*>(b), {m, 7}, opt);
    at::Tensor to = at::from_blob(out, {n, m}, opt);
    return boxes_iou_bev_cpu(ta, tb, to);
}
