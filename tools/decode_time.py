#!/usr/bin/env python3
"""HIP-event time of the decode tail's launches at 1 / 4 / 48 frames: center_decode (topk_class + decode kernels), nms_batch, select_boxes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops
from cpd_amd.engine import ModelConfig
cfg = ModelConfig()
h = w = 188
ld = 16
def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (1, 4, 48):
    torch.manual_seed(B)
    rows = torch.randn(B * h * w, ld, device="cuda")
    rows[:, 8:11] = rows[:, 8:11] * 1.5 - 2.0
    def dec():
        return ops.center_decode(rows[:, 8:], rows[:, 0:], rows[:, 2:], rows[:, 3:], rows[:, 6:], ld, 1, 3, h, w, 500, 8.0, cfg.voxel_size[:2],
                                 cfg.point_cloud_range[:2], cfg.post_center_limit_range, 0.1, sync=False, batch=B, sample_stride=h * w * ld)
    boxes, scores, labels, counts = dec()
    t_dec = timed(dec)
    t_nms = timed(lambda: ops.nms_batch(boxes, counts, 0.8))
    print("%2d frame(s): center_decode %.1f us   nms_batch %.1f us   (boxes per frame %s)" % (B, t_dec, t_nms, counts[:3].tolist()))
