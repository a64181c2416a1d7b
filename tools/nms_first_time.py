import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from cpd_amd import ops
from cpd_amd.synthetic import random_boxes
B, cap = 16, 4096
bb = np.zeros((B, cap, 7), np.float32)
for i in range(B):
    b, s = random_boxes(i, cap, span=75.0)
    bb[i] = b[np.argsort(-s, kind="stable")]
dbb = torch.from_numpy(bb).cuda(); counts = torch.full((B,), cap, dtype=torch.int32, device="cuda")
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("full nms_batch                 %.0f us" % timed(lambda: ops.nms_batch(dbb, counts, 0.8)))
print("nms_batch_first (200 of 512)   %.0f us" % timed(lambda: ops.nms_batch_first(dbb, counts, 0.8, 200, 512)))
keep, num, inc = ops.nms_batch_first(dbb, counts, 0.8, 200, 512)
print("flagged:", inc.tolist())
print("nms_batch_where, no sample     %.0f us" % timed(lambda: ops.nms_batch_where(dbb, counts, inc, 0.8, keep, num)))
