"""Diagnostic: per-tensor gradient error of the HIP trainer and of a torch-fp32 run of the same
reference graph, both against the float64 reference (small config)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import ref_train_torch as R
from test_gpu_train import small_cfg, scene
from cpd_amd.engine import init_state_dict
from cpd_amd.train_engine import CenterPointTrainer
from oracle.binding import Oracle

cfg = small_cfg(); sd = init_state_dict(cfg, seed=3); pts, gt = scene(); o = Oracle()
tr = CenterPointTrainer(cfg, sd, num_max_objs=50)
loss, parts = tr.forward_backward([torch.from_numpy(p).cuda() for p in pts], torch.from_numpy(gt).cuda())
grads = {k: v.cpu().double().numpy() for k, v in tr.grad_dict().items()}
P = R.make_leaves(sd); l64, _, _ = R.forward_loss(o, cfg, P, pts, gt, 50); l64.backward()
R.F64 = torch.float32
P32 = {k: v.detach().float().requires_grad_(True) for k, v in P.items()}
l32, _, _ = R.forward_loss(o, cfg, P32, pts, gt, 50); l32.backward()
print("loss hip %.6f  f64 %.6f  f32 %.6f" % (float(loss), float(l64), float(l32)))
rows = []
for k, leaf in P.items():
    ref = leaf.grad.numpy(); s = max(np.abs(ref).max(), 1e-12)
    rows.append((np.abs(grads[k] - ref).max() / s, np.abs(P32[k].grad.double().numpy() - ref).max() / s, s, k))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
if not flt:
    rows.sort(reverse=True)
rows = [r for r in rows if flt in r[3]]
for e, e32, s, k in rows[:400]:
    print("%-50s hip %.2e  torch32 %.2e  max|ref| %.2e" % (k, e, e32, s))
