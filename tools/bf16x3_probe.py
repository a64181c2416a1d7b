#!/usr/bin/env python3
"""fp32-MFMA vs split-bf16 workgroup kernel on the dense BEV shapes: time and error vs float64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops

SHAPES = [("bev0_128to128", 188, 128, 128, 3, 1), ("bev0_256to128", 188, 256, 128, 3, 1), ("bev1_128to256_s2", 188, 128, 256, 3, 2),
          ("bev1_256to256", 94, 256, 256, 3, 1), ("de1_256to1024", 94, 256, 1024, 1, 1),
          ("shared_512to64", 188, 512, 64, 3, 1), ("head1_64to320", 188, 64, 320, 3, 1)]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
for name, hw, cin, cout, k, stride in SHAPES:
    if k == 3:
        nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, stride, 1, "cuda"); kv = 9
    else:
        nbr, ho, wo, kv = None, hw, hw, 1
    n_in, n_out = batch * hw * hw, batch * ho * wo
    x = torch.randn(n_in, cin, device="cuda") * 3.0
    w = torch.randn(kv, cin, cout, device="cuda") * (2.0 / (kv * cin)) ** 0.5
    pw = ops.pack_weight(w)
    res = {}
    for mode in (False, True):
        out = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=True, bf16x3=mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=True, bf16x3=mode, out=out)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 5, out)
    # float64 reference on a row sample
    rows = torch.randint(0, n_out, (4096,), device="cuda")
    if nbr is None:
        ref = x[rows].double() @ w[0].double()
    else:
        idx = nbr[:, rows].long()
        xp = torch.cat([x, x.new_zeros(1, cin)]).double()
        ref = sum(xp[torch.where(idx[t] < 0, n_in, idx[t])] @ w[t].double() for t in range(kv))
    fl = 2.0 * n_out * kv * cin * cout
    msg = "%-18s rows %7d" % (name, n_out)
    for mode in (False, True):
        ms, out = res[mode]
        err = (out[rows].double() - ref).abs().max().item()
        msg += "  | %s %7.1f us %6.1f TF  max|err| %.2e" % ("bf16x3" if mode else "f32   ", ms * 1e3, fl / ms / 1e9, err)
    print(msg + "  (max|ref| %.1f)" % ref.abs().max().item())
