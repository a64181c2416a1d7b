#!/usr/bin/env python3
"""A/B: 16x16x4 vs 32x32x2 MFMA in the workgroup kernel (interleaved rounds, correctness checked)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops

def run(batch, hw, cin, cout, bm, bn, rounds=5, iters=5):
    os.environ["CPD_GC_WG"], os.environ["CPD_GC_BM"], os.environ["CPD_GC_BN"] = "1", str(bm), str(bn)
    nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, 1, 1, "cuda")
    n = batch * hw * hw
    x = torch.randn((n, cin), device="cuda")
    packed = ops.pack_weight(torch.randn((9, cin, cout), device="cuda") * 0.05)
    outs = {}
    res = {0: [], 1: []}
    flops = 2.0 * float((nbr >= 0).sum()) * cin * cout
    for r in range(rounds):
        for v in (0, 1):
            os.environ["CPD_GC_MFMA32"] = str(v)
            out = torch.empty((n, cout), device="cuda")
            ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out, dense=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out, dense=True)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / iters)
            outs[v] = out
    err = float((outs[0] - outs[1]).abs().max())
    f = lambda v: "%.1f us %.1f TF" % (min(res[v]) * 1e3, flops / min(res[v]) / 1e9)
    print("b%d %dx%d %d->%d wg%dx%d | 16x16x4: %s | 32x32x2: %s | maxdiff %.2e" % (batch, hw, hw, cin, cout, bm, bn, f(0), f(1), err), flush=True)

for b in (1, 8):
    run(b, 188, 128, 128, 64, 128)
    run(b, 188, 128, 128, 128, 128)
    run(b, 94, 256, 256, 64, 128)
    run(b, 94, 256, 256, 128, 128)
    run(b, 188, 512, 64, 64, 64)
    run(b, 188, 512, 64, 128, 64)
