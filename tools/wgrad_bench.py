#!/usr/bin/env python3
"""Weight-gradient kernel on the dense layer shapes of the train step (config 3: ONE frame per GPU), split-bf16 arithmetic:
time per call and useful TFLOP/s as a function of the number of row chunks (CPD_WGRAD_WGS = workgroups to aim for).
usage: CPD_TUNE=1 python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops, train_ops

os.environ["CPD_TUNE"] = "1"
torch.manual_seed(0)


def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [("128->128 3x3 188^2", 188, 128, 128), ("256->256 3x3 94^2", 94, 256, 256), ("256->128 3x3 188^2", 188, 256, 128),
         ("64->64 3x3 188^2", 188, 64, 64)]
for name, hw, ci, co in cases:
    nbr, _, _ = ops.rulebook_conv2d(1, hw, hw, 3, 3, 1, 1, "cuda")
    n = hw * hw
    x = torch.randn(n, ci, device="cuda")
    dy = torch.randn(n, co, device="cuda")
    dw = torch.zeros(9, ci, co, device="cuda")
    flop = 2.0 * 9 * n * ci * co
    for wgs in os.environ.get("WGS", "0,256,512,1024,2048,4096").split(","):
        if wgs == "0":
            os.environ.pop("CPD_WGRAD_WGS", None)
        else:
            os.environ["CPD_WGRAD_WGS"] = wgs
        us = timeit(lambda: train_ops.conv_wgrad(x, ci, dy, co, nbr, 9, n, dw=dw, bf16x3=True))
        print("%-22s WGS %-5s %8.1f us  %6.1f TF useful" % (name, wgs, us, flop / us / 1e6), flush=True)
