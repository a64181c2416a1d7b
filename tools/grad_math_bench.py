#!/usr/bin/env python3
"""Gradient kernels of one dense 128->128 3x3 layer at ONE frame (config 3): BatchNorm backward with / without the max |dz| word,
input gradient and weight gradient in split-bf16 vs scaled split-fp16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops, train_ops as T

torch.manual_seed(0)


def timeit(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for hw, c in ((188, 128), (94, 256)):
    n = hw * hw
    nbr, _, _ = ops.rulebook_conv2d(1, hw, hw, 3, 3, 1, 1, "cuda")
    x = torch.randn(n, c, device="cuda")
    dy = torch.randn(n, c, device="cuda") * 1e-5
    mean, invstd = x.mean(0), (x.var(0, unbiased=False) + 1e-3).rsqrt()
    gamma = torch.rand(c, device="cuda") + 0.5
    y = torch.relu((x - mean) * invstd * gamma)
    am = torch.zeros(T.ABSMAX_WORDS, dtype=torch.int32, device="cuda")
    w = torch.randn(9, c, c, device="cuda") * 0.03
    pw = T.pack_weight_adjoint(w, flip_taps=True)
    dw = torch.zeros(9, c, c, device="cuda")
    print("%dx%d x %d" % (hw, hw, c))
    print("  bn_backward            %7.1f us" % timeit(lambda: T.bn_backward(dy, y, x, mean, invstd, gamma)))
    def with_word():
        am.zero_()
        T.bn_backward(dy, y, x, mean, invstd, gamma, dx_absmax=am)
    print("  bn_backward + word     %7.1f us" % timeit(with_word))
    dz = T.bn_backward(dy, y, x, mean, invstd, gamma, dx_absmax=am)[0]
    for math, kw in (("bf16x3", {}), ("f16x2", {"in_absmax": am})):
        print("  dgrad %-6s (%s) %7.1f us" % (math, ops.gather_conv_tile(n, c, c, c, dense=True, math=math, nbr=nbr),
                                           timeit(lambda: ops.gather_conv(dz, c, pw, nbr, 9, n, c, dense=True, math=math, **kw))))
    for math, kw in (("bf16x3", {}), ("f16x2", {"dy_absmax": am})):
        print("  wgrad %-6s %7.1f us" % (math, timeit(lambda: T.conv_wgrad(x, c, dz, c, nbr, 9, n, dw=dw, math=math, **kw))))
