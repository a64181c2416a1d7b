#!/usr/bin/env python3
"""Runs a few cpd_gather_conv launches of one BEV shape for rocprofv3 --pmc passes (GPU box only).
usage: pmc_conv.py <batch> <hw> <cin> <cout> <wg 0|1> <a> <b>   (a,b = ms,nt or bm,bn)"""
import os
os.environ["CPD_TUNE"] = "1"      # CPD_GC_* knobs are only read with this set

import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops  # noqa: E402

batch, hw, cin, cout, wg, a, b = [int(v) for v in sys.argv[1:8]]
if wg:
    os.environ["CPD_GC_WG"], os.environ["CPD_GC_BM"], os.environ["CPD_GC_BN"] = "1", str(a), str(b)
else:
    os.environ["CPD_GC_WG"], os.environ["CPD_GC_MS"], os.environ["CPD_GC_NT"] = "0", str(a), str(b)
nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, 1, 1, "cuda")
n = batch * hw * hw
x = torch.randn((n, cin), device="cuda")
packed = ops.pack_weight(torch.randn((9, cin, cout), device="cuda") * 0.05)
out = torch.empty((n, cout), device="cuda")
print(ops.gather_conv_tile(n, cin, cout, cin))
for _ in range(3):
    ops.gather_conv(x, cin, packed, nbr, 9, n, cout, out=out)
torch.cuda.synchronize()
