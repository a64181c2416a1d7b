#!/usr/bin/env python3
"""Tile sweep for cpd_gather_conv on the BEV / sparse layer shapes (GPU box only).
For every (ms, nt) instantiation: average launch time (HIP events on the launch stream) and
algorithmic TFLOP/s. Output: one JSON line per shape."""
import json
import os
os.environ["CPD_TUNE"] = "1"      # CPD_GC_* knobs are only read with this set

import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops  # noqa: E402

SHAPES = [  # name, batch*h*w rows (h, w), c_in, c_out, k, stride
    ("bev0_256to128_188", 188, 256, 128, 3, 1),
    ("bev0_128to128_188", 188, 128, 128, 3, 1),
    ("bev1_128to256_s2", 188, 128, 256, 3, 2),
    ("bev1_256to256_94", 94, 256, 256, 3, 1),
    ("de0_128to256_1x1", 188, 128, 256, 1, 1),
    ("de1_256to1024_1x1", 94, 256, 1024, 1, 1),
    ("shared_512to64", 188, 512, 64, 3, 1),
    ("head1_64to320", 188, 64, 320, 3, 1),
    ("head2_320to11", 188, 320, 11, 3, 1),
]


def time_variant(x, cin, packed, nbr, kv, n_out, cout, ms, nt, iters=5, wg=False):
    if wg:
        os.environ["CPD_GC_WG"], os.environ["CPD_GC_BM"], os.environ["CPD_GC_BN"] = "1", str(ms), str(nt)
        want = "tile_conv_kernel<%d,%d>" % (ms, nt)
    else:
        os.environ["CPD_GC_WG"] = "0"
        os.environ["CPD_GC_MS"], os.environ["CPD_GC_NT"] = str(ms), str(nt)
        want = "gather_conv_kernel<%d,%d,true>" % (ms, nt)
    if ops.gather_conv_tile(n_out, cin, cout, x.stride(0)) != want:
        return None
    out = torch.empty((n_out, cout), device="cuda")
    ops.gather_conv(x, cin, packed, nbr, kv, n_out, cout, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gather_conv(x, cin, packed, nbr, kv, n_out, cout, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,4").split(",")]
    for batch in batches:
        for name, hw, cin, cout, k, stride in SHAPES:
            pad = 1 if k == 3 else 0
            if k == 3:
                nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, k, k, stride, pad, "cuda")
                kv = 9
            else:
                nbr, ho, wo, kv = None, hw, hw, 1
            n_in, n_out = batch * hw * hw, batch * ho * wo
            x = torch.randn((n_in, cin), device="cuda")
            packed = ops.pack_weight(torch.randn((kv, cin, cout), device="cuda") * 0.05)
            pairs = int((nbr >= 0).sum()) if nbr is not None else n_out
            flops = 2.0 * pairs * cin * cout
            res = {}
            for ms in (1, 2, 4):
                for nt in (1, 2, 4, 5, 8):
                    t = time_variant(x, cin, packed, nbr, kv, n_out, cout, ms, nt)
                    if t is not None:
                        res["%d,%d" % (ms, nt)] = [round(t * 1e3, 1), round(flops / t / 1e9, 1)]
            for bm in (64, 128):
                for bn in (64, 128):
                    t = time_variant(x, cin, packed, nbr, kv, n_out, cout, bm, bn, wg=True)
                    if t is not None:
                        res["wg%dx%d" % (bm, bn)] = [round(t * 1e3, 1), round(flops / t / 1e9, 1)]
            best = max(res.items(), key=lambda kv_: kv_[1][1])
            print(json.dumps({"shape": name, "batch": batch, "n_out": n_out, "gflop": round(flops / 1e9, 2), "best": best,
                              "us_tflops": res}), flush=True)
    


if __name__ == "__main__":
    main()
