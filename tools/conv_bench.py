#!/usr/bin/env python3
"""Micro-benchmark of cpd_gather_conv / cpd_conv3x3_rows on the layer shapes of BASELINE config 2 (16 frames), per conv
arithmetic, with tuning knobs from the environment (CPD_TUNE=1 ...). Prints us / launch and useful TFLOP/s.
usage: python tools/conv_bench.py [dense|sparse|all] [math,math,...] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpd_amd import ops
from cpd_amd.engine import ModelConfig
from cpd_amd.synthetic import waymo_cloud

which = sys.argv[1] if len(sys.argv) > 1 else "all"
maths = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16x3", "f16x2"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(os.environ.get("FRAMES", "16"))
torch.manual_seed(0)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(label, x, cin, w, nbr, kv, n_out, cout, dense, pairs, **kw):
    pw = ops.pack_weight(w)
    ref = None
    for m in maths:
        name = ops.gather_conv_tile(n_out, cin, cout, x.stride(0), dense=dense, nbr=nbr, math=m)
        out = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=dense, math=m, **kw)
        if ref is None:
            ref = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=dense, math="f32", **kw)
        err = float((out - ref).abs().max())
        us = timeit(lambda: ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=dense, math=m, out=out, **kw))
        print("%-26s %-7s %-34s %8.1f us %7.1f TF  maxdiff_vs_f32 %.2e" % (label, m, name, us, 2.0 * pairs * cin * cout / us / 1e6, err), flush=True)
        if m == "f16x2" and not dense and cin % 32 == 0 and cout % 32 == 0 and os.environ.get("PAIRS", "1") != "0":
            # the same layer on fp16-pair rows (CPD_GC_*_PAIRS): input / output / both (+ the residual)
            xp = ops.rows_to_pairs(x)
            kwp = dict(kw)
            if kw.get("residual") is not None:
                kwp["residual"] = ops.rows_to_pairs(kw["residual"])
            for tag, args in (("pairs in", dict(in_pairs=True, **kw)), ("pairs out", dict(out_pairs=True, **kw)),
                              ("pairs in+out+res", dict(in_pairs=True, out_pairs=True, res_pairs=kw.get("residual") is not None, **kwp))):
                o2 = ops.gather_conv(xp if args.get("in_pairs") else x, cin, pw, nbr, kv, n_out, cout, dense=dense, math=m, **args)
                e2 = float(((ops.pairs_to_rows(o2) if args.get("out_pairs") else o2) - ref).abs().max())
                us2 = timeit(lambda: ops.gather_conv(xp if args.get("in_pairs") else x, cin, pw, nbr, kv, n_out, cout, dense=dense, math=m, out=o2, **args))
                print("%-26s %-7s %-34s %8.1f us %7.1f TF  maxdiff_vs_f32 %.2e" % (label, m, "  " + tag, us2, 2.0 * pairs * cin * cout / us2 / 1e6, e2), flush=True)


if which in ("dense", "all"):
    for (cin, cout, hw, k, stride, label) in [(128, 128, 188, 3, 1, "bev0 128->128"), (256, 128, 188, 3, 1, "bev0 first 256->128"),
                                              (256, 256, 94, 3, 1, "bev1 256->256"), (128, 256, 188, 3, 2, "bev1 strided"),
                                              (512, 64, 188, 3, 1, "shared 512->64"), (64, 320, 188, 3, 1, "heads1 64->320"),
                                              (320, 11, 188, 3, 1, "heads2 320->11"), (128, 256, 188, 1, 1, "deblock0 1x1"),
                                              (256, 1024, 94, 1, 1, "deblock1 256->1024")]:
        if k == 3:
            nbr, ho, wo = ops.rulebook_conv2d(B, hw, hw, 3, 3, stride, 1, "cuda"); kv = 9
        else:
            nbr, ho, wo, kv = None, hw, hw, 1
        n_in, n_out = B * hw * hw, B * ho * wo
        x = torch.randn(n_in, cin, device="cuda")
        w = torch.randn(kv, cin, cout, device="cuda") * (2.0 / (kv * cin)) ** 0.5
        pairs = int((nbr >= 0).sum()) if nbr is not None else n_out
        run(label, x, cin, w, nbr, kv, n_out, cout, True, pairs)

if which in ("sparse", "all"):
    # real level geometry: voxelize B clouds, walk the backbone's index chain
    cfg = ModelConfig()
    vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, 5, cfg.max_voxels)
    clouds = [torch.from_numpy(waymo_cloud(s % 8)).cuda() for s in range(B)]
    _, coords, _, feats, nvox = vox.batch(clouds)
    n = int(nvox[B]); coords = coords[:n]
    shape = cfg.sparse_shape
    index = ops.SiteIndex.build(coords, B, shape)
    levels = []
    for (k, s, p, c) in [([3, 3, 3], [2, 2, 2], [1, 1, 1], 32), ([3, 3, 3], [2, 2, 2], [1, 1, 1], 64), ([3, 3, 3], [2, 2, 2], [0, 1, 1], 128)]:
        o_idx, o_index, o_shape = ops.conv_outset(coords, B, shape, k, s, p)
        nbr_dn = ops.rulebook_conv(o_idx, index, k, s, p)
        nbr = ops.rulebook_subm(o_idx, o_index)
        levels.append((c, coords.shape[0], o_idx.shape[0], nbr_dn, nbr))
        coords, index, shape = o_idx, o_index, o_shape
    for c, n_in, n_out, nbr_dn, nbr in levels:
        x = torch.randn(n_out, c, device="cuda"); xin = torch.randn(n_in, c // 2, device="cuda")
        w = torch.randn(27, c, c, device="cuda") * (2.0 / (27 * c)) ** 0.5
        sc = torch.rand(c, device="cuda") + 0.5; sh = torch.randn(c, device="cuda"); res = torch.randn(n_out, c, device="cuda")
        run("subm %d->%d n=%d" % (c, c, n_out), x, c, w, nbr, 27, n_out, c, False, int((nbr >= 0).sum()), scale=sc, shift=sh, residual=res, relu=True)
        if c // 2 >= 32:
            wd = torch.randn(27, c // 2, c, device="cuda") * (2.0 / (27 * c // 2)) ** 0.5
            run("down %d->%d n=%d" % (c // 2, c, n_out), xin, c // 2, wd, nbr_dn, 27, n_out, c, False, int((nbr_dn >= 0).sum()), scale=sc, shift=sh, relu=True)
