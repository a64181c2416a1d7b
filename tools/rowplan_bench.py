#!/usr/bin/env python3
"""Micro-benchmark of the SubM convs of levels 2-4 (32 / 64 / 128 channels) on FRAMES synthetic frames: the row-wave kernel on
tap-pattern rows (round 3), the row-wave kernel on brick rows, and the staged row-wave kernel (planned rulebook) on brick rows.
CPD_HIP_LIB selects a diagnostic library (tools/build_ablate.sh).   FRAMES=16 python tools/rowplan_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops
from cpd_amd.synthetic import WAYMO, waymo_cloud

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(os.environ.get("FRAMES", "16"))
TILE = int(os.environ.get("TILE", "256"))
DOWN = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1])]


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


vox = ops.Voxelizer(WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 5, 1000000)
pts = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(B)]
_, coords, _, _, nvox, index = vox.batch(pts, index_z_extra=1, canonical=True)
coords = coords[:int(nvox[B])]
shape = index.shape
for lvl, (k, s, p) in enumerate(DOWN):
    coords, index, shape = ops.conv_outset(coords, B, shape, k, s, p)
    c = [32, 64, 128][lvl]
    n = coords.shape[0]
    g = torch.Generator().manual_seed(c)
    w = ops.pack_weight((torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).cuda())
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()
    x = torch.relu(torch.randn(n, c, generator=g)).cuda()
    res = ops.rows_to_pairs(torch.randn(n, c, generator=g).cuda())
    xp = ops.rows_to_pairs(x)
    out = torch.empty_like(xp)
    line = "L%d %3d ch %8d rows:" % (lvl + 2, c, n)
    for order in ("taps", "bricks", "bricks+plan"):
        index.set_order(None)
        if order == "taps":
            idx, _, o2n = ops.order_rows_by_taps(coords, index)
        else:
            idx, _, o2n = ops.order_rows_bricks(coords, index, tile_rows=TILE)
        index.set_order(o2n)
        nbr = ops.rulebook_subm(idx, index)
        pairs = int((nbr >= 0).sum())
        if order == "bricks+plan":
            us_plan = timeit(lambda: ops.rulebook_plan(nbr, TILE))
        name = ops.gather_conv_tile(n, c, c, c, nbr=nbr, math="f16x2", in_pairs=True)
        us = timeit(lambda: ops.gather_conv(xp, c, w, nbr, 27, n, c, scale, shift, res, True, out=out, math="f16x2", in_pairs=True, out_pairs=True, res_pairs=True))
        line += "  %s %s %.0f us (%.0f TF)" % (order, name.replace("_conv_f16p_kernel", ""), us, 2.0 * pairs * c * c / us / 1e6)
    print(line + "  plan build %.0f us" % us_plan, flush=True)
