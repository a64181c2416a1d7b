#!/usr/bin/env python3
"""Row-wave SubM convs of levels 2-4 (32 / 64 / 128 channels, fp16-pair rows, tap-pattern row order) on FRAMES synthetic frames:
128-row / 4-wave workgroups against the wide variant (CPD_GC_RW8: 256 rows / 8 waves; 192 / 6 at 128 columns); outputs compared
bit for bit.   FRAMES=16 python tools/rw_wide_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CPD_TUNE"] = "1"
import torch
from cpd_amd import ops
from cpd_amd.synthetic import WAYMO, waymo_cloud

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(os.environ.get("FRAMES", "16"))
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
    xp = ops.rows_to_pairs(torch.relu(torch.randn(n, c, generator=g)).cuda())
    res = ops.rows_to_pairs(torch.randn(n, c, generator=g).cuda())
    idx, _, o2n = ops.order_rows_by_taps(coords, index)
    index.set_order(o2n)
    nbr = ops.rulebook_subm(idx, index)
    pairs = int((nbr >= 0).sum())
    line = "L%d %3d ch %8d rows:" % (lvl + 2, c, n)
    outs = []
    for widths in (0, 32 + 64 + 128):
        os.environ["CPD_GC_RW8"] = str(widths)
        os.environ["CPD_GC_RW8_MIN"] = "1"
        out = torch.empty_like(xp)
        name = ops.gather_conv_tile(n, c, c, c, nbr=nbr, math="f16x2", in_pairs=True)
        us = timeit(lambda: ops.gather_conv(xp, c, w, nbr, 27, n, c, scale, shift, res, True, out=out, math="f16x2", in_pairs=True, out_pairs=True, res_pairs=True))
        outs.append(out)
        line += "  %s %.0f us (%.0f TF)" % (name.replace("rowwave_conv_", ""), us, 2.0 * pairs * c * c / us / 1e6)
    print(line + "  equal=%s" % bool(torch.equal(outs[0], outs[1])), flush=True)
