#!/usr/bin/env python3
"""Does the ROW ORDER of a sparse level matter for the sub-manifold conv kernels? Canonical (b,z,y,x) order vs tiled
orders (rows of one (ty x tx) column of the grid consecutive), same kernel, same arithmetic, permuted rulebook."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops
from cpd_amd.engine import ModelConfig
from cpd_amd.synthetic import waymo_cloud

B = int(os.environ.get("FRAMES", "16"))
math = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
reps = 10
torch.manual_seed(0)


def timeit(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def reorder(idx, nbr, key):
    n = idx.shape[0]
    perm = torch.argsort(key, stable=True)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=perm.device)
    t = nbr[:, perm].long()
    new = torch.where(t >= 0, inv[t.clamp_min(0)], t).to(torch.int32).contiguous()
    groups = (n + 15) // 16
    valid = torch.zeros((nbr.shape[0], groups * 16), dtype=torch.bool, device=nbr.device)
    valid[:, :n] = new >= 0
    anyv = valid.view(nbr.shape[0], groups, 16).any(-1)                     # [kv, groups]
    bits = (anyv.long() << torch.arange(nbr.shape[0], device=nbr.device).view(-1, 1)).sum(0)
    new.tapmask = bits.to(torch.int32).contiguous()
    return new, perm


cfg = ModelConfig()
vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, 5, cfg.max_voxels)
clouds = [torch.from_numpy(waymo_cloud(s % 8)).cuda() for s in range(B)]
_, coords, _, feats, nvox = vox.batch(clouds)
n = int(nvox[B]); coords = coords[:n]
shape = cfg.sparse_shape
index = ops.SiteIndex.build(coords, B, shape)
for lvl, (k, s, p, c) in enumerate([([3, 3, 3], [2, 2, 2], [1, 1, 1], 32), ([3, 3, 3], [2, 2, 2], [1, 1, 1], 64), ([3, 3, 3], [2, 2, 2], [0, 1, 1], 128)], 1):
    o_idx, o_index, o_shape = ops.conv_outset(coords, B, shape, k, s, p)
    nbr = ops.rulebook_subm(o_idx, o_index)
    n_out = o_idx.shape[0]
    x = torch.randn(n_out, c, device="cuda")
    w = torch.randn(27, c, c, device="cuda") * (2.0 / (27 * c)) ** 0.5
    pw = ops.pack_weight(w)
    pairs = int((nbr >= 0).sum())
    D, H, W = o_shape
    b_, z_, y_, x_ = (o_idx[:, i].long() for i in range(4))
    orders = {"canonical (b,z,y,x)": None}
    for ty, tx in [(8, 8), (16, 16), (32, 32), (64, 64)]:
        orders["tile %dx%d z-major" % (ty, tx)] = ((((b_ * ((H + ty - 1) // ty) + y_ // ty) * ((W + tx - 1) // tx) + x_ // tx) * D + z_) * ty + y_ % ty) * tx + x_ % tx
        orders["tile %dx%d y,z,x" % (ty, tx)] = ((((b_ * ((H + ty - 1) // ty) + y_ // ty) * ((W + tx - 1) // tx) + x_ // tx) * ty + y_ % ty) * D + z_) * tx + x_ % tx
    orders["(b,y,x,z) z innermost"] = ((b_ * H + y_) * W + x_) * D + z_
    if os.environ.get("ORDER_SET", "") == "mask":
        # rows of a chunk of R canonical rows sorted by their 27-bit neighbour pattern: 16-row groups with uniform tap sets
        orders = {"canonical (b,z,y,x)": None}
        mask = ((nbr >= 0).long() << torch.arange(27, device=nbr.device).view(-1, 1)).sum(0)
        rank = torch.arange(n_out, device=nbr.device)
        for R in (256, 1024, 4096, 16384, 1 << 30):
            orders["mask-sorted in chunks of %d" % R] = (rank // R) * (1 << 27) + mask
        pop = ((nbr >= 0).long()).sum(0)
        orders["popcount-sorted in chunks of 1024"] = (rank // 1024) * 32 + pop
    if os.environ.get("ORDER_SET", "") == "band":
        # round 5: (b, y-band, z, y, x) "band-major" order -- a chunk of 4096 rows is then a band of y-lines through ALL z-planes, so a
        # row's z-neighbours sit in its own chunk instead of 1.3 chunks away -- then the usual pattern sort inside 4096-row chunks
        orders = {"canonical (b,z,y,x)": None}
        mask = ((nbr >= 0).long() << torch.arange(27, device=nbr.device).view(-1, 1)).sum(0)
        rank = torch.arange(n_out, device=nbr.device)
        orders["canonical + mask chunks 4096"] = (rank // 4096) * (1 << 27) + mask
        for band in (8, 16, 32, 64):
            bkey = (((b_ * ((H + band - 1) // band) + y_ // band) * D + z_) * band + y_ % band) * W + x_
            brank = torch.empty_like(rank); brank[torch.argsort(bkey, stable=True)] = rank
            orders["band %d" % band] = bkey
            for R in (4096, 16384):
                orders["band %d + mask chunks %d" % (band, R)] = (brank // R) * (1 << 27) + mask
        orders["LOCAL (synthetic: tap t -> row + t - 13)"] = "local"
    for name, key in orders.items():
        if isinstance(key, str):
            off = (torch.arange(27, device=nbr.device).view(-1, 1) - 13)
            nb0 = torch.where(nbr >= 0, (torch.arange(n_out, device=nbr.device).view(1, -1) + off).clamp(0, n_out - 1).to(torch.int32), nbr).contiguous()
            nb, perm = reorder(o_idx, nb0, (torch.arange(n_out, device=nbr.device) // 4096) * (1 << 27) + ((nbr >= 0).long() << torch.arange(27, device=nbr.device).view(-1, 1)).sum(0))
            xin = x
        elif key is None:
            nb, xin = nbr, x
        else:
            nb, perm = reorder(o_idx, nbr, key)
            xin = x[perm].contiguous()
        exe = 0
        if nb.tapmask is not None:
            tm = nb.tapmask.long() & ((1 << 27) - 1)
            exe = sum(int(((tm >> t) & 1).sum()) for t in range(27)) * 16
        kname = ops.gather_conv_tile(n_out, c, c, c, math=math)
        res = []
        for inner in ("0", "1"):
            os.environ["CPD_TUNE"] = "1"; os.environ["CPD_GC_TAPS_INNER"] = inner
            us = timeit(lambda: ops.gather_conv(xin, c, pw, nb, 27, n_out, c, relu=True, math=math))
            res.append(us)
        print("L%d %3d ch n=%7d %-26s %-32s tap-outer %8.1f us  taps-inner %8.1f us (%6.1f TF useful)  executed/useful %.2f" %
              (lvl, c, n_out, name, kname, res[0], res[1], 2.0 * pairs * c * c / min(res) / 1e6, exe / max(pairs, 1)), flush=True)
    coords, index, shape = o_idx, o_index, o_shape
