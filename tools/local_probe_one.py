#!/usr/bin/env python3
"""One level of tools/order_probe.py's ORDER_SET=band experiment for the counter passes of tools/local_pmc.sh: the 32 / 64 / 128-channel
level (LEVEL=1|2|3) in pattern-sorted 4096-row chunks, with its REAL rulebook or (VARIANT=local) a synthetic one of perfect locality
(tap t -> row + t - 13, same tap masks) -- five launches of the row-wave kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops
from cpd_amd.engine import ModelConfig
from cpd_amd.synthetic import waymo_cloud
B = int(os.environ.get("FRAMES", "48")); LEVEL = int(os.environ.get("LEVEL", "1")); local = os.environ.get("VARIANT", "real") == "local"
cfg = ModelConfig()
vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, 5, cfg.max_voxels)
clouds = [torch.from_numpy(waymo_cloud(s % 8)).cuda() for s in range(B)]
_, coords, _, feats, nvox = vox.batch(clouds)
coords = coords[:int(nvox[B])]; shape = cfg.sparse_shape
index = ops.SiteIndex.build(coords, B, shape)
for lvl, (k, s, p, c) in enumerate([([3, 3, 3], [2, 2, 2], [1, 1, 1], 32), ([3, 3, 3], [2, 2, 2], [1, 1, 1], 64), ([3, 3, 3], [2, 2, 2], [0, 1, 1], 128)], 1):
    o_idx, o_index, o_shape = ops.conv_outset(coords, B, shape, k, s, p)
    if lvl == LEVEL:
        break
    coords, index, shape = o_idx, o_index, o_shape
nbr = ops.rulebook_subm(o_idx, o_index)
n = o_idx.shape[0]
ar = torch.arange(n, device="cuda")
if local:
    off = torch.arange(27, device="cuda").view(-1, 1) - 13
    nbr = torch.where(nbr >= 0, (ar.view(1, -1) + off).clamp(0, n - 1).to(torch.int32), nbr).contiguous()
mask = ((nbr >= 0).long() << torch.arange(27, device="cuda").view(-1, 1)).sum(0)
perm = torch.argsort((ar // 4096) * (1 << 27) + mask, stable=True)
inv = torch.empty_like(perm); inv[perm] = ar
t = nbr[:, perm].long()
new = torch.where(t >= 0, inv[t.clamp_min(0)], t).to(torch.int32).contiguous()
groups = (n + 15) // 16
valid = torch.zeros((27, groups * 16), dtype=torch.bool, device="cuda"); valid[:, :n] = new >= 0
new.tapmask = (valid.view(27, groups, 16).any(-1).long() << torch.arange(27, device="cuda").view(-1, 1)).sum(0).to(torch.int32).contiguous()
x = torch.randn(n, c, device="cuda")
pw = ops.pack_weight(torch.randn(27, c, c, device="cuda") * (2.0 / (27 * c)) ** 0.5)
for _ in range(5):
    ops.gather_conv(x, c, pw, new, 27, n, c, relu=True, math="f16x2")
torch.cuda.synchronize()
print("level", LEVEL, "rows", n, "variant", "local" if local else "real")
