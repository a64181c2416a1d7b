#!/usr/bin/env python3
"""Config C5: rulebook build + gather-scatter on a dense-object stress cloud
(Waymo-shape, voxel (0.05, 0.05, 0.1) -> sparse [61, 3008, 3008]); algorithmic GB/s vs the HBM roof.
Algorithmic bytes: voxelize 4*N*C + 4*M*(C+5); rulebook 16*N_in + 4*27*N_out (table) ;
sparse conv layer 4*(N_in*C_in + N_out*C_out) + 4*27*C_in*C_out + 4*27*N_out (table read)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops  # noqa: E402
from cpd_amd.synthetic import WAYMO, waymo_cloud  # noqa: E402


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for n_points in (160000, 1000000):
    pts = torch.from_numpy(waymo_cloud(3, n_points=n_points, n_az=2650 if n_points <= 160000 else 18000)).cuda()
    vs = [0.05, 0.05, 0.1]
    vz = ops.Voxelizer(vs, WAYMO["point_cloud_range"], 5, 5, 1000000)
    shape = [61, 3008, 3008]
    _, c, n, mean, m = vz(pts, coord_cols=4, want_voxels=False)
    t_vox = timeit(lambda: vz(pts, coord_cols=4, want_voxels=False, sync=False))
    t_idx = timeit(lambda: ops.SiteIndex.build(c, 1, shape))
    index = ops.SiteIndex.build(c, 1, shape)
    t_rb = timeit(lambda: ops.rulebook_subm(c, index))
    nbr = ops.rulebook_subm(c, index)
    pairs = int((nbr >= 0).sum())
    out = {"n_points": n_points, "n_active": m, "subm_pairs": pairs}
    feats = {}
    for cch in (16, 32, 64):
        x = torch.randn((m, cch), device="cuda")
        w = ops.pack_weight(torch.randn((27, cch, cch), device="cuda") * 0.05)
        t = timeit(lambda: ops.gather_conv(x, cch, w, nbr, 27, m, cch))
        byts = 4 * (m * cch * 2) + 4 * 27 * cch * cch + 4 * 27 * m
        feats["subm%d" % cch] = {"us": round(t * 1e6, 1), "alg_GBps": round(byts / t / 1e9, 1), "useful_TFLOPs": round(2.0 * pairs * cch * cch / t / 1e12, 2)}
    out.update({
        "voxelize": {"us": round(t_vox * 1e6, 1), "alg_GBps": round((4 * n_points * 5 + 4 * m * 10) / t_vox / 1e9, 1)},
        "index_build": {"us": round(t_idx * 1e6, 1), "bitmap_MB": round(61 * 3008 * 3008 / 8 / 1e6, 1)},
        "rulebook_subm": {"us": round(t_rb * 1e6, 1), "alg_GBps": round((16 * m + 4 * 27 * m) / t_rb / 1e9, 1)},
        "layers": feats})
    print(json.dumps(out), flush=True)
