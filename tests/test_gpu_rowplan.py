"""Round 4: brick row order (cpd_order_rows_bricks), the row plan of a sub-manifold rulebook (cpd_rulebook_plan) and the staged
row-wave kernel (cpd_gather_conv_planned) -- against numpy restatements, the CPU oracle's SubMConv3d (cpd_ref_sparse_conv:
spconv_backbone.py:108-115 semantics) and the row-wave kernel it replaces."""
import numpy as np
import pytest
import torch

from cpd_amd import ops
from cpd_amd.synthetic import WAYMO, waymo_cloud

pytestmark = pytest.mark.gpu

DOWN = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1])]


def _level(frames, depth):
    """canonical site list + canonical index of strided level `depth` (1 = stride 2) of `frames` synthetic clouds, on the device"""
    vox = ops.Voxelizer(WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 5, 1000000)
    pts = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(frames)]
    _, coords, _, _, nvox, index = vox.batch(pts, index_z_extra=1, canonical=True) if frames > 1 else (None,) * 6
    if frames == 1:
        _, coords, _, _, n = vox(pts[0], coord_cols=4)
        g = vox.grid_zyx
        shape = [g[0] + 1, g[1], g[2]]
        index = ops.SiteIndex.build(coords, 1, shape)
    else:
        n = int(nvox[frames])
        coords = coords[:n]
        shape = index.shape
    for k, s, p in DOWN[:depth]:
        coords, index, shape = ops.conv_outset(coords, frames, shape, k, s, p)
    return coords, index, shape


def _brick_key(idx, shape, by=8, bx=8):
    return np.lexsort((idx[:, 3], idx[:, 2], idx[:, 3] // bx, idx[:, 2] // by, idx[:, 1], idx[:, 0]))


@pytest.mark.parametrize("tile", [128, 256])
def test_brick_order_is_the_brick_sort_with_pattern_sorted_tiles(oracle, hip, tile):
    coords, index, shape = _level(2, 2)
    n = coords.shape[0]
    out, n2o, o2n = ops.order_rows_bricks(coords, index, tile_rows=tile)
    c, out, n2o, o2n = coords.cpu().numpy(), out.cpu().numpy(), n2o.cpu().numpy(), o2n.cpu().numpy()
    np.testing.assert_array_equal(np.sort(n2o), np.arange(n))                 # a permutation
    np.testing.assert_array_equal(o2n[n2o], np.arange(n))                     # and its inverse
    np.testing.assert_array_equal(out, c[n2o])
    want = _brick_key(c, shape)                                               # brick position -> canonical row
    nbr = np.asarray(oracle.subm_rulebook(c, 2, shape, [3, 3, 3]))
    pat = np.zeros(n, np.int64)
    for t in range(27):
        pat |= (nbr[t] >= 0).astype(np.int64) << t
    for t0 in range(0, n, tile):
        rows = n2o[t0:t0 + tile]
        mine = want[t0:t0 + tile]
        # the tile holds exactly the rows of brick positions t0 .. t0 + tile - 1, stably sorted by neighbour pattern
        expect = mine[np.argsort(pat[mine], kind="stable")]
        np.testing.assert_array_equal(rows, expect)


@pytest.mark.parametrize("T", [128, 256])
def test_row_plan_lists_the_distinct_inputs_of_every_tile_and_group(hip, T):
    coords, index, shape = _level(1, 1)
    out, _, o2n = ops.order_rows_bricks(coords, index, tile_rows=T)
    index.set_order(o2n)
    nbr = ops.rulebook_subm(out, index)
    ops.rulebook_plan(nbr, T)
    n = out.shape[0]
    tiles = (n + T - 1) // T
    tab = nbr.cpu().numpy()
    slots = nbr.plan[0].cpu().numpy().view(np.uint16).reshape(tiles, 27, T)
    ulist = nbr.plan[1].cpu().numpy().reshape(tiles, 3, 9 * T)
    count = nbr.plan[2].cpu().numpy().reshape(tiles, 4)
    pad = np.full((27, tiles * T), -1, np.int32)
    pad[:, :n] = tab
    longest = 0
    for t in range(tiles):
        for g in range(3):
            blk = pad[9 * g:9 * g + 9, t * T:(t + 1) * T]
            u = np.unique(blk[blk >= 0])
            assert count[t, g] == u.size
            np.testing.assert_array_equal(np.sort(ulist[t, g, :u.size]), u)           # each distinct row once (list order is arrival order)
            sl = slots[t, 9 * g:9 * g + 9]
            assert ((sl == 0xffff) == (blk < 0)).all()
            np.testing.assert_array_equal(ulist[t, g][sl[blk >= 0]], blk[blk >= 0])
            longest = max(longest, u.size)
        assert count[t, 3] == count[t, :3].sum()
    assert longest <= {128: 224, 256: 384}[T], longest    # brick order keeps every group inside the kernel's one-pass window


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("c", [32, 64, 128])
@pytest.mark.parametrize("order", ["bricks", "canonical"])
def test_planned_conv_matches_oracle_and_the_rowwave_kernel(oracle, hip, c, order, tile):
    """SubM c -> c on fp16-pair rows, with the BatchNorm / residual / ReLU epilogue the SparseBasicBlocks use, pair output: the staged
    kernel (asserted through the launch log) vs the oracle on the same fp32 values (<= 1e-4) and vs the row-wave kernel on the same
    table (identical partial products; for c = 32 the same accumulation order: bit-identical). `canonical`: the plan makes no
    assumption on the row order -- lists are longer there (a group can exceed the window: the multi-pass path)."""
    coords, index, shape = _level(2, {32: 1, 64: 2, 128: 2}[c])
    if order == "bricks":
        idx, _, o2n = ops.order_rows_bricks(coords, index, tile_rows=tile)
        index.set_order(o2n)
    else:
        idx = coords
    n = idx.shape[0]
    assert n >= 512 * 128 or c == 128
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, generator=g).cuda() * 2.0
    x = torch.relu(x) + 0.01 * x                                            # post-ReLU-like, both signs
    w = (torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).cuda()
    scale = (torch.rand(c, generator=g) + 0.5).cuda()
    shift = (torch.randn(c, generator=g) * 0.1).cuda()
    res = torch.randn(n, c, generator=g).cuda()
    pw = ops.pack_weight(w)
    xp, rp = ops.rows_to_pairs(x), ops.rows_to_pairs(res)
    nbr = ops.rulebook_subm(idx, index)
    import os
    os.environ["CPD_TUNE"], os.environ["CPD_GC_PLANNED_MIN"] = "1", "1"       # (the 128-channel level of two frames is below 512 tiles)
    try:
        base = ops.gather_conv(xp, c, pw, nbr, 27, n, c, scale, shift, rp, True, math="f16x2", in_pairs=True, out_pairs=True, res_pairs=True)
        ops.rulebook_plan(nbr, tile)
        with ops.launch_log() as log:
            got = ops.gather_conv(xp, c, pw, nbr, 27, n, c, scale, shift, rp, True, math="f16x2", in_pairs=True, out_pairs=True, res_pairs=True)
    finally:
        del os.environ["CPD_TUNE"], os.environ["CPD_GC_PLANNED_MIN"]
    assert log.counts == {"rowplan_conv_f16p_kernel<%d%s>" % (c, ",256" if tile == 256 else ""): 1}, log.counts
    got, base = ops.pairs_to_rows(got), ops.pairs_to_rows(base)
    # c = 32 in brick order (one window pass per group): the row-wave kernel's products in its accumulation order -- what is left is
    # the epilogue's scale / shift / residual arithmetic, contracted differently by the compiler in the two kernels (1 ulp)
    np.testing.assert_allclose(got.cpu().numpy(), base.cpu().numpy(), atol=4e-6 if (c == 32 and order == "bricks") else 2e-5, rtol=0)
    # oracle: SubM conv on the exact fp32 values the pair rows hold, then the same epilogue
    xe, re_ = ops.pairs_to_rows(xp).cpu().numpy(), ops.pairs_to_rows(rp).cpu().numpy()
    w_ref = w.cpu().numpy().reshape(3, 3, 3, c, c).transpose(4, 0, 1, 2, 3).copy()               # (Cout, kD, kH, kW, Cin)
    y = oracle.sparse_conv(xe, w_ref, None, nbr.cpu().numpy())
    want = np.maximum(y * scale.cpu().numpy() + shift.cpu().numpy() + re_, 0.0)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4 * max(1.0, float(np.abs(want).max())), rtol=0)
