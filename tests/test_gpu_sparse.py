"""B2 parity on the GPU: site index, SubM / regular rulebooks (bit-exact incl. canonical order),
sparse conv through the fp32-MFMA gather kernel (<= 1e-4), densify (bit-exact)."""
import numpy as np
import pytest
import torch

from cpd_amd import ops
from cpd_amd.synthetic import KITTI, WAYMO, kitti_cloud, waymo_cloud

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype else t


def random_sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def voxel_coords(oracle, cfg, pts):
    _, c, _ = oracle.voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 5, cfg["max_voxels"])
    return np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)


def sparse_shape(cfg):
    g = ops.voxel_grid_size(cfg["voxel_size"], cfg["point_cloud_range"])
    return [g[0] + 1, g[1], g[2]]


@pytest.mark.parametrize("batch,shape,n", [(2, [7, 33, 65], 3000), (1, [41, 200, 176], 20000), (3, [5, 64, 64], 61440)])
def test_subm_rulebook_bit_exact(oracle, hip, batch, shape, n):
    rng = np.random.default_rng(n)
    idx = random_sites(rng, batch, shape, n)
    index = ops.SiteIndex.build(dev(idx), batch, shape)
    nbr = ops.rulebook_subm(dev(idx), index).cpu().numpy()
    np.testing.assert_array_equal(nbr, oracle.subm_rulebook(idx, batch, shape, [3, 3, 3]))


@pytest.mark.parametrize("ksize,stride,pad", [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
                                              ([3, 1, 1], [2, 1, 1], [0, 0, 0])])
def test_regular_conv_rulebook_bit_exact(oracle, hip, ksize, stride, pad):
    rng = np.random.default_rng(11)
    batch, shape = 2, [11, 40, 52]
    idx = random_sites(rng, batch, shape, 6000)
    out_idx, out_index, out_shape = ops.conv_outset(dev(idx), batch, shape, ksize, stride, pad)
    want_idx = oracle.conv_outset(idx, batch, shape, ksize, stride, pad)
    np.testing.assert_array_equal(out_idx.cpu().numpy(), want_idx)        # canonical (b,z,y,x) order
    assert out_shape == oracle.conv_out_shape(shape, ksize, stride, pad)
    index = ops.SiteIndex.build(dev(idx), batch, shape)
    nbr = ops.rulebook_conv(out_idx, index, ksize, stride, pad).cpu().numpy()
    np.testing.assert_array_equal(nbr, oracle.conv_rulebook(idx, want_idx, batch, shape, ksize, stride, pad))
    # SubM on the canonical output list (perm unused path)
    nbr2 = ops.rulebook_subm(out_idx, out_index).cpu().numpy()
    np.testing.assert_array_equal(nbr2, oracle.subm_rulebook(want_idx, batch, out_shape, [3, 3, 3]))


def test_waymo_level_chain_bit_exact(oracle, hip):
    """Config W: the whole indice chain L0 -> L4 of VoxelResBackBone8x on the 160k cloud."""
    idx = voxel_coords(oracle, WAYMO, waymo_cloud(0))
    shape = sparse_shape(WAYMO)
    assert shape == [41, 1504, 1504]
    d_idx = dev(idx)
    index = ops.SiteIndex.build(d_idx, 1, shape)
    np.testing.assert_array_equal(ops.rulebook_subm(d_idx, index).cpu().numpy(),
                                  oracle.subm_rulebook(idx, 1, shape, [3, 3, 3]))
    cur, cur_d, cur_index = idx, d_idx, index
    for k, s, p in [([3, 3, 3], [2, 2, 2], [1, 1, 1])] * 2 + [([3, 3, 3], [2, 2, 2], [0, 1, 1]), ([3, 1, 1], [2, 1, 1], [0, 0, 0])]:
        o_d, o_index, o_shape = ops.conv_outset(cur_d, 1, shape, k, s, p)
        want = oracle.conv_outset(cur, 1, shape, k, s, p)
        np.testing.assert_array_equal(o_d.cpu().numpy(), want)
        np.testing.assert_array_equal(ops.rulebook_conv(o_d, cur_index, k, s, p).cpu().numpy(),
                                      oracle.conv_rulebook(cur, want, 1, shape, k, s, p))
        cur, cur_d, cur_index, shape = want, o_d, o_index, o_shape
    assert shape == [2, 188, 188]


def run_conv(feat, w, nbr, scale=None, shift=None, residual=None, relu=False):
    cout, cin = w.shape[0], w.shape[-1]
    w_kio = torch.from_numpy(w).reshape(cout, -1, cin).permute(1, 2, 0).contiguous().cuda()
    packed = ops.pack_weight(w_kio)
    out = ops.gather_conv(dev(feat), cin, packed, dev(nbr), nbr.shape[0], nbr.shape[1], cout,
                          dev(scale) if scale is not None else None, dev(shift) if shift is not None else None,
                          dev(residual) if residual is not None else None, relu)
    return out.cpu().numpy()


@pytest.mark.parametrize("cin,cout", [(5, 16), (4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
def test_subm_conv_matches_oracle(oracle, hip, cin, cout):
    rng = np.random.default_rng(cin * 1000 + cout)
    batch, shape = 2, [9, 40, 40]
    idx = random_sites(rng, batch, shape, 5000)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32) * 0.1
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    raw = oracle.sparse_conv(feat, w, bias, nbr)
    # plain conv + bias (the un-fused module path)
    np.testing.assert_allclose(run_conv(feat, w, nbr, None, bias), raw, atol=1e-4, rtol=0)
    # fused BN-affine + residual + ReLU (SparseBasicBlock tail); residual needs cin == cout
    res = feat if cin == cout else None
    want = oracle.affine_rows(oracle.sparse_conv(feat, w, None, nbr), scale, bias, res, True)
    np.testing.assert_allclose(run_conv(feat, w, nbr, scale, bias, res, True), want, atol=1e-4, rtol=0)


@pytest.mark.parametrize("ms,nt", [(1, 1), (1, 4), (2, 2), (2, 4), (2, 8), (4, 1), (4, 4), (4, 8), (1, 8)])
def test_every_tile_shape_gives_the_same_conv(oracle, hip, ms, nt, monkeypatch):
    rng = np.random.default_rng(77)
    batch, shape, cin, cout = 1, [6, 30, 30], 32, 128
    idx = random_sites(rng, batch, shape, 1500)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    monkeypatch.setenv("CPD_TUNE", "1")          # the knobs below are only read when this is set
    monkeypatch.setenv("CPD_GC_MS", str(ms)); monkeypatch.setenv("CPD_GC_NT", str(nt))
    np.testing.assert_allclose(run_conv(feat, w, nbr), oracle.sparse_conv(feat, w, None, nbr), atol=1e-4, rtol=0)


def test_regular_conv_and_densify(oracle, hip):
    rng = np.random.default_rng(21)
    batch, shape, cin, cout = 2, [5, 24, 28], 64, 128
    idx = random_sites(rng, batch, shape, 2500)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    k, s, p = [3, 1, 1], [2, 1, 1], [0, 0, 0]
    w = (rng.normal(size=[cout] + k + [cin]) * np.sqrt(2.0 / (3 * cin))).astype(np.float32)
    out_idx = oracle.conv_outset(idx, batch, shape, k, s, p)
    oshape = oracle.conv_out_shape(shape, k, s, p)
    nbr = oracle.conv_rulebook(idx, out_idx, batch, shape, k, s, p)
    want = oracle.sparse_conv(feat, w, None, nbr)
    got = run_conv(feat, w, nbr)
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=0)
    # dense(): reference layout bit-exact, channels-last layout = the same numbers permuted
    nchw = ops.densify_nchw(dev(want), dev(out_idx), batch, oshape).cpu().numpy()
    ref = oracle.densify(want, out_idx, batch, oshape)
    np.testing.assert_array_equal(nchw, ref)
    nhwc = ops.densify_nhwc(dev(want), dev(out_idx), batch, oshape).cpu().numpy()    # (B,H,W,D*C), ch = z*C + c
    D, C = oshape[0], cout
    np.testing.assert_array_equal(nhwc.reshape(batch, oshape[1], oshape[2], D, C).transpose(0, 4, 3, 1, 2)
                                  .reshape(batch, C * D, oshape[1], oshape[2]), ref)


def test_kitti_c4_subm_path(oracle, hip):
    """Config C4: 20k KITTI cloud at 0.05 m -> high-sparsity SubM layers 4->16->16."""
    idx = voxel_coords(oracle, KITTI, kitti_cloud(0))
    shape = sparse_shape(KITTI)
    assert shape == [41, 1600, 1408]
    rng = np.random.default_rng(4)
    feat = rng.normal(size=(idx.shape[0], 4)).astype(np.float32)
    index = ops.SiteIndex.build(dev(idx), 1, shape)
    nbr = ops.rulebook_subm(dev(idx), index).cpu().numpy()
    want_nbr = oracle.subm_rulebook(idx, 1, shape, [3, 3, 3])
    np.testing.assert_array_equal(nbr, want_nbr)
    w1 = (rng.normal(size=(16, 3, 3, 3, 4)) * 0.2).astype(np.float32)
    w2 = (rng.normal(size=(16, 3, 3, 3, 16)) * 0.1).astype(np.float32)
    x = run_conv(feat, w1, nbr, relu=True)
    y = run_conv(x, w2, nbr, relu=True)
    xr = np.maximum(oracle.sparse_conv(feat, w1, None, want_nbr), 0)
    yr = np.maximum(oracle.sparse_conv(xr, w2, None, want_nbr), 0)
    np.testing.assert_allclose(x, xr, atol=1e-4); np.testing.assert_allclose(y, yr, atol=1e-4)


@pytest.mark.parametrize("cin,cout,n", [(16, 16, 5000), (32, 64, 3000), (128, 128, 2000), (64, 64, 40000)])
def test_tapmask_skipping_is_exact(oracle, hip, cin, cout, n):
    """Device rulebook + tap masks (empty (row group, tap) pairs skipped) == oracle, on clustered
    sites so that many pairs really are empty; regular conv path as well."""
    rng = np.random.default_rng(n)
    batch, shape = 2, [9, 64, 64]
    idx = random_sites(rng, batch, [9, 24, 24], n // 4 if n > 20000 else n // 8)      # dense blob
    far = random_sites(rng, batch, shape, n // 6)                                      # plus scattered sites
    idx = np.unique(np.concatenate([idx, far]), axis=0).astype(np.int32)
    rng.shuffle(idx)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    d_idx = dev(idx)
    index = ops.SiteIndex.build(d_idx, batch, shape)
    nbr = ops.rulebook_subm(d_idx, index)
    want_nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    np.testing.assert_array_equal(nbr.cpu().numpy(), want_nbr)
    tm = nbr.tapmask.cpu().numpy().view(np.uint32)
    valid = np.pad(want_nbr >= 0, ((0, 0), (0, (-idx.shape[0]) % 16))).reshape(27, -1, 16).any(2)     # [27, n_sub]
    want_tm = (valid.astype(np.uint64) << np.arange(27, dtype=np.uint64)[:, None]).sum(0).astype(np.uint32)
    np.testing.assert_array_equal(tm, want_tm)
    w_kio = torch.from_numpy(w).reshape(cout, -1, cin).permute(1, 2, 0).contiguous().cuda()
    got = ops.gather_conv(dev(feat), cin, ops.pack_weight(w_kio), nbr, 27, idx.shape[0], cout).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sparse_conv(feat, w, None, want_nbr), atol=1e-4, rtol=0)
    # strided conv with its own masks
    k, s, p = [3, 3, 3], [2, 2, 2], [1, 1, 1]
    o_d, o_index, o_shape = ops.conv_outset(d_idx, batch, shape, k, s, p)
    nbr2 = ops.rulebook_conv(o_d, index, k, s, p)
    want2 = oracle.conv_rulebook(idx, o_d.cpu().numpy(), batch, shape, k, s, p)
    np.testing.assert_array_equal(nbr2.cpu().numpy(), want2)
    got2 = ops.gather_conv(dev(feat), cin, ops.pack_weight(w_kio), nbr2, 27, o_d.shape[0], cout).cpu().numpy()
    np.testing.assert_allclose(got2, oracle.sparse_conv(feat, w, None, want2), atol=1e-4, rtol=0)


@pytest.mark.parametrize("cin,cout,n,want", [(128, 128, 80000, "<128,2>"), (128, 128, 24000, "<128,1>"), (64, 64, 30000, "<64,1>"),
                                             (32, 32, 30000, "<32,1>"), (32, 32, 80000, "<32,2>"), (64, 128, 12000, "<64,1>"), (64, 64, 80000, "<64,2>")])
@pytest.mark.parametrize("math", ["bf16x3", "f16x2"])
def test_rowwave_split_bf16_variants_match_oracle(oracle, hip, math, cin, cout, n, want):
    """The sparse split kernel (both split arithmetics) in each of its shapes -- 128-row workgroups, the 64-row ones small layers get, and the
    narrower column tiles below that -- against the oracle, on clustered sites (tap skipping active) with BN/residual/ReLU."""
    rng = np.random.default_rng(n + cin)
    batch, shape = 2, [11, 96, 96]
    idx = np.unique(np.concatenate([random_sites(rng, batch, [11, 40, 40], n // 2), random_sites(rng, batch, shape, n // 2)]), axis=0)
    idx = idx.astype(np.int32)
    rows = idx.shape[0]
    feat = rng.normal(size=(rows, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    scale = (rng.random(cout) + 0.5).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(rows, cout)).astype(np.float32)
    d_idx = dev(idx)
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    name = ops.gather_conv_tile(rows, cin, cout, cin, dense=False, math=math)
    assert name == "rowwave_conv_%s_kernel" % ("f16" if math == "f16x2" else "bf16") + want, (name, rows)
    w_kio = torch.from_numpy(w).reshape(cout, -1, cin).permute(1, 2, 0).contiguous().cuda()
    got = ops.gather_conv(dev(feat), cin, ops.pack_weight(w_kio), nbr, 27, rows, cout, dev(scale), dev(shift), dev(res), True,
                          math=math).cpu().numpy()
    ref = oracle.sparse_conv(feat, w, None, nbr.cpu().numpy())
    ref = np.maximum(ref * scale + shift + res, 0)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n,want", [(128, 128, 80000, "<128,2>"), (128, 128, 24000, "<128,1>"), (64, 64, 30000, "<64,1>"), (32, 32, 30000, "<32,1>"),
                                             (32, 32, 80000, "<32,2>"), (64, 128, 12000, "<64,1>"), (64, 64, 80000, "<64,2>"), (64, 128, 600, "<32,1>"),
                                             (16, 32, 20000, "h16"), (16, 16, 30000, "h16"), (16, 16, 900, "h16"), (5, 16, 20000, None)])
def test_fp16_pair_rows_give_the_fp32_rows_result(oracle, hip, cin, cout, n, want):
    """CPD_GC_IN/OUT/RES_PAIRS: activations stored as fp16-pair rows (the f16x2 split made once, by the producing epilogue). The
    row-wave kernel on pair input, with a pair residual and pair output, in each of its shapes -- and the fp32 wave kernel writing
    pairs (the 16 -> 32 layer that feeds level 2) -- against the oracle (1e-4) and against the same call on fp32 rows (the same
    partial products; the pair kernel sums a 32-channel block in natural channel order, the fp32-row kernel in its gather order:
    fp32 rounding of the accumulation, 2e-5 here)."""
    import torch
    from cpd_amd import ops
    rng = np.random.default_rng(n + cin)
    batch, shape = 2, [11, 96, 96]
    idx = np.unique(np.concatenate([random_sites(rng, batch, [11, 40, 40], n // 2), random_sites(rng, batch, shape, n // 2)]), axis=0).astype(np.int32)
    rows = idx.shape[0]
    feat = rng.normal(size=(rows, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    scale = (rng.random(cout) + 0.5).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(rows, cout)).astype(np.float32)
    d_idx = dev(idx)
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    w_kio = torch.from_numpy(w).reshape(cout, -1, cin).permute(1, 2, 0).contiguous().cuda()
    packed = ops.pack_weight(w_kio)
    in_pairs = cin % 32 == 0 or cin == 16          # (16 channels: the wave kernel's K = 16 form on 16-channel pair rows)
    if in_pairs and cin != 16:
        name = ops.gather_conv_tile(rows, cin, cout, cin, dense=False, math="f16x2", in_pairs=True)
        assert name == "rowwave_conv_f16p_kernel" + want, (name, rows)
    x, r = dev(feat), dev(res)
    xp, rp = (ops.rows_to_pairs(x) if in_pairs else x), ops.rows_to_pairs(r)
    np.testing.assert_array_equal(ops.pairs_to_rows(rp).cpu().numpy(), (r.half().float() + (r - r.half().float()).half().float()).cpu().numpy())
    with ops.launch_log() as log:
        got_p = ops.gather_conv(xp, cin, packed, nbr, 27, rows, cout, dev(scale), dev(shift), rp, True, math="f16x2",
                                in_pairs=in_pairs, out_pairs=True, res_pairs=True)
    if cin == 16:
        assert len(log.counts) == 1 and next(iter(log.counts)).startswith("gather_conv_h16_kernel<"), log.counts
    elif in_pairs:
        convs = {k: v for k, v in log.counts.items() if k != "split_finish_kernel"}      # (small launches split their taps: two launches)
        # (f16pe: the same body with its epilogue through LDS -- pair rows out, in place, no tap split: GcParams::epi_lds)
        assert convs in ({"rowwave_conv_f16p_kernel" + want: 1}, {"rowwave_conv_f16pe_kernel" + want: 1}), log.counts
    got = ops.pairs_to_rows(got_p).cpu().numpy()
    ref = oracle.sparse_conv(feat, w, None, nbr.cpu().numpy())
    ref = np.maximum(ref * scale + shift + res, 0)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    plain = ops.gather_conv(x, cin, packed, nbr, 27, rows, cout, dev(scale), dev(shift), r, True, math="f16x2")
    np.testing.assert_allclose(got, plain.cpu().numpy(), atol=5e-5, rtol=0)
    if in_pairs:
        # pair input with fp32 output and no residual
        a = ops.gather_conv(xp, cin, packed, nbr, 27, rows, cout, dev(scale), dev(shift), None, True, math="f16x2", in_pairs=True)
        b = ops.gather_conv(x, cin, packed, nbr, 27, rows, cout, dev(scale), dev(shift), None, True, math="f16x2")
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=5e-5, rtol=0)
        # what the flags refuse: pair input into the dense path / with an absmax block / without f16x2
        from cpd_amd._lib import CpdHipError
        for kw in (dict(dense=True, math="f16x2"), dict(math="bf16x3"), dict(math="f16x2", in_absmax=ops.absmax_rows(x) if cin != 16 else ops.absmax_blocks(1, x.device)[0])):
            with pytest.raises(CpdHipError):
                ops.gather_conv(xp, cin, packed, nbr, 27, rows, cout, in_pairs=True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n,math,pairs,parts", [(128, 128, 20000, "f16x2", False, None), (128, 128, 20000, "f16x2", True, None),
                                                         (64, 64, 30000, "f16x2", True, 3), (32, 64, 9000, "f16x2", False, 2),
                                                         (64, 128, 12000, "bf16x3", False, None), (128, 128, 5000, "f16x2", False, 8)])
def test_tap_split_of_small_sparse_launches(oracle, hip, monkeypatch, cin, cout, n, math, pairs, parts):
    """cpd_gather_conv_ws: a sparse launch with fewer row-wave workgroups than ~2 per CU (one frame, the train step) deals the taps
    of a row tile to several workgroups (partial sums in a workspace) and finishes in a second launch -- parts added in a fixed
    order, then the shared epilogue's arithmetic. Chosen automatically (parts = None) or forced; fp32 and fp16-pair rows, residual,
    BN, ReLU, the absmax guard (pre-scaled input), an output row map; equal to the unsplit launch to fp32 summation order and to
    the oracle to 1e-4; the same bits on every run."""
    import torch
    from cpd_amd import ops
    monkeypatch.setenv("CPD_TUNE", "1")
    rng = np.random.default_rng(n + cin)
    batch, shape = 2, [11, 96, 96]
    idx = np.unique(np.concatenate([random_sites(rng, batch, [11, 40, 40], n // 2), random_sites(rng, batch, shape, n // 2)]), axis=0).astype(np.int32)
    rows = idx.shape[0]
    feat = rng.normal(size=(rows, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    scale = (rng.random(cout) + 0.5).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    res = rng.normal(size=(rows, cout)).astype(np.float32)
    d_idx = dev(idx)
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    packed = ops.pack_weight(torch.from_numpy(w).reshape(cout, -1, cin).permute(1, 2, 0).contiguous().cuda())
    x, r = dev(feat), dev(res)
    perm = torch.randperm(rows, device="cuda").to(torch.int32)
    kw = dict(in_pairs=True, out_pairs=True, res_pairs=True) if pairs else {}
    xin, rin = (ops.rows_to_pairs(x), ops.rows_to_pairs(r)) if pairs else (x, r)
    ref = np.maximum(oracle.sparse_conv(feat, w, None, nbr.cpu().numpy()) * scale + shift + res, 0)

    def run(split, **more):
        if split is None:
            monkeypatch.delenv("CPD_GC_SPLIT", raising=False)
        else:
            monkeypatch.setenv("CPD_GC_SPLIT", str(split))
        with ops.launch_log() as log:
            y = ops.gather_conv(xin, cin, packed, nbr, 27, rows, cout, dev(scale), dev(shift), rin, True, math=math, **kw, **more)
        return (ops.pairs_to_rows(y) if pairs else y), log.counts

    whole, log1 = run(1)
    assert "split_finish_kernel" not in log1, log1
    got, log = run(parts)
    assert log.get("split_finish_kernel", 0) == 1 and len(log) == 2, log
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=1e-4, rtol=0)
    np.testing.assert_allclose(got.cpu().numpy(), whole.cpu().numpy(), atol=2e-5, rtol=0)
    again, _ = run(parts)
    assert torch.equal(got, again)                                   # parts are summed in a fixed order
    if math == "f16x2" and not pairs:
        # the guarded form (pre-scaled input) and an output row map go through the same second launch
        block = ops.absmax_rows(x)
        out_block = ops.absmax_blocks(1, x.device)[0]
        g, lg = run(parts, in_absmax=block, out_absmax=out_block, out_row_map=perm)
        assert any(k.startswith("rowwave_conv_f16s_kernel") for k in lg) and "split_finish_kernel" in lg, lg
        back = torch.empty_like(g)
        back[:] = g[perm.long()]
        np.testing.assert_allclose(back.cpu().numpy(), ref, atol=1e-4, rtol=0)
        assert ops.absmax_value(out_block) == float(g.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [1024, 4096, 8192, 16384])
def test_tap_pattern_row_order(oracle, hip, chunk):
    """ops.order_rows_by_taps: a permutation that stays inside chunks of `chunk` canonical rows, sorts every chunk by the rows'
    27-bit neighbour pattern (ties by canonical position: deterministic), and -- installed in the level's index -- makes the
    sub-manifold rulebook built over the re-ordered site list the canonical rulebook seen through the permutation."""
    import torch
    from cpd_amd import ops
    rng = np.random.default_rng(chunk)
    shape, B = [9, 70, 80], 2
    cells = np.unique(np.stack([rng.integers(0, B, 40000), rng.integers(0, shape[0], 40000), rng.integers(0, shape[1], 40000),
                                rng.integers(0, shape[2], 40000)], 1), axis=0).astype(np.int32)
    idx = torch.from_numpy(cells).cuda()
    index = ops.SiteIndex.build(idx, B, shape)
    nbr_c = ops.rulebook_subm(idx, index).cpu().numpy()
    new_idx, n2o, o2n = ops.order_rows_by_taps(idx, index, chunk_rows=chunk)
    n2o, o2n = n2o.cpu().numpy(), o2n.cpu().numpy()
    n = len(cells)
    assert np.array_equal(np.sort(n2o), np.arange(n)) and np.array_equal(o2n[n2o], np.arange(n))
    assert np.array_equal(n2o // chunk, np.arange(n) // chunk)                         # rows stay in their chunk
    np.testing.assert_array_equal(new_idx.cpu().numpy(), cells[n2o])
    pattern = ((nbr_c >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
    key = (np.arange(n) // chunk) * (1 << 44) + pattern * (1 << 16) + np.arange(n) % chunk
    np.testing.assert_array_equal(n2o, np.argsort(key, kind="stable"))
    index.set_order(torch.from_numpy(o2n).cuda())
    nbr_p = ops.rulebook_subm(new_idx, index)
    want = nbr_c[:, n2o]
    want = np.where(want >= 0, o2n[np.clip(want, 0, None)], -1)
    np.testing.assert_array_equal(nbr_p.cpu().numpy(), want)
    groups = (n + 15) // 16
    hit = np.zeros((27, groups * 16), bool)
    hit[:, :n] = want >= 0
    tm = (hit.reshape(27, groups, 16).any(-1).astype(np.int64) << np.arange(27)[:, None]).sum(0)
    np.testing.assert_array_equal(nbr_p.tapmask.cpu().numpy().astype(np.int64) & ((1 << 27) - 1), tm)
    index.set_order(None)                                                               # canonical again
    np.testing.assert_array_equal(ops.rulebook_subm(idx, index).cpu().numpy(), nbr_c)


@pytest.mark.gpu
def test_kernels_with_more_than_32_taps_fail_loudly_or_run_exactly(oracle, hip):
    """The wave kernel keeps a tile's taps in 32-bit sets: a 45-tap kernel (3 x 3 x 5) on 16 channels has no kernel that can take it
    and is REFUSED (it used to run the first 32 taps); on 64 channels with split arithmetic it goes to the workgroup kernel and matches the oracle."""
    import torch
    from cpd_amd import ops
    from cpd_amd._lib import CpdHipError
    rng = np.random.default_rng(5)
    batch, shape, ks = 2, [11, 96, 96], [3, 3, 5]
    idx = random_sites(rng, batch, shape, 60000).astype(np.int32)
    d_idx = dev(idx)
    index = ops.SiteIndex.build(d_idx, batch, shape)
    nbr = ops.rulebook_subm(d_idx, index, ksize=ks)
    kv, rows = 45, idx.shape[0]
    assert nbr.shape[0] == kv
    for c, ok in ((16, False), (64, True)):
        feat = rng.normal(size=(rows, c)).astype(np.float32)
        w = (rng.normal(size=(c, 3, 3, 5, c)) * np.sqrt(2.0 / (kv * c))).astype(np.float32)
        packed = ops.pack_weight(torch.from_numpy(w).reshape(c, -1, c).permute(1, 2, 0).contiguous().cuda())
        table = nbr.clone()                    # (no tap masks: they are 32-bit)
        if not ok:
            with pytest.raises(CpdHipError):
                ops.gather_conv(dev(feat), c, packed, table, kv, rows, c)
            continue
        with ops.launch_log() as log:
            got = ops.gather_conv(dev(feat), c, packed, table, kv, rows, c, math="f16x2").cpu().numpy()
        assert all(k.startswith(("tile_conv_f16_kernel", "split_finish")) for k in log.counts), log.counts
        np.testing.assert_allclose(got, oracle.sparse_conv(feat, w, None, nbr.cpu().numpy()), atol=1e-4, rtol=0)



@pytest.mark.gpu
def test_chunk_ordered_rulebooks_equal_the_plain_builders(hip):
    """cpd_rulebook_chunk_ordered (round 4): the SubM and the strided tables of a tap-ordered level, built chunk by chunk in canonical
    order and re-ordered in LDS, are bit for bit the tables (and tap masks) cpd_rulebook_subm / cpd_rulebook_conv build over the
    re-ordered list -- on a level with several chunks, a ragged last chunk, and an input level that is itself re-ordered."""
    import torch
    from cpd_amd import ops
    ops.CHUNKED_MIN_ROWS, keep_min = 0, ops.CHUNKED_MIN_ROWS          # (the level here is far below the size the engine switches at)
    rng = np.random.default_rng(77)
    shape, B = [21, 120, 130], 2
    cells = np.unique(np.stack([rng.integers(0, B, 60000), rng.integers(0, shape[0], 60000), rng.integers(0, shape[1], 60000),
                                rng.integers(0, shape[2], 60000)], 1), axis=0).astype(np.int32)
    idx = torch.from_numpy(cells).cuda()
    index = ops.SiteIndex.build(idx, B, shape)
    new_idx, _, o2n = ops.order_rows_by_taps(idx, index, chunk_rows=4096)
    index.set_order(o2n)
    plain = ops.rulebook_subm(new_idx, index)
    chunked = ops.rulebook_subm(new_idx, index, canonical=(idx, o2n, 4096))
    assert new_idx.shape[0] % 4096 != 0 and new_idx.shape[0] > 3 * 4096
    assert torch.equal(plain, chunked) and torch.equal(plain.tapmask, chunked.tapmask)
    # strided level on top of the re-ordered one
    k, s, pd = [3, 3, 3], [2, 2, 2], [1, 1, 1]
    out_c, out_index, out_shape = ops.conv_outset(idx, B, shape, k, s, pd)
    out_new, _, out_o2n = ops.order_rows_by_taps(out_c, out_index, chunk_rows=4096)
    plain_dn = ops.rulebook_conv(out_new, index, k, s, pd)
    chunked_dn = ops.rulebook_conv(out_new, index, k, s, pd, canonical=(out_c, out_o2n, 4096))
    assert torch.equal(plain_dn, chunked_dn) and torch.equal(plain_dn.tapmask, chunked_dn.tapmask)
    # canonical output order (no map) over the re-ordered input, both paddings of the backbone
    for pad in ([1, 1, 1], [0, 1, 1]):
        a = ops.rulebook_conv(out_c, index, k, s, pad)
        b = ops.rulebook_conv(out_c, index, k, s, pad, canonical=(out_c, None, 4096))
        assert torch.equal(a, b) and torch.equal(a.tapmask, b.tapmask)
    ops.CHUNKED_MIN_ROWS = keep_min


@pytest.mark.parametrize("c", [32, 64, 128])
def test_rowwave_lds_epilogue_on_fp32_rows_equals_the_shared_epilogue(hip, c, monkeypatch):
    """... and the fp32-row forms (`rowwave_conv_f16e_kernel`, and `f16se` with a range block on the input: the module path, the train
    step's forward, a guarded re-run): fp32 residual pieces in, fp32 row pieces out."""
    import torch
    from cpd_amd import ops
    rng = np.random.default_rng(c + 1)
    batch, shape = 2, [9, 128, 128]
    idx = random_sites(rng, batch, shape, 90001)
    d_idx = dev(idx)
    rows = idx.shape[0]
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(rows, c, generator=g) * 2).cuda()
    res = torch.randn(rows, c, generator=g).cuda()
    packed = ops.pack_weight((torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).cuda())
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()
    monkeypatch.setenv("CPD_TUNE", "1")
    for guarded in (False, True):
        am = ops.absmax_rows(x) if guarded else None
        for r_, relu in ((res, True), (None, False)):
            outs = {}
            for epi in ("0", "1"):
                monkeypatch.setenv("CPD_GC_RW_EPI", epi)
                with ops.launch_log() as log:
                    outs[epi] = ops.gather_conv(x, c, packed, nbr, 27, rows, c, scale, shift, r_, relu, math="f16x2", in_absmax=am)
                assert list(log.counts) == ["rowwave_conv_f16%s%s_kernel<%d,2>" % ("s" if guarded else "", "e" if epi == "1" else "", c)], log.counts
            np.testing.assert_allclose(outs["1"].cpu().numpy(), outs["0"].cpu().numpy(), atol=4e-6, rtol=0)


@pytest.mark.parametrize("c", [32, 64, 128])
def test_rowwave_lds_epilogue_equals_the_shared_epilogue(hip, c, monkeypatch):
    """Round 4: `rowwave_conv_f16pe_kernel` -- the pair-row row-wave kernel with its epilogue through LDS (transposed tile, 16-byte
    residual pieces in, 16-byte pair pieces out) -- against `rowwave_conv_f16p_kernel` (fragment-shaped epilogue) on the same launch:
    same accumulators, the same scale / shift / residual / ReLU arithmetic (<= 1 ulp apart where the compiler contracts differently),
    with and without the residual, on a row count that is not a multiple of the tile. Both against the oracle elsewhere in this file."""
    import torch
    from cpd_amd import ops
    rng = np.random.default_rng(c)
    batch, shape = 2, [9, 128, 128]
    idx = random_sites(rng, batch, shape, 90001)          # >= 600 row tiles: no tap split (a split launch finishes in split_finish_kernel)
    d_idx = dev(idx)
    rows = idx.shape[0]
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    g = torch.Generator().manual_seed(c)
    xp = ops.rows_to_pairs((torch.randn(rows, c, generator=g) * 2).cuda())
    rp = ops.rows_to_pairs(torch.randn(rows, c, generator=g).cuda())
    packed = ops.pack_weight((torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).cuda())
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.1).cuda()
    monkeypatch.setenv("CPD_TUNE", "1")
    for res, relu in ((rp, True), (None, False)):
        outs = {}
        for epi in ("0", "1"):
            monkeypatch.setenv("CPD_GC_RW_EPI", epi)
            rb = ops.absmax_blocks(1, xp.device)[0]
            with ops.launch_log() as log:
                outs[epi] = ops.gather_conv(xp, c, packed, nbr, 27, rows, c, scale, shift, res, relu, math="f16x2", in_pairs=True, out_pairs=True,
                                            res_pairs=res is not None, out_absmax=rb)
            assert list(log.counts) == ["rowwave_conv_f16p%s_kernel<%d,2>" % ("e" if epi == "1" else "", c)], log.counts
            outs[epi] = (ops.pairs_to_rows(outs[epi]), int(rb.max()))
        np.testing.assert_allclose(outs["1"][0].cpu().numpy(), outs["0"][0].cpu().numpy(), atol=4e-6, rtol=0)
        assert abs(outs["1"][1] - outs["0"][1]) <= 2          # the range guard's absmax word (float bits): the same maximum


@pytest.mark.parametrize("cout", [16, 32])
def test_h16_lds_epilogue_equals_the_shared_epilogue(hip, cout, monkeypatch):
    """Round 4: the level-1 kernels (`gather_conv_h16_kernel`: 16-channel pair rows in, 16- or 32-channel pair rows out) with their
    epilogue through LDS (CPD_GC_H16_EPI, the default) against the shared fragment-shaped epilogue on the same launch: <= 1 ulp apart,
    with the pair residual + ReLU of a SparseBasicBlock (16 -> 16) and without (the 16 -> 32 down-sampling layer's SubM stand-in)."""
    import torch
    from cpd_amd import ops
    rng = np.random.default_rng(cout)
    batch, shape = 2, [9, 128, 128]
    idx = random_sites(rng, batch, shape, 50001)
    d_idx = dev(idx)
    rows = idx.shape[0]
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    g = torch.Generator().manual_seed(cout)
    xp = ops.rows_to_pairs((torch.randn(rows, 16, generator=g) * 2).cuda())
    rp = ops.rows_to_pairs(torch.randn(rows, cout, generator=g).cuda())
    packed = ops.pack_weight((torch.randn(27, 16, cout, generator=g) * (2.0 / (27 * 16)) ** 0.5).cuda())
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    monkeypatch.setenv("CPD_TUNE", "1")
    for res, relu in ((rp, True), (None, False)):
        outs = {}
        for epi in ("0", "1"):
            monkeypatch.setenv("CPD_GC_H16_EPI", epi)
            with ops.launch_log() as log:
                outs[epi] = ops.pairs_to_rows(ops.gather_conv(xp, 16, packed, nbr, 27, rows, cout, scale, shift, res, relu, math="f16x2", in_pairs=True,
                                                              out_pairs=True, res_pairs=res is not None))
            assert len(log.counts) == 1 and next(iter(log.counts)).startswith("gather_conv_h16_kernel<"), log.counts
        np.testing.assert_allclose(outs["1"].cpu().numpy(), outs["0"].cpu().numpy(), atol=4e-6, rtol=0)
