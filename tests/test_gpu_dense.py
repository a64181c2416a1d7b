"""BEV dense head on the GPU: Conv2d / ConvTranspose2d (+BN+ReLU) through the fp32-MFMA gather
kernel with dense pixel rulebooks, against (a) the oracle's direct convolutions and (b) the golden
vectors produced by the reference's own BaseBEVBackbone / SeparateHead (tests/golden)."""
import numpy as np
import pytest
import torch

from cpd_amd import ops

pytestmark = pytest.mark.gpu


def nhwc_rows(x):                      # (B,C,H,W) numpy -> [B*H*W, C] device rows
    b, c, h, w = x.shape
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(b * h * w, c))).cuda()


def rows_nchw(rows, b, h, w):
    return rows.cpu().numpy().reshape(b, h, w, -1).transpose(0, 3, 1, 2)


MATHS = ["f32", "bf16x3", "f16x2"]          # ModelConfig.conv_math values


@pytest.fixture
def small_tiles(monkeypatch):
    """Let the split kernels (workgroup tiles, window kernel) take problems far below their production thresholds, so that
    the small reference goldens run on the SAME kernel families the bench runs (the thresholds are speed-only)."""
    monkeypatch.setenv("CPD_TUNE", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN64", "1")


def conv2d_hip(x_rows, b, h, w, wt, bias=None, stride=1, scale=None, shift=None, relu=False, math="f32", kernels=None):
    """Conv2d(k, stride, pad 1) on channels-last rows exactly the way the engine issues it (dense flag, pixel table with its
    image tag). `kernels`: list that collects the name of the kernel instantiation each call ran."""
    cout, cin, kh, kw = wt.shape
    nbr, ho, wo = ops.rulebook_conv2d(b, h, w, kh, kw, stride, 1, "cuda")
    packed = ops.pack_weight(torch.from_numpy(wt).permute(2, 3, 1, 0).reshape(kh * kw, cin, cout).contiguous().cuda())
    if shift is None and bias is not None:
        shift = bias
    if kernels is not None:
        kernels.append(ops.gather_conv_tile(b * ho * wo, cin, cout, x_rows.stride(0), dense=True, nbr=nbr, math=math))
    out = ops.gather_conv(x_rows, cin, packed, nbr, kh * kw, b * ho * wo, cout,
                          torch.from_numpy(scale).cuda() if scale is not None else None,
                          torch.from_numpy(shift).cuda() if shift is not None else None, None, relu, dense=True, math=math)
    return out, ho, wo


@pytest.mark.parametrize("cin,cout,stride,h,w", [(32, 16, 1, 24, 20), (64, 64, 1, 19, 23), (16, 32, 2, 24, 20), (128, 256, 2, 17, 17),
                                                 (256, 128, 1, 12, 12), (320, 11, 1, 16, 16)])
@pytest.mark.parametrize("math", MATHS)
def test_conv2d_matches_oracle(oracle, hip, small_tiles, math, cin, cout, stride, h, w):
    rng = np.random.default_rng(cin + cout)
    b = 2
    x = rng.normal(size=(b, cin, h, w)).astype(np.float32)
    wt = (rng.normal(size=(cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    want = oracle.conv2d(x, wt, bias, stride, 1)
    got, ho, wo = conv2d_hip(nhwc_rows(x), b, h, w, wt, bias, stride, math=math)
    np.testing.assert_allclose(rows_nchw(got, b, ho, wo), want, atol=1e-4, rtol=0)


# The layer shapes of BASELINE config 2's dense half at sizes where gather_conv picks the kernels the bench runs
# (>= 600 workgroups), against the ORACLE's direct convolution: (cin, cout, k, stride, batch, h, w, expected kernel)
BENCH_SHAPES = [
    (128, 128, 3, 1, 3, 188, 188, "window_conv_bf16_kernel<128>"),      # BEV block 0, 5 layers
    (256, 128, 3, 1, 3, 188, 188, "window_conv_bf16_kernel<128>"),      # BEV block 0, first conv
    (256, 256, 3, 1, 5, 94, 94, "window_conv_bf16_kernel<128>"),        # BEV block 1, 5 layers
    (128, 256, 3, 2, 5, 188, 188, "tile_conv_bf16_kernel<128,128>"),    # BEV block 1, strided conv
    (512, 64, 3, 1, 5, 188, 188, "window_conv_bf16_kernel<64>"),        # CenterHead shared conv
    (64, 320, 3, 1, 3, 188, 188, "window_conv_bf16_kernel<64>"),        # five SeparateHead first convs, fused
    (320, 11, 3, 1, 3, 188, 188, "window_conv_bf16_kernel<16>"),        # five SeparateHead output convs, fused
]


@pytest.mark.parametrize("math", [m for m in MATHS if m != "f32"])
@pytest.mark.parametrize("cin,cout,k,stride,batch,h,w,kernel", BENCH_SHAPES)
def test_bench_kernels_match_oracle_at_full_size(oracle, hip, math, cin, cout, k, stride, batch, h, w, kernel):
    """the kernels that carry the bench, at the bench's image sizes, against the oracle (not against
    a self-authored reference): conv + bias, then the folded-BN affine + ReLU epilogue on top."""
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.normal(size=(batch, cin, h, w)).astype(np.float32)
    wt = (rng.normal(size=(cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    ran = []
    got, ho, wo = conv2d_hip(nhwc_rows(x), batch, h, w, wt, None, stride, scale, shift, True, math=math, kernels=ran)
    assert ran == [kernel.replace("bf16", "f16") if math == "f16x2" else kernel], ran
    want = np.maximum(oracle.conv2d(x, wt, None, stride, 1) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    np.testing.assert_allclose(rows_nchw(got, batch, ho, wo), want, atol=1e-4, rtol=0)


@pytest.mark.parametrize("cin,cout,batch,kernel", [(512, 64, 8, "window_conv_f16_kernel<64,256>"), (320, 11, 8, "window_conv_f16_kernel<16,256>"),
                                                   (64, 320, 8, "window_conv_f16_kernel<64,256>")])
def test_bench_256_row_window_tiles_match_oracle(oracle, hip, cin, cout, batch, kernel):
    """The 256-row window tiles the 64- and 16-column layers (CenterHead shared conv, the fused 64 -> 320 first head convs with their
    five column tiles, the fused output convs) get at the bench's batch size (>= 1024 such row tiles), against the oracle at 188 x 188."""
    rng = np.random.default_rng(cin + cout + 256)
    h = w = 188
    x = rng.normal(size=(batch, cin, h, w)).astype(np.float32)
    wt = (rng.normal(size=(cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    ran = []
    got, ho, wo = conv2d_hip(nhwc_rows(x), batch, h, w, wt, bias, 1, math="f16x2", kernels=ran)
    assert ran == [kernel], ran
    np.testing.assert_allclose(rows_nchw(got, batch, ho, wo), oracle.conv2d(x, wt, bias, 1, 1), atol=1e-4, rtol=0)


@pytest.mark.parametrize("math", [m for m in MATHS if m != "f32"])
@pytest.mark.parametrize("cin,cout,u,batch,h,w", [(128, 256, 1, 3, 188, 188), (256, 256, 2, 5, 94, 94)])
def test_bench_deconv_kernels_match_oracle_at_full_size(oracle, hip, math, cin, cout, u, batch, h, w):
    """The two deblocks (ConvTranspose2d k = s = u as one 1x1 GEMM with a column-group scatter) at bench size."""
    rng = np.random.default_rng(cin + cout + u)
    x = rng.normal(size=(batch, cin, h, w)).astype(np.float32)
    wd = (rng.normal(size=(cin, cout, u, u)) * np.sqrt(2.0 / cin)).astype(np.float32)
    want = oracle.deconv2d(x, wd, u)
    packed = ops.pack_weight(torch.from_numpy(wd).permute(0, 2, 3, 1).reshape(1, cin, u * u * cout).contiguous().cuda())
    n = batch * h * w
    assert ops.gather_conv_tile(n, cin, u * u * cout, cin, dense=True, math=math) == "tile_conv_%s_kernel<128,128>" % ("f16" if math == "f16x2" else "bf16")
    H, W = h * u, w * u
    out = torch.empty((batch * H * W, cout), device="cuda")
    if u == 1:
        ops.gather_conv(nhwc_rows(x), cin, packed, None, 1, n, cout, out=out, dense=True, math=math)
    else:
        bi = torch.arange(batch, device="cuda").view(-1, 1, 1); yy = torch.arange(h, device="cuda").view(1, -1, 1)
        xx = torch.arange(w, device="cuda").view(1, 1, -1)
        maps = torch.stack([((bi * H + 2 * yy + a) * W + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)])
        ops.gather_conv(nhwc_rows(x), cin, packed, None, 1, n, 4 * cout, out=out, out_row_map=maps.to(torch.int32).contiguous(),
                        out_col_group=cout, dense=True, math=math)
    np.testing.assert_allclose(rows_nchw(out, batch, H, W), want, atol=1e-4, rtol=0)


def test_deconv_k2s2_and_k1(oracle, hip):
    rng = np.random.default_rng(3)
    b, cin, cout, h, w = 2, 32, 48, 9, 11
    x = rng.normal(size=(b, cin, h, w)).astype(np.float32)
    for u in (1, 2):
        wd = (rng.normal(size=(cin, cout, u, u)) * 0.2).astype(np.float32)
        want = oracle.deconv2d(x, wd, u)
        w_kio = torch.from_numpy(wd).permute(0, 2, 3, 1).reshape(1, cin, u * u * cout).contiguous().cuda()
        packed = ops.pack_weight(w_kio)
        H, W = h * u, w * u
        # write into the right half of a wider concat buffer to exercise out_ld / column offsets
        cat = torch.zeros((b * H * W, cout + 16), device="cuda")
        dst = cat[:, 16:]
        if u == 1:
            ops.gather_conv(nhwc_rows(x), cin, packed, None, 1, b * h * w, cout, out=dst)
        else:
            bi = torch.arange(b, device="cuda").view(-1, 1, 1); yy = torch.arange(h, device="cuda").view(1, -1, 1)
            xx = torch.arange(w, device="cuda").view(1, 1, -1)
            maps = torch.stack([((bi * H + 2 * yy + a) * W + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)])
            ops.gather_conv(nhwc_rows(x), cin, packed, None, 1, b * h * w, 4 * cout, out=dst,
                            out_row_map=maps.to(torch.int32).contiguous(), out_col_group=cout)
        np.testing.assert_allclose(rows_nchw(cat[:, 16:], b, H, W), want, atol=1e-4, rtol=0)
        assert float(cat[:, :16].abs().max()) == 0.0


def _fold(g, prefix, eps, bias=None):
    s = g[prefix + ".weight"] / np.sqrt(g[prefix + ".running_var"] + np.float32(eps))
    t = g[prefix + ".bias"] - g[prefix + ".running_mean"] * s
    if bias is not None:
        t = t + bias * s
    return s.astype(np.float32), t.astype(np.float32)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("fixture,n_layers", [("bev_backbone", 2), ("bev_backbone_wide", 1)])
def test_bev_backbone_matches_reference_golden(hip, golden, small_tiles, math, fixture, n_layers):
    """BaseBEVBackbone.forward (base_bev_backbone.py:85-122) reproduced on reference weights, in every conv arithmetic.
    The `_wide` fixture has 64/128-channel layers: with `small_tiles` the split modes run it on the workgroup / window
    kernels the bench runs (asserted below); the narrow fixture's 16-channel layers take the fp32 kernels in every mode."""
    ran = []
    g = golden(fixture)
    x = g["bev_in"]
    b, _, h, w = x.shape
    rows = nhwc_rows(x)
    outs = []
    cur_h, cur_w = h, w
    for lvl, (stride, u) in enumerate([(1, 1), (2, 2)]):
        p = "sd.blocks.%d." % lvl
        s, t = _fold(g, p + "2", 1e-3)
        rows, cur_h, cur_w = conv2d_hip(rows, b, cur_h, cur_w, g[p + "1.weight"], None, stride, s, t, True, math=math, kernels=ran)
        for k in range(n_layers):
            s, t = _fold(g, p + "%d" % (5 + 3 * k), 1e-3)
            rows, cur_h, cur_w = conv2d_hip(rows, b, cur_h, cur_w, g[p + "%d.weight" % (4 + 3 * k)], None, 1, s, t, True, math=math,
                                            kernels=ran)
        q = "sd.deblocks.%d." % lvl
        wd = g[q + "0.weight"]
        cin, cout = wd.shape[:2]
        s, t = _fold(g, q + "1", 1e-3)
        packed = ops.pack_weight(torch.from_numpy(wd).permute(0, 2, 3, 1).reshape(1, cin, u * u * cout).contiguous().cuda())
        sc, sh = torch.from_numpy(np.tile(s, u * u)).cuda(), torch.from_numpy(np.tile(t, u * u)).cuda()
        up = torch.empty((b * h * w, cout), device="cuda")
        if u == 1:
            ops.gather_conv(rows, cin, packed, None, 1, b * cur_h * cur_w, cout, sc, sh, None, True, out=up, dense=True, math=math)
        else:
            bi = torch.arange(b, device="cuda").view(-1, 1, 1); yy = torch.arange(cur_h, device="cuda").view(1, -1, 1)
            xx = torch.arange(cur_w, device="cuda").view(1, 1, -1)
            maps = torch.stack([((bi * h + 2 * yy + a) * w + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)])
            ops.gather_conv(rows, cin, packed, None, 1, b * cur_h * cur_w, 4 * cout, sc, sh, None, True, out=up,
                            out_row_map=maps.to(torch.int32).contiguous(), out_col_group=cout, dense=True, math=math)
        outs.append(rows_nchw(up, b, h, w))
    got = np.concatenate(outs, 1)
    np.testing.assert_allclose(got, g["bev_out"], atol=1e-4, rtol=0)
    if math != "f32" and fixture.endswith("_wide"):
        assert all(k.startswith(("window_conv_", "tile_conv_bf16", "tile_conv_f16")) for k in ran), ran


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("fixture", ["center_head", "center_head_wide"])
def test_center_head_matches_reference_golden(hip, golden, small_tiles, math, fixture):
    """shared_conv + SeparateHead (center_head.py:11-45,73-80) on reference weights (`_wide`: 128 -> 64 -> 64 -> c channels,
    which the split modes run on the window kernels: <64> for the 64-column convs, <16> for the output convs)."""
    g = golden(fixture)
    ran = []
    x = g["head_in"]
    b, _, h, w = x.shape
    s, t = _fold(g, "shared.1", 1e-5, g["shared.0.bias"])
    mid, _, _ = conv2d_hip(nhwc_rows(x), b, h, w, g["shared.0.weight"], None, 1, s, t, True, math=math, kernels=ran)
    np.testing.assert_allclose(rows_nchw(mid, b, h, w), g["shared_out"], atol=1e-4, rtol=0)
    for name in ["center", "center_z", "dim", "rot", "hm"]:
        p = "sep.%s." % name
        s, t = _fold(g, p + "0.1", 1e-5, g[p + "0.0.bias"])
        hcur, _, _ = conv2d_hip(mid, b, h, w, g[p + "0.0.weight"], None, 1, s, t, True, math=math, kernels=ran)
        out, _, _ = conv2d_hip(hcur, b, h, w, g[p + "1.weight"], g[p + "1.bias"], 1, math=math, kernels=ran)
        np.testing.assert_allclose(rows_nchw(out, b, h, w), g["out." + name], atol=1e-4, rtol=0)
    if math != "f32" and fixture.endswith("_wide"):
        assert all(k.startswith("window_conv_") for k in ran), ran


@pytest.mark.parametrize("bm,bn", [(64, 64), (64, 128), (128, 64), (128, 128)])
def test_workgroup_kernel_matches_wave_kernel_and_oracle(oracle, hip, bm, bn, monkeypatch):
    """tile_conv_kernel<BM,BN> (LDS-staged) against the oracle, incl. ragged row tails, stride 2,
    BN-affine + ReLU epilogue and the deconv column-group scatter."""
    rng = np.random.default_rng(bm + bn)
    b, cin, cout, h, w = 2, 64, 128, 21, 19
    x = rng.normal(size=(b, cin, h, w)).astype(np.float32)
    wt = (rng.normal(size=(cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    monkeypatch.setenv("CPD_TUNE", "1")          # the knobs below are only read when this is set
    monkeypatch.setenv("CPD_GC_WG", "1"); monkeypatch.setenv("CPD_GC_BM", str(bm)); monkeypatch.setenv("CPD_GC_BN", str(bn))
    assert ops.gather_conv_tile(b * h * w, cin, cout, cin) == "tile_conv_kernel<%d,%d>" % (bm, bn)
    for stride in (1, 2):
        want = np.maximum(oracle.conv2d(x, wt, None, stride, 1) * scale[None, :, None, None] + shift[None, :, None, None], 0)
        got, ho, wo = conv2d_hip(nhwc_rows(x), b, h, w, wt, None, stride, scale, shift, True)
        np.testing.assert_allclose(rows_nchw(got, b, ho, wo), want, atol=1e-4, rtol=0)
    # 1x1 with column groups (ConvTranspose2d k=s=2)
    wd = (rng.normal(size=(cin, 64, 2, 2)) * 0.2).astype(np.float32)
    want = oracle.deconv2d(x, wd, 2)
    packed = ops.pack_weight(torch.from_numpy(wd).permute(0, 2, 3, 1).reshape(1, cin, 256).contiguous().cuda())
    H, W = 2 * h, 2 * w
    bi = torch.arange(b, device="cuda").view(-1, 1, 1); yy = torch.arange(h, device="cuda").view(1, -1, 1)
    xx = torch.arange(w, device="cuda").view(1, 1, -1)
    maps = torch.stack([((bi * H + 2 * yy + a) * W + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)])
    out = torch.zeros((b * H * W, 64), device="cuda")
    ops.gather_conv(nhwc_rows(x), cin, packed, None, 1, b * h * w, 256, out=out, out_row_map=maps.to(torch.int32).contiguous(),
                    out_col_group=64)
    np.testing.assert_allclose(rows_nchw(out, b, H, W), want, atol=1e-4, rtol=0)


@pytest.mark.parametrize("math", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("cin,cout,k,stride", [(128, 128, 3, 1), (256, 256, 3, 1), (128, 256, 3, 2), (256, 1024, 1, 1)])
def test_split_bf16_conv_is_fp32_accurate(hip, math, cin, cout, k, stride):
    """CPD_GC_BF16X3: fp32 operands split exactly into three bf16 terms, six partial products on the
    bf16 matrix pipe. Against a float64 reference it must be as accurate as the fp32-MFMA kernel
    (<= 1e-4 absolute at O(20) outputs, and within 1.5x of the fp32 kernel's own error)."""
    from cpd_amd import ops
    torch.manual_seed(cin + cout)
    batch, hw = 2, 188
    if k == 3:
        nbr, ho, wo = ops.rulebook_conv2d(batch, hw, hw, 3, 3, stride, 1, "cuda"); kv = 9
    else:
        nbr, ho, wo, kv = None, hw, hw, 1
    n_in, n_out = batch * hw * hw, batch * ho * wo
    x = torch.randn(n_in, cin, device="cuda") * 3.0
    w = torch.randn(kv, cin, cout, device="cuda") * (2.0 / (kv * cin)) ** 0.5
    pw = ops.pack_weight(w)
    assert ops.gather_conv_tile(n_out, cin, cout, cin, dense=True, math=math).startswith(("tile_conv_bf16_kernel", "tile_conv_f16_kernel"))
    fast = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=True, math=math)
    exact = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, dense=True, bf16x3=False)
    rows = torch.randint(0, n_out, (2048,), device="cuda")
    if nbr is None:
        ref = x[rows].double() @ w[0].double()
    else:
        idx = nbr[:, rows].long()
        xp = torch.cat([x, x.new_zeros(1, cin)]).double()
        ref = sum(xp[torch.where(idx[t] < 0, n_in, idx[t])] @ w[t].double() for t in range(kv))
    e_fast = (fast[rows].double() - ref).abs().max().item()
    e_exact = (exact[rows].double() - ref).abs().max().item()
    assert e_fast <= 1e-4, (e_fast, e_exact)
    assert e_fast <= 1.5 * e_exact + 1e-6, (e_fast, e_exact)
    # scale invariance of the weights: bf16x3 splits exactly at any magnitude; f16x2 pre-scales every output column by a power
    # of two, so tiny / huge weights cost nothing either
    for wscale in (1e-6, 1e+5):
        pws = ops.pack_weight(w * wscale)
        got = ops.gather_conv(x, cin, pws, nbr, kv, n_out, cout, dense=True, math=math)
        e_w = (got[rows].double() - ref * wscale).abs().max().item()
        assert e_w <= 1.5 * wscale * max(e_fast, e_exact) + 1e-30, (wscale, e_w, e_fast)
    # scale of the activations: exact at any magnitude for bf16x3; f16x2 has an ABSOLUTE floor of 2^-25 per activation
    # (fp16 subnormals), i.e. tiny activations lose relative -- never absolute -- accuracy
    small = ops.gather_conv(x * 1e-4, cin, pw, nbr, kv, n_out, cout, dense=True, math=math)
    e_small = (small[rows].double() - ref * 1e-4).abs().max().item()
    if math == "bf16x3":
        assert e_small <= 1.5e-4 * max(e_fast, e_exact) + 1e-12, (e_small, e_fast)
    else:
        assert e_small <= 2e-7, (e_small, e_fast)
        # the range contract: an activation beyond fp16's range is LOUD (inf / NaN in the rows it feeds), not silently wrong
        xb = x.clone()
        xb[rows[0]] = 7e4
        bad = ops.gather_conv(xb, cin, pw, nbr, kv, n_out, cout, dense=True, math=math)
        assert not torch.isfinite(bad).all()
        # ... and GUARDED: with the input's absmax block the kernel pre-scales by a power of two and the same
        # input gives the fp32 answer -- measured here (guard=True), or left behind by the producing layer's epilogue (below)
        want = ops.gather_conv(xb, cin, pw, nbr, kv, n_out, cout, dense=True, math="f32")
        good = ops.gather_conv(xb, cin, pw, nbr, kv, n_out, cout, dense=True, math=math, guard=True)
        assert torch.isfinite(good).all()
        rowmax = want.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        assert float(((good - want).abs() / rowmax).max()) <= 1e-4
        # producer -> consumer: layer 1 (weights x 3e4: outputs ~1e5..1e6) records max |out|, layer 2 reads it
        blocks = ops.absmax_blocks(2, "cuda")
        big = ops.pack_weight(w * 3e4)
        y1 = ops.gather_conv(x, cin, big, nbr, kv, n_out, cout, None, None, None, True, dense=True, math=math, guard=True, out_absmax=blocks[0])
        assert ops.absmax_value(blocks[0]) == float(y1.abs().max()) and ops.absmax_value(blocks[0]) > 65504
        if cout % 32 == 0 and k == 3 and stride == 1:
            w2 = torch.randn(kv, cout, 64, device="cuda") * (2.0 / (kv * cout)) ** 0.5
            pw2 = ops.pack_weight(w2)
            y2 = ops.gather_conv(y1, cout, pw2, nbr, kv, n_out, 64, dense=True, math=math, in_absmax=blocks[0], out_absmax=blocks[1])
            y2_ref = ops.gather_conv(y1, cout, pw2, nbr, kv, n_out, 64, dense=True, math="f32")
            assert torch.isfinite(y2).all()
            assert float((y2 - y2_ref).abs().max()) <= 1e-4 * float(y2_ref.abs().max())
            assert ops.absmax_value(blocks[1]) == float(y2.abs().max())
            assert not torch.isfinite(ops.gather_conv(y1, cout, pw2, nbr, kv, n_out, 64, dense=True, math=math)).all()     # unguarded: loud


@pytest.mark.parametrize("cin,cout,batch,h,w", [(128, 128, 3, 188, 188), (256, 256, 5, 94, 94), (64, 320, 2, 188, 188),
                                                 (512, 64, 3, 188, 188), (32, 64, 3, 131, 200), (320, 11, 3, 188, 188), (64, 3, 3, 188, 188)])
def test_window_conv_matches_the_table_path(hip, cin, cout, batch, h, w):
    """cpd_conv3x3_rows (no rulebook: one gathered + split window per dy, image borders masked on the fragments)
    against cpd_gather_conv on the pixel table, with the full epilogue, and against float64 on sampled rows --
    rows at image corners / edges / frame boundaries included."""
    from cpd_amd import ops
    from cpd_amd._lib import lib
    torch.manual_seed(cin * 3 + cout)
    nbr, ho, wo = ops.rulebook_conv2d(batch, h, w, 3, 3, 1, 1, "cuda")
    assert (ho, wo) == (h, w) and nbr.image == (batch, h, w)
    n = batch * h * w
    assert lib().cpd_conv3x3_rows_supported(batch, h, w, cin, cout, 3) == 1
    assert ops.gather_conv_tile(n, cin, cout, cin, dense=True, bf16x3=True, nbr=nbr).startswith("window_conv_bf16_kernel")
    x = torch.randn(n, cin, device="cuda") * 2.0
    wgt = torch.randn(9, cin, cout, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    pw = ops.pack_weight(wgt)
    scale = torch.rand(cout, device="cuda") + 0.5
    shift = torch.randn(cout, device="cuda")
    res = torch.randn(n, cout, device="cuda")
    got = ops.gather_conv(x, cin, pw, nbr, 9, n, cout, scale, shift, res, True, dense=True, bf16x3=True)
    plain = nbr.clone()                                  # same table without the geometry tag -> rulebook kernels
    want = ops.gather_conv(x, cin, pw, plain, 9, n, cout, scale, shift, res, True, dense=True, bf16x3=True)
    # narrow outputs take the fp32-MFMA wave kernel on the table path: two fp32-level results, different summation orders
    assert torch.allclose(got, want, rtol=1e-5, atol=2e-5 if cout % 64 == 0 else 1e-4), (got - want).abs().max().item()
    edge = [0, 1, w - 1, w, 2 * w - 1, (h - 1) * w, h * w - 1, h * w, h * w + w - 1, n - w, n - 1, n // 2, 127, 128, 129]
    rows = torch.cat([torch.tensor(edge, device="cuda"), torch.randint(0, n, (1024,), device="cuda")])
    idx = nbr[:, rows].long()
    xp = torch.cat([x, x.new_zeros(1, cin)]).double()
    ref = sum(xp[torch.where(idx[t] < 0, n, idx[t])] @ wgt[t].double() for t in range(9))
    ref = torch.relu(ref * scale.double() + shift.double() + res[rows].double())
    assert (got[rows].double() - ref).abs().max().item() <= 1e-4
    # strided input / output rows (the BEV concat buffer is written and read in column slices)
    xin = torch.randn(n, cin + 32, device="cuda")
    buf = torch.zeros(n, cout + 64, device="cuda")
    ops.gather_conv(xin[:, 32:], cin, pw, nbr, 9, n, cout, dense=True, bf16x3=True, out=buf[:, 64:])
    want2 = ops.gather_conv(xin[:, 32:], cin, pw, plain, 9, n, cout, dense=True, bf16x3=True)
    assert torch.allclose(buf[:, 64:], want2, rtol=1e-5, atol=2e-5 if cout % 64 == 0 else 1e-4) and not buf[:, :64].any()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw,k,stride,math", [(512, 64, 188, 3, 1, "f16x2"), (128, 256, 188, 3, 2, "f16x2"), (256, 256, 94, 3, 1, "bf16x3")])
def test_stage_split_of_small_dense_launches(hip, monkeypatch, cin, cout, hw, k, stride, math):
    """cpd_gather_conv_ws on the workgroup (tile) kernel: a one-frame dense layer with few workgroups and many (tap, 32-channel)
    stages (512 -> 64: 553 workgroups x 144 stages) gives contiguous shares of its stages to 2-4 workgroups per tile; the second
    launch adds the parts in order and runs the epilogue (BN, residual, ReLU, absmax). Equal to the unsplit launch to fp32 summation
    order, to the fp32 kernel to 1e-4 of the output's scale, and bit-identical from run to run."""
    monkeypatch.setenv("CPD_TUNE", "1")
    monkeypatch.setenv("CPD_GC_WINDOW", "0")               # the rulebook path (strided layers take it anyway)
    torch.manual_seed(cin + cout)
    nbr, ho, wo = ops.rulebook_conv2d(1, hw, hw, k, k, stride, 1, "cuda")
    n_in, n_out, kv = hw * hw, ho * wo, k * k
    x = torch.randn(n_in, cin, device="cuda")
    w = torch.randn(kv, cin, cout, device="cuda") * (2.0 / (kv * cin)) ** 0.5
    pw = ops.pack_weight(w)
    sc = torch.rand(cout, device="cuda") + 0.5
    sh = torch.randn(cout, device="cuda")
    res = torch.randn(n_out, cout, device="cuda")

    def run(split):
        monkeypatch.setenv("CPD_GC_SPLIT_TILE", str(split))
        blk = ops.absmax_blocks(1, x.device)[0]
        with ops.launch_log() as log:
            y = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, sc, sh, res, True, dense=True, math=math, out_absmax=blk)
        return y, log.counts, ops.absmax_value(blk)

    whole, log0, m0 = run(0)
    assert "split_finish_kernel" not in log0 and any(k_.startswith("tile_conv_") for k_ in log0), log0
    got, log1, m1 = run(1)
    assert log1.get("split_finish_kernel", 0) == 1 and any(k_.startswith("tile_conv_") for k_ in log1), log1
    exact = ops.gather_conv(x, cin, pw, nbr, kv, n_out, cout, sc, sh, res, True, dense=True, math="f32")
    tol = 1e-4 * float(exact.abs().max())
    assert float((got - exact).abs().max()) <= tol
    assert float((got - whole).abs().max()) <= 0.2 * tol
    assert m1 == float(got.abs().max()) and m0 == float(whole.abs().max())
    again, _, _ = run(1)
    assert torch.equal(got, again)



# ---------------------------------------------------------------------------------------------------------------------------------
# Round 5: fp16-PAIR dense maps (ModelConfig.pair_rows_dense; window_conv_f16p_kernel / tile_conv_f16p_kernel). The kernels take the
# stored (h, l) bits as MFMA fragments; against the ORACLE on the bench's layer shapes at the batch size that selects the pair tiles,
# and against the fp32-row kernels on pair-exact inputs (same partial products, same order: <= 5e-6).

def _pairs_case(rng, batch, cin, h, w):
    x = np.maximum(rng.normal(size=(batch, cin, h, w)), 0).astype(np.float32)              # a post-ReLU map
    rows = nhwc_rows(x)
    pairs = ops.rows_to_pairs(rows)
    exact = ops.pairs_to_rows(pairs)                                                        # what the pair map holds (h + l: 22 significand bits)
    return exact.cpu().numpy().reshape(batch, h, w, cin).transpose(0, 3, 1, 2), exact, pairs


PAIR_SHAPES = [
    (128, 128, 3, 1, 8, 188, 188, "window_conv_f16p_kernel<128,128>"),     # BEV block 0
    (256, 128, 3, 1, 8, 188, 188, "window_conv_f16p_kernel<128,128>"),     # BEV block 0, first conv (reads the densified pair map)
    (256, 256, 3, 1, 8, 94, 94, "window_conv_f16p_kernel<128,128>"),       # BEV block 1
    (128, 256, 3, 2, 8, 188, 188, "tile_conv_f16p_kernel<128,128>"),       # BEV block 1, strided conv
    (512, 64, 3, 1, 8, 188, 188, "window_conv_f16p_kernel<64,256>"),       # CenterHead shared conv (reads the pair concat map)
    (64, 320, 3, 1, 8, 188, 188, "window_conv_f16p_kernel<64,256>"),       # fused first head convs
    (320, 11, 3, 1, 8, 188, 188, "window_conv_f16p_kernel<16,256>"),       # fused output convs: pair rows in, fp32 rows out
    # the reference's eval batch (4 frames per GPU): the 128-row forms of the 64- and 16-column tiles
    (512, 64, 3, 1, 4, 188, 188, "window_conv_f16p_kernel<64,128>"),
    (64, 320, 3, 1, 4, 188, 188, "window_conv_f16p_kernel<64,128>"),
    (320, 11, 3, 1, 4, 188, 188, "window_conv_f16p_kernel<16,128>"),
    (256, 256, 3, 1, 2, 94, 94, "window_conv_f16p_kernel<64,128>"),        # two frames: 256 output columns as four 64-column tiles
]


@pytest.mark.parametrize("cin,cout,k,stride,batch,h,w,kernel", PAIR_SHAPES)
def test_pair_dense_kernels_match_oracle_and_the_fp32_row_kernels(oracle, hip, cin, cout, k, stride, batch, h, w, kernel):
    rng = np.random.default_rng(cin * 3 + cout + stride)
    x_nchw, exact, pairs = _pairs_case(rng, batch, cin, h, w)
    wt = (rng.normal(size=(cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32); shift = rng.normal(size=cout).astype(np.float32)
    nbr, ho, wo = ops.rulebook_conv2d(batch, h, w, k, k, stride, 1, "cuda")
    packed = ops.pack_weight(torch.from_numpy(wt).permute(2, 3, 1, 0).reshape(k * k, cin, cout).contiguous().cuda())
    sc, sh = torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda()
    n_out = batch * ho * wo
    pout = cout % 64 == 0
    blk = ops.absmax_blocks(1, "cuda")[0]
    with ops.launch_log() as log:
        got = ops.gather_conv(pairs, cin, packed, nbr, k * k, n_out, cout, sc, sh, None, True, dense=True, math="f16x2",
                              in_pairs=True, out_pairs=pout, out_absmax=blk)
    assert log.counts == {kernel: 1}, log.counts
    got_rows = ops.pairs_to_rows(got) if pout else got
    # the block holds the fp32 maximum; the stored h + l is that value rounded to 22 bits (it may round UP by a unit of the stored format)
    assert ops.absmax_value(blk) >= float(got_rows.abs().max()) * (1.0 - 2.0 ** -20)
    # (a) the fp32-row kernel on the same (pair-exact) input: the same products in the same order, output stored to 22 bits
    base = ops.gather_conv(exact, cin, packed, nbr, k * k, n_out, cout, sc, sh, None, True, dense=True, math="f16x2")
    want_rows = ops.pairs_to_rows(ops.rows_to_pairs(base)) if pout else base
    # (not bitwise: pair-exact inputs whose low term sits exactly on fp16's rounding tie re-split to another (h, l) with the same sum
    # in the fp32-row kernel, so single products move by 2^-22 relative and sums by a few fp32 ulps of their largest partial sum;
    # measured 1.9e-6 at |out| <= 8 on 36 M outputs -- fifty times inside the 1e-4 contract)
    assert float((got_rows - want_rows).abs().max()) <= 5e-6
    # (b) the oracle
    want = np.maximum(oracle.conv2d(x_nchw, wt, None, stride, 1) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    np.testing.assert_allclose(rows_nchw(got_rows, batch, ho, wo), want, atol=1e-4, rtol=0)


@pytest.mark.parametrize("cin,cout,u,batch,h,w", [(128, 256, 1, 8, 188, 188), (256, 256, 2, 8, 94, 94)])
def test_pair_deconv_kernels_match_oracle(oracle, hip, cin, cout, u, batch, h, w):
    """The two deblocks on pair maps, writing their column block of a 512-wide pair concat buffer (row stride 512, column offset 256),
    the ConvTranspose(k = s = 2) through the row map / column-group scatter."""
    rng = np.random.default_rng(cin + cout + u + 5)
    x_nchw, exact, pairs = _pairs_case(rng, batch, cin, h, w)
    wd = (rng.normal(size=(cin, cout, u, u)) * np.sqrt(2.0 / cin)).astype(np.float32)
    want = np.maximum(oracle.deconv2d(x_nchw, wd, u), 0)
    packed = ops.pack_weight(torch.from_numpy(wd).permute(0, 2, 3, 1).reshape(1, cin, u * u * cout).contiguous().cuda())
    n = batch * h * w
    H, W = h * u, w * u
    cat = torch.full((batch * H * W, 512), float("nan"), device="cuda")
    dst = cat[:, 256:256 + cout]
    with ops.launch_log() as log:
        if u == 1:
            ops.gather_conv(pairs, cin, packed, None, 1, n, cout, None, None, None, True, out=dst, dense=True, math="f16x2", in_pairs=True, out_pairs=True)
        else:
            bi = torch.arange(batch, device="cuda").view(-1, 1, 1); yy = torch.arange(h, device="cuda").view(1, -1, 1)
            xx = torch.arange(w, device="cuda").view(1, 1, -1)
            maps = torch.stack([((bi * H + 2 * yy + a) * W + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)])
            ops.gather_conv(pairs, cin, packed, None, 1, n, 4 * cout, None, None, None, True, out=dst, out_row_map=maps.to(torch.int32).contiguous(),
                            out_col_group=cout, dense=True, math="f16x2", in_pairs=True, out_pairs=True)
    assert log.counts == {"tile_conv_f16p_kernel<128,128>": 1}, log.counts
    assert bool(torch.isnan(cat[:, :256]).all())                               # nothing outside the block is touched
    got = ops.pairs_to_rows(dst.contiguous())
    np.testing.assert_allclose(rows_nchw(got, batch, H, W), want, atol=1e-4, rtol=0)


def test_dense_pair_shapes_the_kernels_do_not_take_are_refused(hip):
    """nothing else reads dense pair rows: a shape neither pair kernel takes raises instead of computing on the wrong format; a small
    128-column layer (too few rows for the window tiles) goes to the pair tile kernel through its pixel table"""
    from cpd_amd._lib import CpdHipError
    x = torch.zeros((4 * 50 * 50, 128), device="cuda")
    nbr, _, _ = ops.rulebook_conv2d(4, 50, 50, 3, 3, 1, 1, "cuda")
    packed = ops.pack_weight(torch.zeros((9, 128, 128), device="cuda"))
    with ops.launch_log() as log:
        ops.gather_conv(x, 128, packed, nbr, 9, x.shape[0], 128, dense=True, math="f16x2", in_pairs=True, out_pairs=True)
    assert log.counts == {"tile_conv_f16p_kernel<128,128>": 1}, log.counts
    packed64 = ops.pack_weight(torch.zeros((9, 128, 64), device="cuda"))
    with pytest.raises(CpdHipError):
        ops.gather_conv(x, 128, packed64, nbr, 9, x.shape[0], 64, dense=True, math="f16x2", in_pairs=True, out_pairs=True)   # 64 columns, too few rows for the 256-row window tile
    with pytest.raises(CpdHipError):
        ops.gather_conv(x, 128, packed, nbr, 9, x.shape[0], 128, dense=True, math="f32", in_pairs=True, out_pairs=True)
