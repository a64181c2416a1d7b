"""Training kernels (config 3) against plain torch fp32 CPU references: BatchNorm statistics /
backward vs torch autograd, conv weight and input gradients vs autograd through the gather-matmul
form of the same convolution, transposed rulebooks vs their definition, Adam vs a numpy step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cpd_amd import ops, train_ops as T

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def random_sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize("n,c", [(5000, 16), (33333, 128), (2000, 320), (70000, 64)])
def test_bn_stats_and_backward_match_torch(hip, n, c):
    rng = np.random.default_rng(n + c)
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 2, c) + rng.normal(size=c)).astype(np.float32)
    res = rng.normal(size=(n, c)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32); beta = rng.normal(size=c).astype(np.float32)
    dy = rng.normal(size=(n, c)).astype(np.float32)
    eps = 1e-3
    # torch reference: y = relu(bn(x) + res)
    xt = torch.tensor(x, requires_grad=True); rt = torch.tensor(res, requires_grad=True)
    gt = torch.tensor(gamma, requires_grad=True); bt = torch.tensor(beta, requires_grad=True)
    yt = F.relu(F.batch_norm(xt, None, None, gt, bt, training=True, eps=eps) + rt)
    yt.backward(torch.tensor(dy))
    # HIP
    xd = dev(x)
    s1, s2 = T.bn_stats(xd)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0)
    np.testing.assert_allclose(mean.cpu().numpy(), x.mean(0), atol=2e-5)
    np.testing.assert_allclose(var.cpu().numpy(), x.var(0), rtol=2e-4, atol=1e-5)
    invstd = torch.rsqrt(var + eps)
    scale = dev(gamma) * invstd
    shift = dev(beta) - mean * scale
    y = T.affine_rows(xd, scale, shift, dev(res), True)
    np.testing.assert_allclose(y.cpu().numpy(), yt.detach().numpy(), atol=2e-4)
    dx, dgamma, dbeta, dres = T.bn_backward(dev(dy), y, xd, mean, invstd, dev(gamma), want_dres=True)
    np.testing.assert_allclose(dres.cpu().numpy(), rt.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(dbeta.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(dgamma.cpu().numpy(), gt.grad.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), atol=2e-4)
    np.testing.assert_allclose(T.col_sum(dev(dy)).cpu().numpy(), dy.sum(0), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(T.relu_backward(dev(dy), y).cpu().numpy(), dy * (y.cpu().numpy() > 0), atol=0)


def gather_matmul(x, w_kio, nbr):
    """Autograd-capable torch form of cpd_gather_conv: sum_t x[nbr[t]] @ W[t] (masked)."""
    out = 0
    for t in range(nbr.shape[0]):
        idx = torch.from_numpy(nbr[t].astype(np.int64))
        g = x[idx.clamp_min(0)] * (idx >= 0).float()[:, None]
        out = out + g @ w_kio[t]
    return out


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (32, 64), (64, 64), (128, 128), (64, 11)])
def test_subm_conv_gradients_match_autograd(oracle, hip, cin, cout):
    rng = np.random.default_rng(cin * 7 + cout)
    batch, shape = 2, [7, 24, 24]
    idx = random_sites(rng, batch, shape, 2500)
    n = idx.shape[0]
    nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    dy = rng.normal(size=(n, cout)).astype(np.float32)
    xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True)
    gather_matmul(xt, wt, nbr).backward(torch.tensor(dy))
    nbr_d = dev(nbr)
    dw = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n)
    np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=2e-3)
    # input gradient: same table, taps flipped, weights transposed (SubM symmetry)
    wd = torch.from_numpy(w).flip(0).transpose(1, 2).contiguous().cuda()          # [27, cout, cin]
    dx = ops.gather_conv(dev(dy), cout, ops.pack_weight(wd), nbr_d, 27, n, cin)
    np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), atol=2e-4)
    # accumulate flag
    dw2 = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n, dw=dw.clone(), accumulate=True)
    np.testing.assert_allclose(dw2.cpu().numpy(), 2 * wt.grad.numpy(), rtol=1e-4, atol=4e-3)


def wgrad_float64(x, dy, nbr):
    out = np.zeros((nbr.shape[0], x.shape[1], dy.shape[1]))
    for t in range(nbr.shape[0]):
        j = np.nonzero(nbr[t] >= 0)[0]
        out[t] = x[nbr[t, j]].astype(np.float64).T @ dy[j].astype(np.float64)
    return out


@pytest.mark.parametrize("cin,cout,sites", [(32, 32, 2500), (32, 64, 2500), (64, 128, 2500), (128, 128, 9000), (128, 32, 2500),
                                            (96, 160, 2500), (256, 64, 700)])
def test_wgrad_split_bf16_is_fp32_equivalent(oracle, hip, cin, cout, sites):
    """The split-bf16 weight gradient (flags CPD_GC_BF16X3): compacted pair ring, k-major staging, six bf16 products.
    Against float64 it must be as accurate as the fp32-MFMA kernel (both accumulate in fp32)."""
    rng = np.random.default_rng(cin + 3 * cout)
    batch, shape = 2, [7, 40, 40]
    idx = random_sites(rng, batch, shape, sites)
    n = idx.shape[0]
    nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    nbr[5] = -1                                     # a tap nobody has: its gradient is exactly zero
    x = (rng.normal(size=(n, cin)) * np.exp(rng.normal(size=(n, 1)) * 2)).astype(np.float32)     # wide dynamic range
    dy = rng.normal(size=(n, cout)).astype(np.float32)
    want = wgrad_float64(x, dy, nbr)
    nbr_d = dev(nbr)
    got = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n, bf16x3=True).cpu().numpy()
    ref32 = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n).cpu().numpy()
    scale = np.abs(want).max()
    err, err32 = np.abs(got - want).max() / scale, np.abs(ref32 - want).max() / scale
    assert err <= max(2.0 * err32, 2e-7), (err, err32)
    assert not got[5].any()
    # accumulate + determinism (fixed-order chunk sum: bitwise repeatable)
    again = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n, bf16x3=True).cpu().numpy()
    np.testing.assert_array_equal(got, again)
    acc = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr_d, 27, n, dw=torch.from_numpy(got).cuda(), accumulate=True, bf16x3=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * got, rtol=1e-6, atol=1e-6 * scale)


def test_wgrad_split_bf16_dense_and_1x1(hip):
    """Dense pixel tables (borders only lack a tap) and the table-free 1x1 case, strided row pitch."""
    rng = np.random.default_rng(12)
    b, h, w, cin, cout = 2, 30, 33, 64, 128
    nbr, _, _ = ops.rulebook_conv2d(b, h, w, 3, 3, 1, 1, "cuda")
    n = b * h * w
    buf = torch.from_numpy(rng.normal(size=(n, cin + 32)).astype(np.float32)).cuda()
    x = buf[:, 32:]                                                              # row pitch != channel count
    dy = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).cuda()
    want = wgrad_float64(x.cpu().numpy(), dy.cpu().numpy(), nbr.cpu().numpy())
    got = T.conv_wgrad(x, cin, dy, cout, nbr, 9, n, bf16x3=True).cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 5e-7
    one = T.conv_wgrad(x, cin, dy, cout, None, 1, n, bf16x3=True).cpu().numpy()
    want1 = x.cpu().numpy().astype(np.float64).T @ dy.cpu().numpy().astype(np.float64)
    assert np.abs(one[0] - want1).max() / np.abs(want1).max() < 5e-7


def test_strided_sparse_conv_gradients(oracle, hip):
    rng = np.random.default_rng(5)
    batch, shape, cin, cout = 2, [9, 20, 22], 32, 64
    k, s, p = [3, 3, 3], [2, 2, 2], [1, 1, 1]
    idx = random_sites(rng, batch, shape, 2000)
    d_idx = dev(idx)
    out_idx, out_index, out_shape = ops.conv_outset(d_idx, batch, shape, k, s, p)
    index = ops.SiteIndex.build(d_idx, batch, shape)
    nbr = ops.rulebook_conv(out_idx, index, k, s, p)
    nbr_np = nbr.cpu().numpy()
    n_in, n_out = idx.shape[0], out_idx.shape[0]
    # transposed table against its definition
    nbr_t = T.rulebook_conv_transpose(d_idx, batch, shape, k, s, p, out_index).cpu().numpy()
    want_t = np.full((27, n_in), -1, np.int32)
    tt, jj = np.nonzero(nbr_np >= 0)
    want_t[tt, nbr_np[tt, jj]] = jj
    np.testing.assert_array_equal(nbr_t, want_t)
    # ... and over an output level whose rows were re-ordered after its index was built (tap-pattern order, cpd_index_set_order):
    # the transposed table names the NEW rows, like the forward table built over the re-ordered site list
    new_idx, n2o, o2n = ops.order_rows_by_taps(out_idx, out_index, chunk_rows=1024)
    out_index.set_order(o2n)
    nbr_p = ops.rulebook_conv(new_idx, index, k, s, p).cpu().numpy()
    np.testing.assert_array_equal(nbr_p, nbr_np[:, n2o.cpu().numpy()])
    nbr_tp = T.rulebook_conv_transpose(d_idx, batch, shape, k, s, p, out_index).cpu().numpy()
    want_tp = np.full((27, n_in), -1, np.int32)
    tt, jj = np.nonzero(nbr_p >= 0)
    want_tp[tt, nbr_p[tt, jj]] = jj
    np.testing.assert_array_equal(nbr_tp, want_tp)
    assert not np.array_equal(nbr_tp, nbr_t)
    out_index.set_order(None)
    x = rng.normal(size=(n_in, cin)).astype(np.float32)
    w = (rng.normal(size=(27, cin, cout)) * 0.05).astype(np.float32)
    dy = rng.normal(size=(n_out, cout)).astype(np.float32)
    xt = torch.tensor(x, requires_grad=True); wt = torch.tensor(w, requires_grad=True)
    gather_matmul(xt, wt, nbr_np).backward(torch.tensor(dy))
    dw = T.conv_wgrad(dev(x), cin, dev(dy), cout, nbr, 27, n_out)
    np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=2e-3)
    wd = torch.from_numpy(w).transpose(1, 2).contiguous().cuda()                   # no flip: transposed table
    dx = ops.gather_conv(dev(dy), cout, ops.pack_weight(wd), dev(nbr_t), 27, n_in, cin)
    np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), atol=2e-4)


def test_conv2d_and_deconv_gradients_match_torch(hip):
    """Dense BEV convs: stride 1 (flip), stride 2 (transposed pixel table), ConvTranspose2d(k=s=2)."""
    rng = np.random.default_rng(8)
    b, cin, cout, h, w = 2, 32, 64, 12, 14
    x = rng.normal(size=(b, cin, h, w)).astype(np.float32)
    rows = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1).reshape(-1, a.shape[1]))).cuda()
    nchw = lambda r, hh, ww: r.cpu().numpy().reshape(b, hh, ww, -1).transpose(0, 3, 1, 2)
    for stride in (1, 2):
        wt = (rng.normal(size=(cout, cin, 3, 3)) * 0.1).astype(np.float32)
        xt = torch.tensor(x, requires_grad=True); wtt = torch.tensor(wt, requires_grad=True)
        y = F.conv2d(xt, wtt, None, stride, 1)
        ho, wo = y.shape[2:]
        dy = rng.normal(size=y.shape).astype(np.float32)
        y.backward(torch.tensor(dy))
        nbr, _, _ = ops.rulebook_conv2d(b, h, w, 3, 3, stride, 1, "cuda")
        w_kio = torch.from_numpy(wt).permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()
        dw = T.conv_wgrad(rows(x), cin, rows(dy), cout, nbr, 9, b * ho * wo)
        want_dw = wtt.grad.permute(2, 3, 1, 0).reshape(9, cin, cout).numpy()
        np.testing.assert_allclose(dw.cpu().numpy(), want_dw, rtol=1e-4, atol=2e-3)
        if stride == 1:
            wd, tab = w_kio.flip(0).transpose(1, 2).contiguous().cuda(), nbr
        else:
            wd, tab = w_kio.transpose(1, 2).contiguous().cuda(), T.rulebook_conv2d_transpose(b, h, w, 3, 3, 2, 1, "cuda")
        dx = ops.gather_conv(rows(dy), cout, ops.pack_weight(wd), tab, 9, b * h * w, cin)
        np.testing.assert_allclose(nchw(dx, h, w), xt.grad.numpy(), atol=2e-4)
    # ConvTranspose2d(k=s=2): forward = 1x1 GEMM + scatter; dgrad = 4-tap gather through the same row maps
    wd = (rng.normal(size=(cin, cout, 2, 2)) * 0.1).astype(np.float32)
    xt = torch.tensor(x, requires_grad=True); wdt = torch.tensor(wd, requires_grad=True)
    y = F.conv_transpose2d(xt, wdt, None, 2)
    dy = rng.normal(size=y.shape).astype(np.float32)
    y.backward(torch.tensor(dy))
    H, W = 2 * h, 2 * w
    bi = torch.arange(b, device="cuda").view(-1, 1, 1); yy = torch.arange(h, device="cuda").view(1, -1, 1)
    xx = torch.arange(w, device="cuda").view(1, 1, -1)
    maps = torch.stack([((bi * H + 2 * yy + a) * W + 2 * xx + c).reshape(-1) for a in range(2) for c in range(2)]).to(torch.int32).contiguous()
    w_t = torch.from_numpy(wd).permute(2, 3, 1, 0).reshape(4, cout, cin).contiguous().cuda()      # [tap][co][ci]
    dx = ops.gather_conv(rows(dy), cout, ops.pack_weight(w_t), maps, 4, b * h * w, cin)
    np.testing.assert_allclose(nchw(dx, h, w), xt.grad.numpy(), atol=2e-4)
    # dW[ci][(a,b,co)] = sum_pix x[pix][ci] * dy[(2y+a,2x+b)][co]  == wgrad with in/out roles swapped per tap
    dwt = T.conv_wgrad(rows(dy), cout, rows(x), cin, maps, 4, b * h * w)          # [tap][co][ci]
    want = wdt.grad.permute(2, 3, 1, 0).reshape(4, cout, cin).numpy()
    np.testing.assert_allclose(dwt.cpu().numpy(), want, rtol=1e-4, atol=2e-3)


def test_adam_step(hip):
    rng = np.random.default_rng(2)
    n = 100003
    p = rng.normal(size=n).astype(np.float32); g = rng.normal(size=n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pd, md, vd = dev(p), dev(m), dev(v)
    lr, b1, b2, eps, wd = 3e-3, 0.9, 0.99, 1e-8, 1e-2
    for step in (1, 2, 3):
        T.adam_step(pd, dev(g), md, vd, lr, b1, b2, eps, wd, step, grad_scale=0.5)
        gs = g * 0.5
        m = b1 * m + (1 - b1) * gs; v = b2 * v + (1 - b2) * gs * gs
        p = p - lr * wd * p
        p = p - lr * (m / (1 - b1 ** step)) / (np.sqrt(v / (1 - b2 ** step)) + eps)
    np.testing.assert_allclose(pd.cpu().numpy(), p, rtol=1e-5, atol=1e-6)


def test_bn_stats_finalize_matches_torch_batchnorm(hip):
    """Fused stats + finalize: (mean, invstd, scale, shift) and the running-stat update of
    torch.nn.BatchNorm1d in training mode (momentum 0.01, eps 1e-3: spconv_backbone.py:410)."""
    from cpd_amd import train_ops
    torch.manual_seed(0)
    for n, c, ld in [(70001, 16, 16), (35344, 128, 512), (501, 320, 320), (9000, 3, 16)]:
        buf = torch.randn(n, ld, device="cuda") * 2 + 0.5
        x = buf[:, :c]
        bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        rm, rv = bn.running_mean.clone(), bn.running_var.clone()
        y_ref = bn(x.contiguous())
        mean, invstd, scale, shift = train_ops.bn_stats_finalize(x, 1e-3, 0.01, bn.weight.detach(), bn.bias.detach(), rm, rv)
        y = train_ops.affine_rows(x, scale, shift)
        torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(rm, bn.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, bn.running_var, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(mean, x.double().mean(0).float(), rtol=1e-5, atol=1e-5)


def test_adjoint_packing_gives_input_gradient(hip):
    """cpd_gather_conv on cpd_pack_weight_adjoint(W, flip) with the SAME SubM rulebook is the
    input gradient of the forward conv (autograd of the gather-matmul definition)."""
    from cpd_amd import ops, train_ops
    torch.manual_seed(1)
    g = torch.Generator().manual_seed(3)
    coords = torch.unique(torch.stack([torch.zeros(4000, dtype=torch.int32), *(torch.randint(0, s, (4000,), generator=g,
                          dtype=torch.int32) for s in (9, 40, 40))], 1), dim=0).cuda()
    shape = [9, 40, 40]
    index = ops.SiteIndex.build(coords, 1, shape)
    nbr = ops.rulebook_subm(coords, index)
    n, cin, cout = coords.shape[0], 32, 48
    x = torch.randn(n, cin, device="cuda", requires_grad=True)
    w = torch.randn(27, cin, cout, device="cuda") * 0.1
    idx = torch.where(nbr < 0, n, nbr).long()
    xp = torch.cat([x, x.new_zeros(1, cin)])
    y = sum(xp[idx[t]] @ w[t] for t in range(27))
    dy = torch.randn_like(y)
    y.backward(dy)
    dx = ops.gather_conv(dy, cout, train_ops.pack_weight_adjoint(w, True), nbr, 27, n, cin)
    torch.testing.assert_close(dx, x.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("batch,n_obj", [(2, 30), (1, 0), (3, 120)])
def test_fused_center_loss_matches_the_torch_restatement(hip, batch, n_obj):
    """cpd_center_loss (loss parts + gradient in three launches) against cpd_amd.center_loss (torch autograd; itself
    pinned on goldens produced by the reference's FocalLossCenterNet / RegLossCenterNet, tests/test_center_loss.py),
    including: no objects at all (the num_pos == 0 branch), several objects on one pixel, saturated logits (clamped
    sigmoid -> zero gradient). A NaN target gives a NaN loss in both (the reference multiplies NaN by its zero mask)."""
    from cpd_amd import center_loss
    rng = np.random.default_rng(batch * 100 + n_obj)
    h, w, nc, ld, hm_col, K = 47, 52, 3, 16, 8, 150
    n = batch * h * w
    rows = torch.from_numpy(rng.normal(size=(n, ld)).astype(np.float32) * 2.0).cuda()
    rows[::97, hm_col] = 15.0                       # sigmoid above 1 - 1e-4: clamped, gradient 0
    rows[5::89, hm_col + 1] = -15.0                 # below 1e-4
    heat = torch.from_numpy((rng.random((batch, nc, h, w)) ** 6).astype(np.float32)).cuda() * 0.999
    tgt = torch.zeros((batch, K, 8), dtype=torch.float32, device="cuda")
    inds = torch.zeros((batch, K), dtype=torch.int64, device="cuda")
    masks = torch.zeros((batch, K), dtype=torch.int64, device="cuda")
    if n_obj:
        tgt[:, :n_obj] = torch.from_numpy(rng.normal(size=(batch, n_obj, 8)).astype(np.float32)).cuda()
        inds[:, :n_obj] = torch.from_numpy(rng.integers(0, h * w, size=(batch, n_obj))).cuda()
        inds[:, 1] = inds[:, 0]                     # two objects on one pixel: gradients add up
        masks[:, :n_obj] = 1
        masks[:, 2] = 0
        b_i = torch.arange(batch, device="cuda")[:, None].expand(batch, n_obj)
        cls = torch.from_numpy(rng.integers(0, nc, size=(batch, n_obj))).cuda()
        heat.view(batch, nc, h * w)[b_i, cls, inds[:, :n_obj]] = 1.0          # the peaks
    cw = [1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 2.0, 2.0]
    leaf = rows.clone().requires_grad_(True)
    want, parts = center_loss.center_head_loss(leaf, batch, h, w, heat, tgt, inds, masks, nc, hm_col=hm_col, code_weights=cw)
    want.backward()
    losses, d_rows = T.center_loss(rows, batch, h * w, nc, hm_col, heat, tgt, inds, masks, cw)
    got = losses.cpu().numpy()
    np.testing.assert_allclose(got[0], float(want.detach()), rtol=2e-6)
    np.testing.assert_allclose(got[1], float(parts["hm_loss"]), rtol=2e-6)
    np.testing.assert_allclose(got[2], float(parts["loc_loss"]), rtol=2e-6, atol=1e-12)
    g = leaf.grad
    scale = g.abs().max().item()
    assert (d_rows - g).abs().max().item() <= 2e-6 * scale, ((d_rows - g).abs().max().item(), scale)
    assert not d_rows[:, hm_col + nc:].any()        # padding columns carry no gradient
    again_l, again_d = T.center_loss(rows, batch, h * w, nc, hm_col, heat, tgt, inds, masks, cw)
    assert torch.equal(again_l, losses) and torch.equal(again_d, d_rows)        # deterministic
    if n_obj:
        tgt[0, 3, 4] = float("nan")
        nan_want, _ = center_loss.center_head_loss(rows, batch, h, w, heat, tgt, inds, masks, nc, hm_col=hm_col, code_weights=cw)
        nan_got, _ = T.center_loss(rows, batch, h * w, nc, hm_col, heat, tgt, inds, masks, cw)
        assert torch.isnan(nan_want) and torch.isnan(nan_got[0]) and torch.isnan(nan_got[2]) and not torch.isnan(nan_got[1])


def _absmax_word(t):
    """What cpd_bn_bwd_apply leaves behind for its dx: an absmax block whose largest word is the bits of max |t|."""
    return T.absmax_block(t)


@pytest.mark.parametrize("mag", [1e-12, 1e-7, 1e-3, 1.0, 1e4])
def test_scaled_split_fp16_gradient_convs(oracle, hip, monkeypatch, mag):
    """Gradients through the split-fp16 kernels (cpd_gather_conv_scaled / cpd_conv3x3_rows_scaled / cpd_conv_wgrad_scaled):
    dz of ANY magnitude -- far below fp16's range here -- pre-scaled by the power of two derived from its max |dz| word.
    Input gradient (row-wave, tile and window kernels) and weight gradient against float64, to the accuracy of the
    split-bf16 kernels they replace in the train step."""
    monkeypatch.setenv("CPD_TUNE", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN64", "1")
    rng = np.random.default_rng(7)
    # --- sparse: SubM 64 -> 64 over clustered sites
    batch, shape, cin, cout = 2, [9, 48, 48], 64, 64
    idx = random_sites(rng, batch, shape, 9000)
    n = idx.shape[0]
    d_idx = dev(idx)
    nbr = ops.rulebook_subm(d_idx, ops.SiteIndex.build(d_idx, batch, shape))
    x = (rng.normal(size=(n, cin)) * np.exp(rng.normal(size=(n, 1)))).astype(np.float32)
    dz = (rng.normal(size=(n, cout)) * np.exp(rng.normal(size=(n, 1)) * 2) * mag).astype(np.float32)    # wide dynamic range
    w = (rng.normal(size=(27, cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    d_x, d_dz, d_w = dev(x), dev(dz), dev(w)
    am = _absmax_word(d_dz)
    nbr_h = nbr.cpu().numpy()
    want_dw = wgrad_float64(x, dz, nbr_h)
    got = T.conv_wgrad(d_x, cin, d_dz, cout, nbr, 27, n, math="f16x2", dy_absmax=am).cpu().numpy()
    ref = T.conv_wgrad(d_x, cin, d_dz, cout, nbr, 27, n, math="bf16x3").cpu().numpy()
    s = np.abs(want_dw).max()
    e16, e3 = np.abs(got - want_dw).max() / s, np.abs(ref - want_dw).max() / s
    assert e16 <= max(2.0 * e3, 3e-7), (e16, e3)
    # roles swapped (ConvTranspose weight gradient: the GRADIENT is the gathered operand)
    got_t = T.conv_wgrad(d_dz, cout, d_x, cin, nbr, 27, n, math="f16x2", in_absmax=am).cpu().numpy()
    want_t = wgrad_float64(dz, x, nbr_h)
    assert np.abs(got_t - want_t).max() / np.abs(want_t).max() <= max(2.0 * e3, 3e-7)
    # input gradient = gather_conv on the adjoint weights, same rulebook
    pw_adj = T.pack_weight_adjoint(d_w, flip_taps=True)
    name = ops.gather_conv_tile(n, cout, cin, cout, dense=False, math="f16x2")
    assert name.startswith("rowwave_conv_f16_kernel"), name
    dx16 = ops.gather_conv(d_dz, cout, pw_adj, nbr, 27, n, cin, math="f16x2", in_absmax=am).double().cpu().numpy()
    dx3 = ops.gather_conv(d_dz, cout, pw_adj, nbr, 27, n, cin, math="bf16x3").double().cpu().numpy()
    idx_t = np.where(nbr_h < 0, n, nbr_h)
    dzp = np.concatenate([dz.astype(np.float64), np.zeros((1, cout))])
    want_dx = sum(dzp[idx_t[t]] @ w[26 - t].astype(np.float64).T for t in range(27))
    s = np.abs(want_dx).max()
    e16, e3 = np.abs(dx16 - want_dx).max() / s, np.abs(dx3 - want_dx).max() / s
    assert e16 <= max(2.0 * e3, 3e-7), (e16, e3)
    # without the word the same call loses the small gradients (what kept the gradient convs on split-bf16 before)
    if mag <= 1e-12:
        raw = ops.gather_conv(d_dz, cout, pw_adj, nbr, 27, n, cin, math="f16x2").double().cpu().numpy()
        assert np.abs(raw - want_dx).max() / s > 1e-3
    # --- dense 3x3 128 -> 128: window kernel (pixel table with the geometry tag) and tile kernel (plain table)
    b, h, wd, c = 2, 40, 44, 128
    nb2, _, _ = ops.rulebook_conv2d(b, h, wd, 3, 3, 1, 1, "cuda")
    n2 = b * h * wd
    dz2 = torch.from_numpy((rng.normal(size=(n2, c)) * np.exp(rng.normal(size=(n2, 1)) * 2) * mag).astype(np.float32)).cuda()
    w2 = torch.from_numpy((rng.normal(size=(9, c, c)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)).cuda()
    pw2 = T.pack_weight_adjoint(w2, flip_taps=True)
    am2 = _absmax_word(dz2)
    assert ops.gather_conv_tile(n2, c, c, c, dense=True, math="f16x2", nbr=nb2).startswith("window_conv_f16_kernel")
    plain = nb2.clone()
    assert ops.gather_conv_tile(n2, c, c, c, dense=True, math="f16x2", nbr=plain).startswith("tile_conv_f16_kernel")
    rows = torch.cat([torch.tensor([0, 1, wd - 1, wd, h * wd - 1, h * wd, n2 - 1], device="cuda"), torch.randint(0, n2, (1500,), device="cuda")])
    ii = nb2[:, rows].long()
    dzp2 = torch.cat([dz2, dz2.new_zeros(1, c)]).double()
    want2 = sum(dzp2[torch.where(ii[t] < 0, n2, ii[t])] @ w2[8 - t].double().T for t in range(9))
    s2 = want2.abs().max().item()
    for table in (nb2, plain):
        got2 = ops.gather_conv(dz2, c, pw2, table, 9, n2, c, dense=True, math="f16x2", in_absmax=am2)
        ref2 = ops.gather_conv(dz2, c, pw2, table, 9, n2, c, dense=True, math="bf16x3")
        e16 = (got2[rows].double() - want2).abs().max().item() / s2
        e3 = (ref2[rows].double() - want2).abs().max().item() / s2
        assert e16 <= max(2.0 * e3, 3e-7), (e16, e3)


def test_bn_backward_leaves_the_absmax_word(hip):
    torch.manual_seed(5)
    n, c = 30000, 96
    x = torch.randn(n, c, device="cuda") * 3 + 1
    dy = torch.randn(n, c, device="cuda") * 1e-6
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = (var + 1e-3).rsqrt()
    gamma = torch.rand(c, device="cuda") + 0.5
    y = torch.relu((x - mean) * invstd * gamma)
    am = torch.zeros(T.ABSMAX_WORDS, dtype=torch.int32, device="cuda")
    dx, _, _, _ = T.bn_backward(dy, y, x, mean, invstd, gamma, dx_absmax=am)
    words = am.view(16, 32)
    assert not words[:, 1:].any()                                   # one word per 128-byte line, nothing else touched
    assert words[:, 0].max().view(torch.float32).item() == dx.abs().max().item()


def test_batched_weight_packing_matches_the_per_tensor_calls(hip):
    """cpd_pack_batch_run (three launches for every image of a model) against cpd_pack_weight / cpd_pack_weight_adjoint,
    bit for bit: channel counts with and without split images, flipped and unflipped adjoints, 1x1 and 27-tap kernels."""
    torch.manual_seed(11)
    shapes = [(27, 5, 16), (27, 16, 16), (27, 32, 64), (9, 128, 128), (1, 128, 256), (9, 64, 3), (27, 64, 64), (4, 256, 96)]
    flat = torch.randn(sum(k * a * b for k, a, b in shapes), device="cuda") * 0.1
    jobs, want, off = [], [], 0
    for k, a, b in shapes:
        w = flat[off:off + k * a * b].view(k, a, b)
        off += k * a * b
        for adjoint, flip in ((False, False), (True, True), (True, False)):
            packed = torch.full((T.packed_floats(k, b if adjoint else a, a if adjoint else b),), float("nan"), device="cuda")
            jobs.append((w, packed, adjoint, flip))
            want.append(T.pack_weight_adjoint(w, flip) if adjoint else ops.pack_weight(w))
    batch = T.PackBatch(jobs)
    batch.run(3)
    for (w, packed, adjoint, flip), ref in zip(jobs, want):
        assert torch.equal(packed.view(torch.int32), ref.view(torch.int32)), (tuple(w.shape), adjoint, flip)
    flat.mul_(1.7)                                    # new weights, same table
    batch.run(3)
    for (w, packed, adjoint, flip) in jobs:
        ref = T.pack_weight_adjoint(w, flip) if adjoint else ops.pack_weight(w)
        assert torch.equal(packed.view(torch.int32), ref.view(torch.int32))


def test_scaled_gradient_convs_do_not_hide_nan_or_overflow(hip, monkeypatch):
    """A NaN (or inf) in dz must reach the input gradient, and a maximum word that is too small (stale) must show as inf / NaN,
    never as a silently wrong finite number: the contract of the pre-scaled split-fp16 path."""
    monkeypatch.setenv("CPD_TUNE", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN64", "1")
    torch.manual_seed(3)
    b, h, w, c = 1, 24, 24, 64
    nbr, _, _ = ops.rulebook_conv2d(b, h, w, 3, 3, 1, 1, "cuda")
    n = b * h * w
    dz = torch.randn(n, c, device="cuda") * 1e-6
    wgt = torch.randn(9, c, c, device="cuda") * 0.05
    pw = T.pack_weight_adjoint(wgt, flip_taps=True)
    x = torch.randn(n, c, device="cuda")
    # NaN in the gradient: BatchNorm backward would leave a NaN maximum -> scaling off, NaN propagates
    bad = dz.clone()
    bad[100, 7] = float("nan")
    am = torch.zeros(T.ABSMAX_WORDS, dtype=torch.int32, device="cuda")
    am[0] = 0x7fc00000
    dx = ops.gather_conv(bad, c, pw, nbr, 9, n, c, dense=True, math="f16x2", in_absmax=am)
    assert torch.isnan(dx).any()
    dw = T.conv_wgrad(x, c, bad, c, nbr, 9, n, math="f16x2", dy_absmax=am)
    assert torch.isnan(dw).any()
    # a maximum word 2^20 too small: the scaled values leave fp16's range -> inf / NaN in the result, not a finite lie
    stale = T.absmax_block(dz * 2.0 ** -20)
    dx = ops.gather_conv(dz, c, pw, nbr, 9, n, c, dense=True, math="f16x2", in_absmax=stale)
    assert not torch.isfinite(dx).all()
    # and the honest word gives the float64 answer
    good = T.absmax_block(dz)
    dx = ops.gather_conv(dz, c, pw, nbr, 9, n, c, dense=True, math="f16x2", in_absmax=good)
    idx = torch.where(nbr < 0, n, nbr).long()
    dzp = torch.cat([dz, dz.new_zeros(1, c)]).double()
    want = sum(dzp[idx[t]] @ wgt[8 - t].double().T for t in range(9))
    assert (dx.double() - want).abs().max().item() <= 1e-6 * want.abs().max().item()


def _targets_args():
    return ((188, 188), [-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15], 3)


def test_center_targets_kernel_matches_the_reference_golden(hip, golden):
    """cpd_center_targets against what the reference's own CenterHead.assign_target_of_single_head produced (tests/golden/center_loss.npz,
    tools of make_golden.py): heat map to float rounding of the CPU's exp / division, masks and indices exactly, targets to 1e-5."""
    g = golden("center_loss")
    gt = torch.from_numpy(g["gt_boxes"])[None].cuda()
    heat, tgt, inds, masks = T.center_targets(gt, *_targets_args(), feature_map_stride=8, num_max_objs=500, gaussian_overlap=0.1, min_radius=2)
    np.testing.assert_allclose(heat[0].cpu().numpy(), g["heatmap"], atol=1e-6)
    np.testing.assert_array_equal(masks[0].cpu().numpy(), g["mask"])
    np.testing.assert_array_equal(inds[0].cpu().numpy(), g["inds"])
    np.testing.assert_allclose(tgt[0].cpu().numpy(), g["ret_boxes"], atol=1e-5)


@pytest.mark.parametrize("batch,m,k", [(1, 40, 500), (3, 120, 50), (2, 0, 16), (2, 7, 500), (1, 700, 500)])
def test_center_targets_kernel_matches_the_torch_restatement(hip, batch, m, k):
    """... and against cpd_amd.center_loss.assign_targets (itself pinned on the golden above) on random boxes: padding rows interleaved
    with boxes, more boxes than slots (the head's boxes are compacted BEFORE the truncation), degenerate boxes (dx <= 0), boxes on the
    map's border and outside it (patches clipped), several boxes on one pixel and class (max-merge), no box at all, class ids above
    num_classes (clamped like the restatement), an odd map size (the clear kernel's byte tail). Masks / indices / radii exactly. The
    kernel divides like the reference on the CPU; torch on the GPU multiplies by the reciprocal of a scalar divisor, which moves a
    pixel coordinate by up to two ulps (1.5e-5 at 100 pixels: the fractional-offset targets agree to that), so the heat maps are
    compared to 1e-6 and in their sets of peaks."""
    from cpd_amd import center_loss
    rng = np.random.default_rng(1000 * batch + m + k)
    gt = np.zeros((batch, m, 8), np.float32)
    if m:
        gt[..., 0:2] = rng.uniform(-80, 80, (batch, m, 2))
        gt[..., 2] = rng.uniform(-1, 3, (batch, m))
        gt[..., 3:6] = rng.uniform(0.3, 12.0, (batch, m, 3))
        gt[..., 6] = rng.uniform(-4, 4, (batch, m))
        gt[..., 7] = rng.integers(1, 4, (batch, m))
        gt[:, ::5] = 0                                    # padding rows in between
        if m > 6:
            gt[:, 1, 3] = 0.0                             # degenerate box: valid = False, zeros everywhere
            gt[:, 2, 7] = 7                               # a class id above num_classes
            gt[:, 3, :2] = gt[:, 4, :2]                   # two boxes on one pixel
            gt[:, 3, 7] = gt[:, 4, 7]
            gt[:, 6, :2] = [75.15, -75.19]                # a corner pixel: the patch is clipped on two sides
    dev_gt = torch.from_numpy(gt).cuda()
    for hw in ((188, 188), (47, 51)):
        args = (hw,) + _targets_args()[1:]
        kw = dict(feature_map_stride=8 if hw[0] == 188 else 32, num_max_objs=k, gaussian_overlap=0.1, min_radius=2)
        want = center_loss.assign_targets(dev_gt, *args, **kw)
        got = T.center_targets(dev_gt, *args, **kw)
        assert torch.equal(got[3], want[3]) and torch.equal(got[2], want[2])
        torch.testing.assert_close(got[1], want[1], rtol=0, atol=4e-5)      # (x - int x at x ~ 100: two ulps of the coordinate = 1.5e-5)
        torch.testing.assert_close(got[0], want[0], rtol=0, atol=1e-6)
        assert torch.equal(got[0] == 1.0, want[0] == 1.0)
        again = T.center_targets(dev_gt, *args, **kw)
        assert all(torch.equal(a, b) for a, b in zip(got, again))     # deterministic (max-merge is order-independent)
