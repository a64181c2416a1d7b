"""Config 3 (train step) parity: the HIP trainer's hand-written forward/backward against the torch-CPU
float64 autograd graph of the reference modules (tests/ref_train_torch.py) on a reduced geometry, plus
properties of the optimiser step. Tolerances: loss 1e-4 relative; each gradient tensor, relative to
its own max magnitude, within max(2e-3, 3 x the error a torch-fp32 run of the same reference graph
makes) of the float64 reference -- BatchNorm bias gradients are sums with heavy cancellation, where
fp32 (torch's or ours) is ~1e-2 off float64; conv biases in front of a BatchNorm have an exactly zero
gradient, checked absolutely. Every ReLU of the reference is pinned to the branch the HIP forward took
(see ref_train_torch: a handful of the ~2e7 activations sit within fp32 rounding of the kink), after
checking that the two forwards disagree on fewer than 1e-5 of the branches."""
import numpy as np
import pytest
import torch

from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud

import ref_train_torch

pytestmark = pytest.mark.gpu


def small_cfg():
    return ModelConfig(point_cloud_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], post_center_limit_range=[-20, -20, -2, 20, 20, 4],
                       bev_num_filters=[64, 128], bev_num_upsample_filters=[128, 128], bev_layer_nums=[2, 2],
                       max_obj_per_sample=100)


def scene(batch=2, n_gt=8, seed=0):
    pts = [waymo_cloud(b, n_points=30000 - 5000 * b) for b in range(batch)]
    for p in pts:
        p[:, :2] *= 0.3
    rng = np.random.default_rng(seed)
    gt = np.zeros((batch, n_gt + 2, 8), np.float32)                       # two zero rows = collate padding
    for b in range(batch):
        for i in range(n_gt):
            gt[b, i] = [rng.uniform(-18, 18), rng.uniform(-18, 18), rng.uniform(0, 1), 4.5 * rng.uniform(0.9, 1.1),
                        2.0 * rng.uniform(0.9, 1.1), 1.6, rng.uniform(-3, 3), rng.integers(1, 4)]
    return pts, gt


def hip_relu_masks(tr, cfg, batch, hw):
    """(layer name -> y > 0) of the trainer's last forward, in the reference's layouts."""
    h, w = hw
    masks = {}
    sc = cfg.shared_conv_channel
    for c in tr.layers:
        if not (c.relu and c.has_bn) or c.saved is None:
            continue
        m = (c.saved[4] > 0).cpu()
        if c.saved[8]:                                        # dense rows [B*H*W, C] -> NCHW
            hh = h if m.shape[0] == batch * h * w else h // 2
            m = m.view(batch, hh, m.shape[0] // (batch * hh), m.shape[1]).permute(0, 3, 1, 2)
        if c.name == "dense_head.heads.first":
            for hi, n in enumerate(cfg.head_names()):
                masks["dense_head.heads_list.0.%s.0.0" % n] = m[:, hi * sc:(hi + 1) * sc]
        else:
            masks[c.name] = m
    return masks


@pytest.fixture(scope="module")
def trainer_and_ref(oracle, hip):
    from cpd_amd.train_engine import CenterPointTrainer
    cfg = small_cfg()
    sd = init_state_dict(cfg, seed=3)
    pts, gt = scene()
    tr = CenterPointTrainer(cfg, sd, num_max_objs=50)
    sd0 = tr.state_dict()
    rows = tr.forward([torch.from_numpy(p).cuda() for p in pts])
    masks = hip_relu_masks(tr, cfg, len(pts), tr.tape["hw"])
    loss, d_rows, parts = tr.loss(rows, torch.from_numpy(gt).cuda())
    tr.backward(d_rows)
    grads = {k: v.cpu() for k, v in tr.grad_dict().items()}
    # the un-pinned float64 forward must take the same ReLU branches almost everywhere
    taps = {}
    P = ref_train_torch.make_leaves(sd)
    free_loss, _, _ = ref_train_torch.forward_loss(oracle, cfg, P, pts, gt, num_max_objs=50, taps=taps)
    flips = sum(int(((y > 0) != masks[k]).sum()) for k, (_, y) in taps.items())
    total = sum(y.numel() for _, y in taps.values())
    assert flips <= 1e-5 * total, (flips, total)
    assert abs(float(free_loss) - float(loss)) <= 1e-4 * abs(float(free_loss))
    P = ref_train_torch.make_leaves(sd)
    ref_loss, ref_parts, _ = ref_train_torch.forward_loss(oracle, cfg, P, pts, gt, num_max_objs=50, masks=masks)
    ref_loss.backward()
    P32 = ref_train_torch.make_leaves(sd, torch.float32)
    ref_train_torch.forward_loss(oracle, cfg, P32, pts, gt, num_max_objs=50, masks=masks)[0].backward()
    return dict(P32=P32, cfg=cfg, sd=sd, sd0=sd0, tr=tr, loss=float(loss), parts=parts, grads=grads, P=P, ref_loss=float(ref_loss),
                ref_parts=ref_parts, pts=pts, gt=gt)


def test_state_dict_round_trip(trainer_and_ref):
    """Flat-buffer layouts <-> reference names/layouts lose nothing."""
    sd, sd0 = trainer_and_ref["sd"], trainer_and_ref["sd0"]
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        assert k in sd0, k
        np.testing.assert_array_equal(sd0[k].cpu().numpy(), v.float().numpy(), err_msg=k)


def test_loss_matches_reference(trainer_and_ref):
    t = trainer_and_ref
    assert abs(t["loss"] - t["ref_loss"]) <= 1e-4 * abs(t["ref_loss"])
    for k in ("hm_loss", "loc_loss"):
        assert abs(float(t["parts"][k]) - float(t["ref_parts"][k])) <= 1e-4 * abs(float(t["ref_parts"][k])) + 1e-6


def test_gradients_match_reference(trainer_and_ref):
    t = trainer_and_ref
    worst = []
    for k, leaf in t["P"].items():
        ref = leaf.grad.numpy()
        got = t["grads"][k].double().numpy()
        assert got.shape == ref.shape, k
        if np.abs(ref).max() < 1e-9:                       # bias in front of a batch-stat BatchNorm
            assert np.abs(got).max() <= 1e-3, k
            continue
        scale = np.abs(ref).max()
        err = np.abs(got - ref).max() / scale
        err32 = np.abs(t["P32"][k].grad.double().numpy() - ref).max() / scale
        worst.append((err / max(2e-3, 3 * err32), err, err32, k))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:8]


def test_running_stats_follow_batchnorm_semantics(trainer_and_ref, oracle):
    """running = (1-m)*running + m*batch (unbiased var), m = 0.01, for the first sparse BatchNorm."""
    t = trainer_and_ref
    cfg, sd = t["cfg"], t["sd"]
    feats, coords = ref_train_torch.voxelize_batch(oracle, cfg, t["pts"])
    nbr = oracle.subm_rulebook(coords, len(t["pts"]), cfg.sparse_shape, [3, 3, 3])
    z = ref_train_torch._gconv(torch.as_tensor(feats, dtype=torch.float64), sd["backbone_3d.conv_input.0.weight"].double(), nbr)
    mean, var = z.mean(0), z.var(0, unbiased=True)
    new = t["tr"].state_dict()
    want_m = 0.99 * sd["backbone_3d.conv_input.1.running_mean"].double() + 0.01 * mean
    want_v = 0.99 * sd["backbone_3d.conv_input.1.running_var"].double() + 0.01 * var
    np.testing.assert_allclose(new["backbone_3d.conv_input.1.running_mean"].cpu().double().numpy(), want_m.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(new["backbone_3d.conv_input.1.running_var"].cpu().double().numpy(), want_v.numpy(), rtol=1e-5, atol=1e-6)


def test_optimizer_step_matches_torch_adam(trainer_and_ref):
    """One cpd_adam_step on the flat buffer == decoupled-weight-decay Adam on the same gradient
    (fastai_optim.py:132-150: p -= lr*wd*p, then Adam), with the grad-norm clip of train_utils.py:43."""
    tr = trainer_and_ref["tr"]
    st = tr.store
    p0, g = st.flat.clone(), st.grad.clone()
    tr.total_steps = None
    tr.optimizer_step()
    norm = float(g.norm())
    gs = g * min(1.0, tr.grad_clip / (norm + 1e-6)) if norm > tr.grad_clip else g
    b1, b2 = tr.betas
    m = (1 - b1) * gs
    v = (1 - b2) * gs * gs
    want = p0 * (1 - tr.lr * tr.weight_decay) - tr.lr * (m / (1 - b1)) / ((v / (1 - b2)).sqrt() + 1e-8)
    np.testing.assert_allclose(st.flat.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-7)


def test_training_reduces_loss_and_exports_to_engine(hip):
    from cpd_amd.train_engine import CenterPointTrainer
    cfg = small_cfg()
    pts, gt = scene(seed=1)
    dev_pts = [torch.from_numpy(p).cuda() for p in pts]
    dev_gt = torch.from_numpy(gt).cuda()
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=5), lr=1e-3, num_max_objs=50)
    losses = [float(tr.step(dev_pts, dev_gt)[0]) for _ in range(12)]
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.6 * losses[0], losses
    eng = CenterPointEngine(cfg, {k: v.cpu() for k, v in tr.state_dict().items()})
    res = eng.forward(dev_pts)
    assert len(res) == 2 and all(torch.isfinite(r["pred_boxes"]).all() for r in res)


@pytest.mark.parametrize("conv_math,grad_math", [("f16x2", "f16x2"), ("f16x2", "bf16x3"), ("bf16x3", None), ("f32", None)])
def test_every_layer_backward_is_locally_exact(hip, monkeypatch, conv_math, grad_math):
    """Inside a real step: each layer's BatchNorm backward sums, weight gradient and input gradient
    against float64 torch formulas evaluated on that layer's own saved tensors (no ReLU-kink or
    error-propagation effects): 1e-4 of the tensor's max magnitude. In every arithmetic the trainer offers: split-fp16
    forward with scaled split-fp16 gradients (the default) or split-bf16 gradients, split-bf16 throughout, fp32 MFMA."""
    from cpd_amd.train_engine import CenterPointTrainer, _Conv
    if grad_math is not None:
        monkeypatch.setenv("CPD_TRAIN_GRAD_MATH", grad_math)
    monkeypatch.setenv("CPD_TUNE", "1")              # the split kernels also for this small scene (they are sized for full frames)
    monkeypatch.setenv("CPD_GC_BF16_MIN", "1")
    monkeypatch.setenv("CPD_GC_BF16_MIN64", "1")
    cfg = small_cfg()
    cfg.conv_math = conv_math
    pts, gt = scene(seed=2)
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=7), num_max_objs=50)
    assert tr.store.grad_math == (grad_math or ("f16x2" if conv_math == "f16x2" else "bf16x3"))
    orig = _Conv.backward
    report = []

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-6))

    def checked(self, dy, nbr_adj, n_in, need_dx=True, add=None, dx_out=None):
        x, nbr, n_out, z, y, mean, invstd, has_res, dense, up_map = self.saved
        st = self.store
        dyc = dy.clone()
        res = orig(self, dy, nbr_adj, n_in, need_dx, add, dx_out)
        torch.cuda.synchronize()                  # the weight gradient runs on the trainer's second stream
        if not self.has_bn or (self.mode == "up" and self.up > 1):
            return res
        g = dyc.double() * (y > 0) if self.relu else dyc.double()
        xh = (z.double() - mean.double()) * invstd.double()
        dbeta, dgamma = g.sum(0), (g * xh).sum(0)
        n = z.shape[0]
        dz = st.p(self.gn).double() * invstd.double() * (g - dbeta / n - xh * dgamma / n)
        tbl = nbr if nbr is not None else torch.arange(n_out, device=z.device, dtype=torch.int32)[None]
        idx = torch.where(tbl < 0, x.shape[0], tbl).long()
        xp = torch.cat([x[:, :self.c_in].double(), x.new_zeros(1, self.c_in).double()])
        dw = torch.stack([xp[idx[t]].T @ dz for t in range(self.kv)])
        errs = [rel(st.g(self.be), dbeta), rel(st.g(self.gn), dgamma), rel(st.g(self.wn), dw)]
        if need_dx and res[0] is not None:
            w = st.p(self.wn).double()
            tbl = nbr_adj if nbr_adj is not None else torch.arange(n_in, device=z.device, dtype=torch.int32)[None]
            idx = torch.where(tbl < 0, n, tbl).long()
            dzp = torch.cat([dz, dz.new_zeros(1, self.c_out)])
            dx = dz.new_zeros(n_in, self.c_in)
            for t in range(self.kv):
                dx += dzp[idx[t]] @ (w[self.kv - 1 - t] if self.mode == "same" else w[t]).T
            if add is not None:
                dx += add.double()
            errs.append(rel(res[0], dx))
        report.append((max(errs), self.name, errs))
        return res

    _Conv.backward = checked
    try:
        tr.forward_backward([torch.from_numpy(p).cuda() for p in pts], torch.from_numpy(gt).cuda())
    finally:
        _Conv.backward = orig
    assert len(report) >= 30
    report.sort(reverse=True)
    assert report[0][0] <= 1e-4, report[:5]


def test_full_size_config3_step_is_locally_exact_on_the_benchmarked_kernels(hip):
    """VERDICT r2 next #2a: ONE CenterPointTrainer step on the FULL configuration -- ModelConfig() defaults, a 160k-point cloud,
    full BEV widths, no CPD_TUNE -- with the per-layer check of test_every_layer_backward_is_locally_exact (every layer's
    BatchNorm sums, weight gradient and input gradient against float64 formulas on its own saved tensors, 1e-4 of the tensor's
    maximum), and the kernel instantiations that carry the measured step (profiles/r02_train_kernel_stats.csv) asserted through
    the library's launch log. float64 checks run on the GPU (the dense layers are 35344 x 9 x 256 x 256)."""
    from cpd_amd import ops
    from cpd_amd.synthetic import gt_boxes
    from cpd_amd.train_engine import CenterPointTrainer, _Conv
    cfg = ModelConfig()
    pts = [torch.from_numpy(waymo_cloud(3)).cuda()]
    gt = torch.from_numpy(gt_boxes(3)).cuda()[None]
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=7))
    assert tr.store.math == "f16x2" and tr.store.grad_math == "f16x2"
    orig = _Conv.backward
    report = []

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-6))

    def checked(self, dy, nbr_adj, n_in, need_dx=True, add=None, dx_out=None):
        x, nbr, n_out, z, y, mean, invstd, has_res, dense, up_map = self.saved
        st = self.store
        dyc = dy.clone()
        res = orig(self, dy, nbr_adj, n_in, need_dx, add, dx_out)
        torch.cuda.synchronize()
        if not self.has_bn or (self.mode == "up" and self.up > 1):
            return res
        g = dyc.double() * (y > 0) if self.relu else dyc.double()
        xh = (z.double() - mean.double()) * invstd.double()
        dbeta, dgamma = g.sum(0), (g * xh).sum(0)
        n = z.shape[0]
        dz = st.p(self.gn).double() * invstd.double() * (g - dbeta / n - xh * dgamma / n)
        del g, xh
        tbl = nbr if nbr is not None else torch.arange(n_out, device=z.device, dtype=torch.int32)[None]
        xp = torch.cat([x[:, :self.c_in].double(), x.new_zeros(1, self.c_in).double()])
        dw = []
        for t in range(self.kv):                        # one tap at a time: the gathered operand of a dense layer is 72 MB in float64
            idx = torch.where(tbl[t] < 0, x.shape[0], tbl[t]).long()
            dw.append(xp[idx].T @ dz)
        errs = [rel(st.g(self.be), dbeta), rel(st.g(self.gn), dgamma), rel(st.g(self.wn), torch.stack(dw))]
        del dw, xp
        if need_dx and res[0] is not None:
            w = st.p(self.wn).double()
            tbl = nbr_adj if nbr_adj is not None else torch.arange(n_in, device=z.device, dtype=torch.int32)[None]
            dzp = torch.cat([dz, dz.new_zeros(1, self.c_out)])
            dx = dz.new_zeros(n_in, self.c_in)
            for t in range(self.kv):
                idx = torch.where(tbl[t] < 0, n, tbl[t]).long()
                dx += dzp[idx] @ (w[self.kv - 1 - t] if self.mode == "same" else w[t]).T
            if add is not None:
                dx += add.double()
            errs.append(rel(res[0], dx))
        report.append((max(errs), self.name, errs))
        return res

    _Conv.backward = checked
    try:
        with ops.launch_log() as log:
            tr.forward_backward(pts, gt)
            torch.cuda.synchronize()
    finally:
        _Conv.backward = orig
    assert len(report) >= 30
    report.sort(reverse=True)
    assert report[0][0] <= 1e-4, report[:5]
    # the instantiations of the measured train step (one frame, full widths): weight gradients, input-gradient and forward convs
    for name in ("wgrad_f16_kernel<128,128>", "wgrad_f16_kernel<64,64>", "wgrad_f16_kernel<32,32>", "tile_conv_f16s_kernel<64,128>",
                 "tile_conv_f16_kernel<64,128>", "rowwave_conv_f16s_kernel<64,2>", "rowwave_conv_f16_kernel<64,2>",
                 "rowwave_conv_f16s_kernel<32,2>", "rowwave_conv_f16_kernel<32,2>", "rowwave_conv_f16s_kernel<128,1>",
                 "rowwave_conv_f16_kernel<128,1>",
                 # the one-frame 3 x 3 BEV / head layers on 64-column window tiles (round 3's small-batch tile rule)
                 "window_conv_f16_kernel<64,128>", "window_conv_f16s_kernel<64,128>", "window_conv_f16_kernel<16,128>"):
        # (a row-wave layer large enough to run unsplit takes the LDS-epilogue form of its kernel: f16e / f16se, round 4)
        assert log.counts.get(name, 0) + log.counts.get(name.replace("_kernel<", "e_kernel<"), 0) > 0, (name, sorted(log.counts))


def test_step_is_bitwise_the_same_with_and_without_the_side_streams(hip):
    """Round 5 moved the index chain of the strided stages to its own HIP stream (a stage ahead of the forward layers) and the target
    assignment off the main stream. Neither changes a kernel or an argument: three optimiser steps with the streams on give, bit for
    bit, the parameters and the losses of three steps with everything on the main stream (index_side_stream = False) -- a missing
    event (a conv reading a rulebook still being built) or a block the allocator recycled across streams would show up here; the torch
    target path on its side stream (fused_targets = False) against the same path in line likewise."""
    from cpd_amd.train_engine import CenterPointTrainer
    cfg = small_cfg()
    scenes = [scene(seed=s) for s in (1, 2, 3)]

    def run(index_stream, fused, early):
        tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=5), lr=1e-3, num_max_objs=50)
        tr.index_side_stream, tr.fused_targets, tr.early_targets = index_stream, fused, early
        losses = []
        for pts, gt in scenes:
            loss, _ = tr.step([torch.from_numpy(p).cuda() for p in pts], torch.from_numpy(gt).cuda())
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, tr.store.flat.clone()

    for fused in (True, False):
        l0, p0 = run(False, fused, False)
        l1, p1 = run(True, fused, True)
        assert l0 == l1, (fused, l0, l1)
        assert torch.equal(p0, p1), "parameters differ after three steps (fused targets: %s)" % fused
