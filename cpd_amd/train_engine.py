"""Train step of the hot path (BASELINE config 3): forward with batch-statistics BatchNorm, backward,
gradient all-reduce, Adam -- one process per GPU, frames sharded by rank.

Mirrors one iteration of tools/train_utils/train_utils.py:20-60 for the one-stage CenterPoint graph
(MeanVFE -> VoxelResBackBone8x -> HeightCompression -> BaseBEVBackbone -> CenterHead.get_loss), see
SURVEY Appendix B:
  * every conv, forward and backward, is the rulebook kernel `cpd_gather_conv`: the input gradient
    is the same kernel on the adjoint weights (tap-flipped for SubM / stride-1, with the transposed
    rulebook for strided convs), the weight gradient is `cpd_conv_wgrad`;
  * BatchNorm uses batch statistics (`cpd_bn_stats` / `cpd_bn_finalize` / `cpd_affine_rows`) and its
    backward is `cpd_bn_bwd_*`; the SparseBasicBlock residual is handled inside those kernels;
  * all parameters, gradients and Adam moments live in four flat fp32 buffers, so the data-parallel
    exchange is ONE RCCL all-reduce (DistributedDataParallel's role at tools/train.py:143) and the
    optimiser ONE launch (`cpd_adam_step`: Adam(betas=(0.9, 0.99)) with decoupled weight decay and a
    OneCycle schedule -- optimization/__init__.py:19-53, fastai_optim.py:132-150);
  * target assignment and the focal / L1 losses are cpd_amd.center_loss (torch on the device).
There is no autograd graph: each layer object keeps what its backward needs from the forward.
"""
import math
import os
from typing import Dict, List

import torch

from . import _lib, center_loss, dist_utils, ops, train_ops
from .engine import _DOWN, ModelConfig


class _Flat:
    """Named fp32 parameters packed into one buffer (plus gradient and Adam moment buffers)."""

    def __init__(self):
        self._pending = []
        self.slots = {}
        self.flat = self.grad = self.m = self.v = None
        self.math = "f16x2"     # forward conv arithmetic of layers with >= 32 input channels (ModelConfig.conv_math)
        # gradients span many orders of magnitude (1e-3 ... 1e-8), below fp16's normal range. Everything that consumes dz -- the
        # input gradient (cpd_gather_conv on the adjoint weights) and the weight gradient -- therefore either runs split-bf16
        # (exact split at any magnitude, six products), or -- grad_math "f16x2" -- split-fp16 (three products) with dz
        # PRE-SCALED by a power of two: the BatchNorm backward that produces dz also leaves the bits of max |dz| in a device
        # block of words (`absmax`, one block per layer, zeroed at the start of every backward pass) and the consuming kernels derive
        # the scale from it (cpd_gather_conv_scaled / cpd_conv_wgrad_scaled). Layers whose dz does not come out of a
        # BatchNorm backward (the head's output convs, <= 3 channels) are not split-arithmetic layers anyway.
        self.grad_math = "f16x2"
        self.dgrad_math = "bf16x3"      # arithmetic of gradient convs WITHOUT an absmax word
        self.bf16x3 = True
        self.n_absmax = 0
        self.absmax = None
        self.update_stats = True    # whether the current forward updates the BatchNorm running statistics
        self.side = None        # HIP stream for the weight gradients (independent of the input gradients of the same layer)

    def mark(self):
        """An event at the current point of the current stream (None without a side stream)."""
        if self.side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def on_side(self, fn, *tensors, after=None):
        """Run fn() on the side stream, ordered after `after` (an event from mark(); default: everything issued so far
        on the current stream); `tensors` are the current-stream tensors it reads (kept alive for the side stream).
        Without a side stream: fn() in place."""
        if self.side is None:
            return fn()
        ev = after if after is not None else self.mark()
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            return fn()

    def join_side(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def add(self, name, t):
        assert name not in self.slots
        t = t.detach().to(torch.float32).contiguous()
        self.slots[name] = None
        self._pending.append((name, t))
        return name

    def finalize(self, device):
        off = 0
        for name, t in self._pending:
            self.slots[name] = (off, tuple(t.shape))
            off += (t.numel() + 3) // 4 * 4               # 16-byte aligned slots
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        for name, t in self._pending:
            self.p(name).copy_(t)
        self.grad = torch.zeros_like(self.flat)
        self.absmax = torch.zeros(max(self.n_absmax, 1) * train_ops.ABSMAX_WORDS, dtype=torch.int32, device=device)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self._pending = None

    def _view(self, buf, name):
        off, shape = self.slots[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def p(self, name):
        return self._view(self.flat, name)

    def g(self, name):
        return self._view(self.grad, name)


class _Conv:
    """conv (+bias) (+ batch-stat BatchNorm) (+ residual) (+ ReLU), parameters in the flat store.

    `mode`: 'same' = SubM / stride-1 conv (adjoint = same rulebook, flipped taps);
            'strided' = adjoint uses a transposed rulebook passed to backward();
            'up' = ConvTranspose2d(k = s = u) as one 1x1 GEMM with column-group scatter."""

    def __init__(self, store, name, w_kio, bias, bn, eps, momentum, relu, mode="same", up=1):
        self.store, self.name = store, name
        self.kv, self.c_in, self.c_out = w_kio.shape
        self.mode, self.up = mode, up
        self.relu = relu
        self.eps, self.momentum = eps, momentum
        self.wn = store.add(name + ".w", w_kio)
        self.bn_ = store.add(name + ".b", bias) if bias is not None else None
        self.has_bn = bn is not None
        if self.has_bn:
            self.gn = store.add(name + ".gamma", bn["weight"])
            self.be = store.add(name + ".beta", bn["bias"])
            self.running_mean = bn["running_mean"].clone().float()
            self.running_var = bn["running_var"].clone().float()
        self.c_bn = self.c_out // (up * up)                # channels after the column-group scatter
        self.am_slot = store.n_absmax                      # this layer's max |dz| word (see _Flat.grad_math)
        store.n_absmax += 1
        self.pw = self.pw_adj = None
        self.saved = None

    def to(self, device):
        if self.has_bn:
            self.running_mean = self.running_mean.to(device).contiguous()
            self.running_var = self.running_var.to(device).contiguous()

    def repack(self):
        """Forward and adjoint packed images of the current weights (after every optimiser step)."""
        w = self.store.p(self.wn)
        # the images are rewritten in place (no allocation per step)
        self.pw = ops.pack_weight(w, out=getattr(self, "pw", None))
        if self.mode == "up" and self.up > 1:
            u2 = self.up * self.up
            w_t = w.view(self.c_in, u2, self.c_bn).permute(1, 2, 0).contiguous()     # [tap, co, ci]
            self.pw_adj = ops.pack_weight(w_t, out=getattr(self, "pw_adj", None))
        else:
            self.pw_adj = train_ops.pack_weight_adjoint(w, flip_taps=(self.mode == "same"), out=getattr(self, "pw_adj", None))

    # ---------------------------------------------------------------- forward
    def forward(self, x, nbr, n_out, residual=None, dense=False, out=None, up_map=None, n_up=None, update_stats=None):
        st = self.store
        if update_stats is None:                              # the trainer's per-forward switch (CenterPointTrainer.forward)
            update_stats = st.update_stats
        bias = st.p(self.bn_) if self.bn_ else None
        if not self.has_bn:                                   # final head convs: conv + bias only
            y = ops.gather_conv(x, self.c_in, self.pw, nbr, self.kv, n_out, self.c_out, None, bias, None, self.relu,
                                out=out, dense=dense, math=self.store.math)
            self.saved = (x, nbr, n_out, None, y, None, None, False, dense, None)
            return y
        if self.mode == "up" and self.up > 1:
            z = torch.empty((n_up, self.c_bn), dtype=torch.float32, device=x.device)
            ops.gather_conv(x, self.c_in, self.pw, None, 1, n_out, self.c_out, None, None, None, False, out=z,
                            out_row_map=up_map, out_col_group=self.c_bn, dense=dense, math=self.store.math)
        else:
            z = ops.gather_conv(x, self.c_in, self.pw, nbr, self.kv, n_out, self.c_out, None, bias, None, False,
                                dense=dense, math=self.store.math)
        self.n_total = None
        if getattr(st, "sync_bn", False):
            # SyncBatchNorm (tools/train.py:32,117 --sync_bn): statistics over every rank's rows (train_ops.bn_stats_finalize_sync)
            mean, invstd, scale, shift, self.n_total = train_ops.bn_stats_finalize_sync(
                z, self.eps, self.momentum, st.p(self.gn), st.p(self.be),
                self.running_mean if update_stats else None, self.running_var if update_stats else None, group=st.group)
        else:
            mean, invstd, scale, shift = train_ops.bn_stats_finalize(
                z, self.eps, self.momentum, st.p(self.gn), st.p(self.be),
                self.running_mean if update_stats else None, self.running_var if update_stats else None)
        y = train_ops.affine_rows(z, scale, shift, residual, self.relu, out=out)
        self.saved = (x, nbr, n_out, z, y, mean, invstd, residual is not None, dense, up_map)
        return y

    # ---------------------------------------------------------------- backward
    def backward(self, dy, nbr_adj, n_in, need_dx=True, add=None, dx_out=None):
        """dy -> (dx, d_residual). `add` is summed into dx (second branch of a fork)."""
        st = self.store
        x, nbr, n_out, z, y, mean, invstd, has_res, dense, up_map = self.saved
        self.saved = None
        dres = None
        am = None                                 # device word with the bits of max |dz|: the gradient convs may run split-fp16
        gmath, wmath = st.dgrad_math, ("bf16x3" if st.bf16x3 else "f32")
        if self.has_bn:
            if st.grad_math == "f16x2" and st.math != "f32":
                am = st.absmax[self.am_slot * train_ops.ABSMAX_WORDS:(self.am_slot + 1) * train_ops.ABSMAX_WORDS]
                gmath = wmath = "f16x2"
            dz, _, _, dres = train_ops.bn_backward(dy, y if self.relu else None, z, mean, invstd, st.p(self.gn),
                                                   want_dres=has_res, dgamma=st.g(self.gn), dbeta=st.g(self.be), dx_absmax=am,
                                                   sync=(self.n_total, st.group) if getattr(self, "n_total", None) is not None else None)
        else:
            dz = train_ops.relu_backward(dy, y) if self.relu else dy
        gw = st.g(self.wn)
        gb = st.g(self.bn_) if self.bn_ else None              # bias gradient = column sums of dz: rides with the weight gradient
        if gb is not None and os.environ.get("CPD_TRAIN_BIAS_SIDE", "1") == "0":
            train_ops.col_sum(dz, out=gb)
            gb = None
        if self.mode == "up" and self.up > 1:
            u2 = self.up * self.up
            # dW[tap][co][ci] = sum_pix dz_up[map[tap][pix]][co] * x[pix][ci]  (roles of in/dy swapped)
            def wgrad_up():
                if gb is not None:
                    train_ops.col_sum(dz, out=gb)
                tmp = train_ops.conv_wgrad(dz, self.c_bn, x, self.c_in, up_map, u2, n_out, math=wmath, in_absmax=am)
                gw.view(self.c_in, u2, self.c_bn).copy_(tmp.permute(2, 0, 1))
            ready = st.mark()                   # dz is complete here; the input gradient is ISSUED first (it is on the
            dx = None                           # critical path), the weight gradient after it but ordered on `ready`
            if need_dx:
                dx = ops.gather_conv(dz, self.c_bn, self.pw_adj, up_map, u2, n_out, self.c_in, None, None, add, False,
                                     out=dx_out, dense=dense, math=gmath, in_absmax=am)
            st.on_side(wgrad_up, dz, x, after=ready)
            return dx, dres
        if nbr is None:                                       # 1x1 conv: identity rulebook for the weight gradient
            nbr_w = torch.arange(n_out, dtype=torch.int32, device=dz.device).view(1, -1)
        else:
            nbr_w = nbr
        # the weight gradient needs (x, dz), the input gradient (dz, W): independent, so they run on two streams and
        # share the chip (at one frame per GPU neither fills it)
        ready = st.mark()
        dx = None
        if need_dx:
            dx = ops.gather_conv(dz, self.c_out, self.pw_adj, nbr_adj, self.kv, n_in, self.c_in, None, None, add, False,
                                 out=dx_out, dense=dense, math=gmath, in_absmax=am)
        def wgrad():                                          # everything that only needs (x, dz): off the input-gradient chain
            if gb is not None:
                train_ops.col_sum(dz, out=gb)
            train_ops.conv_wgrad(x, self.c_in, dz, self.c_out, nbr_w, self.kv, n_out, dw=gw, math=wmath, dy_absmax=am)
        st.on_side(wgrad, x, dz, nbr_w, after=ready)
        return dx, dres


def one_cycle(step, total_steps, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """OneCycle of learning_schedules_fastai.py:57-82: cosine lr_max/div -> lr_max -> lr_max/(div*1e4),
    momentum moms[0] -> moms[1] -> moms[0]. Returns (lr, beta1) for `step` in [0, total_steps)."""
    def anneal(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    a1 = int(total_steps * pct_start)
    low = lr_max / div_factor
    if step < a1:
        pct = step / max(a1, 1)
        return anneal(low, lr_max, pct), anneal(moms[0], moms[1], pct)
    pct = (step - a1) / max(total_steps - a1, 1)
    return anneal(lr_max, low / 1e4, pct), anneal(moms[1], moms[0], pct)


class CenterPointTrainer:
    """One data-parallel replica: `step(points_list, gt_boxes)` runs forward, loss, backward, the
    gradient all-reduce over `process_group` (RCCL) and the Adam update; returns the loss dict."""

    SPARSE_BN = (1e-3, 0.01)   # spconv_backbone.py:410
    BEV_BN = (1e-3, 0.01)      # base_bev_backbone.py:38
    HEAD_BN = (1e-5, 0.1)      # nn.BatchNorm2d defaults, center_head.py:24,78

    def __init__(self, cfg: ModelConfig, state_dict: Dict[str, torch.Tensor], device="cuda", lr=3e-3, betas=(0.9, 0.99),
                 weight_decay=1e-5, grad_clip=32.0, total_steps=None, process_group=None, world_size=1,
                 num_max_objs=500, code_weights=None, sync_bn=False):
        """sync_bn: the reference's `--sync_bn` (tools/train.py:32,117: SyncBatchNorm.convert_sync_batchnorm; off by default there and
        here) -- every BatchNorm's batch statistics, and the two sums of its backward pass, are all-reduced over `process_group`; a step of
        N ranks x 1 frame then normalises like one process with the N frames in its batch."""
        self.cfg = cfg
        self.device = torch.device(device)
        self.lr, self.betas, self.weight_decay, self.grad_clip = lr, betas, weight_decay, grad_clip
        self.total_steps = total_steps
        self.pg, self.world = process_group, world_size
        self.num_max_objs = num_max_objs
        self.code_weights = code_weights
        self.fused_loss = True
        self.fused_targets = os.environ.get("CPD_TRAIN_FUSED_TARGETS", "1") != "0"    # cpd_center_targets instead of center_loss.assign_targets
        self.steps_done = 0
        self.store = _Flat()
        self.store.sync_bn = bool(sync_bn) and (world_size > 1 or bool(os.environ.get("CPD_FORCE_DIST")))
        self.store.group = process_group
        self.store.math = cfg.conv_math
        self.store.dgrad_math = "f32" if cfg.conv_math == "f32" else "bf16x3"
        self.store.bf16x3 = cfg.conv_math != "f32"
        self.store.grad_math = os.environ.get("CPD_TRAIN_GRAD_MATH", "f16x2" if cfg.conv_math == "f16x2" else "bf16x3")
        # which split images the step reads: the fp16 one (forward and -- with their max |dz| word -- gradient convs), the bf16
        # one only when some gradient conv still runs split-bf16. A layer without a BatchNorm behind it has no such word; with
        # everything else on fp16 those (the head's <= 3-channel output convs) take the fp32 kernels rather than keep a third
        # image of every weight fresh.
        self._pack_images = {"f32": 0, "bf16x3": 1, "f16x2": 2}[cfg.conv_math]
        if cfg.conv_math == "f16x2" and self.store.grad_math == "f16x2":
            self.store.dgrad_math, self.store.bf16x3 = "f32", False
        elif cfg.conv_math != "f32":
            self._pack_images |= 1
        if self.device.type == "cuda" and os.environ.get("CPD_TRAIN_SIDE_STREAM", "1") != "0":
            self.store.side = _lib.side_stream(self.device, "wgrad")
        # the strided stages' index chain on its own stream, one stage ahead of the forward convolutions (forward(); CPD_TRAIN_INDEX_STREAM=0:
        # every table first, on the main stream)
        self.index_side_stream = self.device.type == "cuda" and os.environ.get("CPD_TRAIN_INDEX_STREAM", "1") != "0"
        self.async_repack = os.environ.get("CPD_TRAIN_ASYNC_REPACK", "0") != "0"            # optimizer_step: repack on the side stream (measured: 9.60 -> 9.93 ms/step -- off)
        self._repack_ev = None
        self.early_targets = os.environ.get("CPD_TRAIN_EARLY_TARGETS", "1") != "0"      # forward_backward: targets_early()
        self._voxelizers = []
        self._bev_cache = {}
        self._build(state_dict)
        dense_first = self.store._pending[self._n_sparse_slots][0]
        self.store.finalize(self.device)
        # gradient buckets (dist_utils.BucketedReduce): [dense_offset, end) = BEV backbone + head, complete when the BEV half of the
        # backward pass is; [0, dense_offset) = sparse backbone. CPD_TRAIN_BUCKETS=0: one all-reduce after the whole backward (round 4)
        self.dense_offset = self.store.slots[dense_first][0]
        self.bucketed_reduce = os.environ.get("CPD_TRAIN_BUCKETS", "1") != "0"
        self._reduce = None
        for c in self.layers:
            c.to(self.device)
        # every packed image (forward + adjoint) of every layer is rebuilt after each optimiser step: three launches for all of
        # them (train_ops.PackBatch) instead of eight per layer; ConvTranspose(k = s > 1) layers, whose adjoint image is built
        # from a permuted copy of the weights, keep their own repack()
        jobs, self._repack_single = [], []
        for c in self.layers:
            if c.mode == "up" and c.up > 1:
                self._repack_single.append(c)
                continue
            w = self.store.p(c.wn)
            c.pw = torch.empty((train_ops.packed_floats(c.kv, c.c_in, c.c_out),), dtype=torch.float32, device=self.device)
            c.pw_adj = torch.empty((train_ops.packed_floats(c.kv, c.c_out, c.c_in),), dtype=torch.float32, device=self.device)
            jobs.append((w, c.pw, False, False))
            jobs.append((w, c.pw_adj, True, c.mode == "same"))
        self._pack = train_ops.PackBatch(jobs) if (jobs and self.device.type == "cuda") else None
        self._repack_all()

    # ------------------------------------------------------------------ graph
    @staticmethod
    def _bn(sd, name):
        return {k: sd[name + "." + k] for k in ("weight", "bias", "running_mean", "running_var")}

    @staticmethod
    def _sparse_w(sd, name):
        w = sd[name + ".weight"]                               # (Cout, kD, kH, kW, Cin)
        return w.reshape(w.shape[0], -1, w.shape[-1]).permute(1, 2, 0)

    @staticmethod
    def _w2d(w):
        return w.permute(2, 3, 1, 0).reshape(w.shape[2] * w.shape[3], w.shape[1], w.shape[0])

    def _final_depth(self):
        shape = self.cfg.sparse_shape
        for stage in ["conv2", "conv3", "conv4", "conv_out"]:
            k, s, pd = _DOWN[stage]
            shape = ops.conv_out_shape(shape, k, s, pd)
        return shape[0]

    def _build(self, sd):
        cfg, st = self.cfg, self.store
        self.layers: List[_Conv] = []

        def mk(name, w_kio, bias, bn_name, bnp, relu=True, mode="same", up=1):
            c = _Conv(st, name, w_kio, bias, self._bn(sd, bn_name) if bn_name else None, bnp[0], bnp[1], relu, mode, up)
            self.layers.append(c)
            return c

        p = "backbone_3d."
        S = {}
        S["conv_input"] = mk(p + "conv_input.0", self._sparse_w(sd, p + "conv_input.0"), None, p + "conv_input.1",
                             self.SPARSE_BN)

        def block(name):
            return (mk(name + ".conv1", self._sparse_w(sd, name + ".conv1"), sd.get(name + ".conv1.bias"), name + ".bn1",
                       self.SPARSE_BN),
                    mk(name + ".conv2", self._sparse_w(sd, name + ".conv2"), sd.get(name + ".conv2.bias"), name + ".bn2",
                       self.SPARSE_BN))

        S["conv1"] = [block(p + "conv1.0"), block(p + "conv1.1")]
        for stage in ["conv2", "conv3", "conv4"]:
            S[stage + ".down"] = mk(p + stage + ".0.0", self._sparse_w(sd, p + stage + ".0.0"), None, p + stage + ".0.1",
                                    self.SPARSE_BN, mode="strided")
            S[stage] = [block(p + stage + ".1"), block(p + stage + ".2")]
        S["conv_out"] = mk(p + "conv_out.0", self._sparse_w(sd, p + "conv_out.0"), None, p + "conv_out.1", self.SPARSE_BN,
                           mode="strided")
        self.sparse = S
        self._n_sparse_slots = len(st._pending)          # the flat store lays its slots out in creation order: [sparse | BEV | head]

        p = "backbone_2d."
        depth, C = self._final_depth(), cfg.out_features
        self.depth = depth
        self.bev_levels = []
        for lvl in range(len(cfg.bev_layer_nums)):
            convs = []
            names = ["blocks.%d.1" % lvl] + ["blocks.%d.%d" % (lvl, 4 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            bns = ["blocks.%d.2" % lvl] + ["blocks.%d.%d" % (lvl, 5 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            for i, (cn, bnn) in enumerate(zip(names, bns)):
                w = sd[p + cn + ".weight"]
                if lvl == 0 and i == 0:     # reference channel c*D+z (height_compression.py:136-138) -> ours z*C+c
                    w = w.reshape(w.shape[0], C, depth, 3, 3).permute(0, 2, 1, 3, 4).reshape(w.shape[0], depth * C, 3, 3)
                mode = "strided" if (i == 0 and cfg.bev_layer_strides[lvl] != 1) else "same"
                convs.append(mk(p + cn, self._w2d(w), None, p + bnn, self.BEV_BN, mode=mode))
            u = cfg.bev_upsample_strides[lvl]
            wd = sd[p + "deblocks.%d.0.weight" % lvl]                          # (Cin, Cout, u, u)
            w_kio = wd.permute(0, 2, 3, 1).reshape(1, wd.shape[0], u * u * wd.shape[1])
            de = mk(p + "deblocks.%d.0" % lvl, w_kio, None, p + "deblocks.%d.1" % lvl, self.BEV_BN, mode="up", up=u)
            self.bev_levels.append((convs, de, u, wd.shape[1]))

        p = "dense_head."
        self.shared = mk(p + "shared_conv.0", self._w2d(sd[p + "shared_conv.0.weight"]), sd.get(p + "shared_conv.0.bias"),
                         p + "shared_conv.1", self.HEAD_BN)
        # the five SeparateHead branches: first convs fused along Cout (BatchNorm is per channel, so
        # five BatchNorm2d(64) are one 320-channel one), second convs one small conv each
        names = cfg.head_names()
        sc = cfg.shared_conv_channel
        w1 = torch.cat([self._w2d(sd[p + "heads_list.0.%s.0.0.weight" % n]) for n in names], dim=2)
        b1 = torch.cat([sd[p + "heads_list.0.%s.0.0.bias" % n] for n in names])
        bn1 = {k: torch.cat([sd[p + "heads_list.0.%s.0.1.%s" % (n, k)] for n in names])
               for k in ("weight", "bias", "running_mean", "running_var")}
        self.head1 = _Conv(st, p + "heads.first", w1, b1, bn1, self.HEAD_BN[0], self.HEAD_BN[1], True)
        self.layers.append(self.head1)
        self.head2 = []
        self.head_slices = {}
        col = 0
        for hi, n in enumerate(names):
            q = p + "heads_list.0.%s.1" % n
            c = mk(q, self._w2d(sd[q + ".weight"]), sd[q + ".bias"], None, (0.0, 0.0), relu=False)
            self.head2.append((c, hi * sc, col))
            self.head_slices[n] = (col, c.c_out)
            col += c.c_out
        self.n_head_out = col
        self.head_ld = 16 * ((col + 15) // 16)

    # ------------------------------------------------------------------ tables
    def _bev_tables(self, batch, h, w):
        key = (batch, h, w)
        if key not in self._bev_cache:
            dev = self.device
            t = {}
            t["s1"] = ops.rulebook_conv2d(batch, h, w, 3, 3, 1, 1, dev)
            t["s2"] = ops.rulebook_conv2d(batch, h, w, 3, 3, 2, 1, dev)
            h2, w2 = t["s2"][1], t["s2"][2]
            t["s2_t"] = train_ops.rulebook_conv2d_transpose(batch, h, w, 3, 3, 2, 1, dev)
            t["s1_half"] = ops.rulebook_conv2d(batch, h2, w2, 3, 3, 1, 1, dev)
            b_i = torch.arange(batch, device=dev).view(-1, 1, 1)
            yy = torch.arange(h2, device=dev).view(1, -1, 1)
            xx = torch.arange(w2, device=dev).view(1, 1, -1)
            maps = [((b_i * h + 2 * yy + a) * w + 2 * xx + bb).reshape(-1) for a in range(2) for bb in range(2)]
            t["up2"] = torch.stack(maps).to(torch.int32).contiguous()
            self._bev_cache[key] = t
        return self._bev_cache[key]

    # ------------------------------------------------------------------ forward / backward
    def _voxelize(self, points_list):
        cfg = self.cfg
        while len(self._voxelizers) < len(points_list):
            self._voxelizers.append(ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, cfg.num_point_features,
                                                  cfg.max_points_per_voxel, cfg.max_voxels, device=self.device))
        return [self._voxelizers[b](pts, batch_idx=b, coord_cols=4, want_voxels=False, want_mean=True, sync=False)
                for b, pts in enumerate(points_list)]

    @staticmethod
    def _voxelize_finish(outs):
        ms = torch.cat([o[4] for o in outs]).tolist()          # the read-back of the voxel counts
        feats = torch.cat([o[3][:m] for o, m in zip(outs, ms)])
        coords = torch.cat([o[1][:m] for o, m in zip(outs, ms)])
        return feats, coords

    def _blocks_fwd(self, blocks, x, nbr):
        n = x.shape[0]
        for c1, c2 in blocks:
            y = c1.forward(x, nbr, n)
            x = c2.forward(y, nbr, n, residual=x)
        return x

    def _blocks_bwd(self, blocks, dy, nbr):
        n = dy.shape[0]
        for c1, c2 in reversed(blocks):
            d1, dres = c2.backward(dy, nbr, n)
            dy, _ = c1.backward(d1, nbr, n, add=dres)
        return dy

    def forward(self, points_list, update_stats=True):
        """Training-mode forward; returns head rows [B*H*W, head_ld] and keeps the tape. update_stats=False (gradient
        checks, evaluation passes on batch statistics) leaves every BatchNorm's running_mean / running_var untouched."""
        cfg, S = self.cfg, self.sparse
        self.store.update_stats = bool(update_stats)
        batch = len(points_list)
        feats, coords = self._voxelize_finish(self._voxelize(points_list))
        shape = cfg.sparse_shape
        # The rulebooks depend on coordinates only, and their output-set sizes are the step's host read-backs. Round 5: the index chain of
        # the strided stages (output set, both rulebooks, the transposed one for the backward pass) runs on its own HIP stream, one
        # stage ahead of the convolutions -- the read-backs wait for that stream only, so the conv chain is still queued without
        # bubbles, and the 25 small index launches of a step overlap layers that do not fill the chip at one frame per GPU.
        # (`index_side_stream=False`: the same order of calls on the main stream.) Tables are kept until the next step's
        # chain starts, and that start waits for everything queued on the main stream (caching-allocator safety across streams).
        stages = ["conv2", "conv3", "conv4", "conv_out"]
        n0 = coords.shape[0]
        side = None
        self._index_keep = keep = []

        def begin(stage, coords, shape):
            k, s, pd = _DOWN[stage]
            return ops.conv_outset_begin(coords, batch, shape, k, s, pd)

        if self.index_side_stream and coords.is_cuda:
            side = _lib.side_stream(self.device, "index")
            main = torch.cuda.current_stream(self.device)
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                pending = begin(stages[0], coords, shape)            # stage 2's output set is marked and counted beside the level-0 index build
        index = ops.SiteIndex.build(coords, batch, shape)
        ev_index = main.record_event() if side is not None else None

        def stage_tables(stage, coords, index, shape, pending, after=None):
            """a stage's output set (`pending`: its conv_outset_begin), the next stage's conv_outset_begin, its rulebooks; coords / index /
            shape = the stage's input level; `after`: an event of the main stream that `index` is complete at"""
            k, s, pd = _DOWN[stage]
            out_idx, out_index, out_shape = ops.conv_outset_end(pending)
            if after is not None:
                torch.cuda.current_stream(self.device).wait_event(after)
            nxt = begin(stages[stages.index(stage) + 1], out_idx, out_shape) if stage != "conv_out" else None
            nbr_dn = ops.rulebook_conv(out_idx, index, k, s, pd)
            nbr_dn_t = train_ops.rulebook_conv_transpose(coords, batch, shape, k, s, pd, out_index)
            nbr_sub = None if stage == "conv_out" else ops.rulebook_subm(out_idx, out_index)
            rows = None
            if stage == "conv_out":                          # HeightCompression's row map (used by the backward pass): coordinates only, too
                (d, h, w), ci = out_shape, out_idx.long()
                rows = ((ci[:, 0] * h + ci[:, 2]) * w + ci[:, 3]) * d + ci[:, 1]
            return (stage, nbr_dn, nbr_dn_t, nbr_sub, out_idx.shape[0], out_idx, out_index, out_shape, rows, nxt)

        def queue_tables(stage, coords, index, shape, pending, after=None):
            if side is None:
                return stage_tables(stage, coords, index, shape, pending), None
            with torch.cuda.stream(side):
                t = stage_tables(stage, coords, index, shape, pending, after)
                return t, side.record_event()

        nbr0 = ops.rulebook_subm(coords, index)
        keep.append((coords, index, nbr0))
        tape = {"nbr0": nbr0, "stages": []}
        if getattr(self, "_repack_ev", None) is not None:      # the previous optimiser step's weight images (optimizer_step)
            torch.cuda.current_stream(self.device).wait_event(self._repack_ev)
            self._repack_ev = None
        x = S["conv_input"].forward(feats, nbr0, n0)
        x = self._blocks_fwd(S["conv1"], x, nbr0)
        if side is None:
            pending = begin(stages[0], coords, shape)
        # a stage's chain is queued after the previous stage's layers (the count read-back inside can still block the host for a moment;
        # the main stream has its work by then), its output set having been marked a stage earlier
        nxt = queue_tables(stages[0], coords, index, shape, pending, after=ev_index)
        for si, stage in enumerate(stages):
            t, ev = nxt
            keep.append(t)
            if ev is not None:
                main.wait_event(ev)
            _, nbr_dn, nbr_dn_t, nbr_sub, n_stage = t[:5]
            n_in = x.shape[0]
            if stage == "conv_out":
                x = S["conv_out"].forward(x, nbr_dn, n_stage)
            else:
                x = S[stage + ".down"].forward(x, nbr_dn, n_stage)
                x = self._blocks_fwd(S[stage], x, nbr_sub)
            tape["stages"].append((stage, nbr_sub, nbr_dn_t, n_in))
            coords, index, shape = t[5], t[6], t[7]
            if si + 1 < len(stages):
                nxt = queue_tables(stages[si + 1], coords, index, shape, t[9])
        dense_rows = t[8]
        d, h, w = shape
        C = x.shape[1]
        dense = ops.densify_nhwc(x, coords, batch, shape).view(batch * h * w, d * C)
        tape["dense_rows"] = dense_rows
        tape["dense_shape"] = (batch, h, w, d, C)

        T = self._bev_tables(batch, h, w)
        n_full = batch * h * w
        c_cat = sum(cfg.bev_num_upsample_filters)
        cat = torch.empty((n_full, c_cat), dtype=torch.float32, device=self.device)
        x = dense
        col = 0
        cur = (h, w)
        lv_tape = []
        for lvl, (convs, de, u, c_up) in enumerate(self.bev_levels):
            stride = cfg.bev_layer_strides[lvl]
            if stride == 1:
                (nbr0, ho, wo), adj0 = (T["s1"] if cur == (h, w) else T["s1_half"]), None
            elif stride == 2 and cur == (h, w):
                (nbr0, ho, wo), adj0 = T["s2"], T["s2_t"]
            else:
                raise NotImplementedError("BEV stride pattern outside the shipped configs")
            n_in_lvl = x.shape[0]
            n_lvl = batch * ho * wo
            nbr_same = T["s1"][0] if (ho, wo) == (h, w) else T["s1_half"][0]
            x = convs[0].forward(x, nbr0, n_lvl, dense=True)
            for cv in convs[1:]:
                x = cv.forward(x, nbr_same, n_lvl, dense=True)
            cur = (ho, wo)
            dst = cat[:, col:col + c_up]
            if u == 1:
                de.forward(x, None, n_lvl, dense=True, out=dst)
            elif u == 2 and (ho * 2, wo * 2) == (h, w):
                de.forward(x, None, n_lvl, dense=True, out=dst, up_map=T["up2"], n_up=n_full)
            else:
                raise NotImplementedError("upsample stride outside the shipped configs")
            lv_tape.append((nbr0 if adj0 is None else adj0, nbr_same, n_in_lvl, n_lvl, col, c_up))
            col += c_up
        tape["levels"] = lv_tape
        s1 = T["s1"][0]
        s = self.shared.forward(cat, s1, n_full, dense=True)
        h1 = self.head1.forward(s, s1, n_full, dense=True)
        rows = torch.zeros((n_full, self.head_ld), dtype=torch.float32, device=self.device)
        sc = cfg.shared_conv_channel
        for c, in_col, out_col in self.head2:
            c.forward(h1[:, in_col:in_col + sc], s1, n_full, out=rows[:, out_col:out_col + c.c_out])
        tape.update(s1=s1, n_full=n_full, cat=cat, h1_cols=h1.shape[1], hw=(h, w), batch=batch)
        self.tape = tape
        return rows

    def backward(self, d_rows):
        """Back-propagate d(loss)/d(head rows); fills the flat gradient buffer."""
        cfg, S, tp = self.cfg, self.sparse, self.tape
        s1, n_full = tp["s1"], tp["n_full"]
        sc = cfg.shared_conv_channel
        self.store.absmax.zero_()                               # the layers' max |dz| words (raised by their BatchNorm backward)
        d_h1 = torch.empty((n_full, tp["h1_cols"]), dtype=torch.float32, device=self.device)
        for c, in_col, out_col in self.head2:
            c.backward(d_rows[:, out_col:out_col + c.c_out], s1, n_full, dx_out=d_h1[:, in_col:in_col + sc])
        d_s, _ = self.head1.backward(d_h1, s1, n_full)
        d_cat, _ = self.shared.backward(d_s, s1, n_full)
        carry = None                                            # gradient into the previous level's output
        for lvl in reversed(range(len(self.bev_levels))):
            convs, de, u, c_up = self.bev_levels[lvl]
            adj0, nbr_same, n_in_lvl, n_lvl, col, _ = tp["levels"][lvl]
            d, _ = de.backward(d_cat[:, col:col + c_up], None, n_lvl, add=carry)
            for cv in reversed(convs[1:]):
                d, _ = cv.backward(d, nbr_same, n_lvl)
            carry, _ = convs[0].backward(d, adj0, n_in_lvl)
        self._start_dense_bucket()                              # the BEV + head gradients go on the wire under the sparse half's backward
        batch, h, w, dd, C = tp["dense_shape"]
        dx = carry.view(batch * h * w * dd, C).index_select(0, tp["dense_rows"])      # HeightCompression backward
        for stage, nbr, nbr_dn_t, n_in in reversed(tp["stages"]):
            if stage == "conv_out":
                dx, _ = S["conv_out"].backward(dx, nbr_dn_t, n_in)
            else:
                dx = self._blocks_bwd(S[stage], dx, nbr)
                dx, _ = S[stage + ".down"].backward(dx, nbr_dn_t, n_in)
        dx = self._blocks_bwd(S["conv1"], dx, tp["nbr0"])
        S["conv_input"].backward(dx, None, 0, need_dx=False)
        self.store.join_side()                  # every weight gradient is in the flat buffer from here on
        self.tape = None

    def _start_dense_bucket(self):
        """N > 1: start the all-reduce of the dense half's gradient bucket. Its gradients come from both streams (BatchNorm / bias
        terms on the main one, weight gradients on the side one), so the collective is launched from the side stream after that
        stream has been made to wait for the main stream's work so far: the main stream -- the critical input-gradient chain -- never
        waits."""
        self._reduce = None
        if not (self.bucketed_reduce and (self.pg is not None or self.world > 1)):
            return
        red = dist_utils.BucketedReduce(self.store.grad, self.world, self.pg)
        if red.active:
            side = self.store.side
            if side is not None:
                ev = self.store.mark()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    red.start(self.dense_offset, self.store.grad.numel())
            else:
                red.start(self.dense_offset, self.store.grad.numel())
        self._reduce = red

    def _targets(self, gt_boxes, hw):
        cfg = self.cfg
        fn = train_ops.center_targets if (self.fused_targets and gt_boxes.is_cuda) else center_loss.assign_targets
        return fn(gt_boxes, hw, cfg.point_cloud_range, cfg.voxel_size, cfg.num_class, cfg.feature_map_stride, num_max_objs=self.num_max_objs)

    def targets_early(self, gt_boxes, after=None):
        """CenterHead.assign_targets at the START of the step, on the (then idle) weight-gradient stream: the targets depend on the
        ground-truth boxes only, their ~50 small torch launches and two host read-backs would otherwise sit between the forward and
        the backward pass on the main stream -- the read-backs there make the host wait for the whole forward, and the backward's
        launches are then issued in real time instead of a step ahead. Returns what `loss(..., targets=)` takes (None without a
        side stream). `after`: an event of the main stream the targets are ordered after (default: everything queued on it so far)."""
        side = self.store.side
        if side is None or not gt_boxes.is_cuda:
            return None
        shape = self.cfg.sparse_shape
        for stage in ["conv2", "conv3", "conv4", "conv_out"]:
            k, s, pd = _DOWN[stage]
            shape = ops.conv_out_shape(shape, k, s, pd)
        hw = (shape[1], shape[2])
        main = torch.cuda.current_stream(self.device)
        gt_boxes.record_stream(side)
        with torch.cuda.stream(side):
            if after is not None:                        # (gt_boxes may have been produced on the main stream)
                side.wait_event(after)
            else:
                side.wait_stream(main)
            tg = self._targets(gt_boxes, hw)
            ev = side.record_event()
        for t in tg:
            t.record_stream(main)
        return (tg, ev, hw)

    def loss(self, rows, gt_boxes, targets=None):
        """CenterHead.assign_targets + get_loss (center_head.py:159-250). Returns (loss, d_rows, parts). `targets`: what
        targets_early(gt_boxes) returned for the same boxes."""
        cfg = self.cfg
        h, w = self.tape["hw"]
        batch = self.tape["batch"]
        if targets is not None and targets[2] == (h, w):
            (heat, tgt, inds, masks), ev, _ = targets
            torch.cuda.current_stream(self.device).wait_event(ev)
        else:
            heat, tgt, inds, masks = self._targets(gt_boxes, (h, w))
        if self.fused_loss:                     # one HIP launch group: loss parts + d(loss)/d(rows), no autograd graph
            losses, d_rows = train_ops.center_loss(rows, batch, h * w, cfg.num_class, self.head_slices["hm"][0], heat, tgt, inds,
                                                   masks, self.code_weights)
            return losses[0], d_rows, {"hm_loss": losses[1], "loc_loss": losses[2]}
        leaf = rows.detach().requires_grad_(True)           # torch restatement (tests compare the two)
        loss, parts = center_loss.center_head_loss(leaf, batch, h, w, heat, tgt, inds, masks, cfg.num_class,
                                                   hm_col=self.head_slices["hm"][0], code_weights=self.code_weights)
        loss.backward()
        return loss.detach(), leaf.grad, parts

    def forward_backward(self, points_list, gt_boxes):
        # Target assignment is three launches of cpd_center_targets inside loss() (`fused_targets`, round 5). The torch restatement
        # (fused_targets = False: ~50 small torch launches, two host read-backs) runs on the side stream, ordered after the START of the step
        # only, and is issued once the forward pass is queued: its read-backs then wait for the side stream alone while the main
        # stream works through the forward, and the host -- a step's worth of launches ahead by then -- queues loss and backward without
        # ever waiting for the forward to finish. (Issued first, the host is away from the main stream for ~1 ms at the start of every
        # step: measured, no gain.)
        ev0 = self.store.mark()
        rows = self.forward(points_list)
        tg = self.targets_early(gt_boxes, after=ev0) if (self.early_targets and not self.fused_targets) else None
        loss, d_rows, parts = self.loss(rows, gt_boxes, targets=tg)
        self.backward(d_rows)
        return loss, parts

    def optimizer_step(self):
        st = self.store
        ev = getattr(self, "allreduce_events", None)         # bench.py --mode train: HIP events around the collective
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._reduce is not None:                                          # the dense bucket is on the wire since the middle of the backward pass
            scale = self._reduce.finish()                                    # (sparse bucket now; then wait for both)
            self._reduce = None
        else:
            scale = dist_utils.reduce_gradients(st.grad, self.world, self.pg)   # one collective (no-op without a process group)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        clip = None
        if self.grad_clip:                                           # clip_grad_norm_, train_utils.py:43 -- factor stays on the device
            norm = st.grad.norm() * scale
            clip = torch.clamp(self.grad_clip / (norm + 1e-6), max=1.0).reshape(1).float()
        lr, b1 = self.lr, self.betas[0]
        if self.total_steps:
            lr, b1 = one_cycle(min(self.steps_done, self.total_steps - 1), self.total_steps, self.lr)
        self.steps_done += 1
        train_ops.adam_step(st.flat, st.grad, st.m, st.v, lr, b1, self.betas[1], 1e-8, self.weight_decay,
                            self.steps_done, grad_scale=scale, grad_scale_dev=clip)
        # the packed weight images are rebuilt on the side stream: the next step's voxelizer / index chain / target assignment do not
        # read them, its first convolution waits for the event (forward())
        side = st.side
        if side is not None and self.async_repack:
            ev = st.mark()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                self._repack_all()
                self._repack_ev = side.record_event()
        else:
            self._repack_all()

    def _repack_all(self):
        if self._pack is None:
            for c in self.layers:
                c.repack()
            return
        self._pack.run(self._pack_images)
        for c in self._repack_single:
            c.repack()

    def step(self, points_list, gt_boxes):
        loss, parts = self.forward_backward(points_list, gt_boxes)
        self.optimizer_step()
        return loss, parts

    # ------------------------------------------------------------------ export
    def state_dict(self):
        """Current parameters and BatchNorm statistics under the reference's names and layouts
        (loadable by CenterPointEngine / the reference's Detector3DTemplate)."""
        return self._export(self.store.p, True)

    def grad_dict(self):
        """The last backward's gradients, keyed and laid out like state_dict()'s parameters."""
        return self._export(self.store.g, False)

    def _export(self, V, with_stats):
        cfg = self.cfg
        sd = {}

        def bn_out(c, name, sl=slice(None)):
            sd[name + ".weight"] = V(c.gn)[sl].clone()
            sd[name + ".bias"] = V(c.be)[sl].clone()
            if with_stats:
                sd[name + ".running_mean"] = c.running_mean[sl].clone()
                sd[name + ".running_var"] = c.running_var[sl].clone()
                sd[name + ".num_batches_tracked"] = torch.tensor(self.steps_done, dtype=torch.long)

        def sparse_out(c, name, k):
            w = V(c.wn)                                      # [kv, ci, co]
            sd[name + ".weight"] = w.permute(2, 0, 1).reshape(c.c_out, k[0], k[1], k[2], c.c_in).clone()
            if c.bn_:
                sd[name + ".bias"] = V(c.bn_).clone()

        def conv2d_out(c, name, k=3):
            w = V(c.wn)                                      # [k*k, ci, co]
            sd[name + ".weight"] = w.view(k, k, c.c_in, c.c_out).permute(3, 2, 0, 1).clone()
            if c.bn_:
                sd[name + ".bias"] = V(c.bn_).clone()

        p = "backbone_3d."
        S = self.sparse
        sparse_out(S["conv_input"], p + "conv_input.0", [3, 3, 3]); bn_out(S["conv_input"], p + "conv_input.1")

        def block_out(blk, name):
            for c, cn, bn in ((blk[0], ".conv1", ".bn1"), (blk[1], ".conv2", ".bn2")):
                sparse_out(c, name + cn, [3, 3, 3]); bn_out(c, name + bn)

        block_out(S["conv1"][0], p + "conv1.0"); block_out(S["conv1"][1], p + "conv1.1")
        for stage in ["conv2", "conv3", "conv4"]:
            sparse_out(S[stage + ".down"], p + stage + ".0.0", _DOWN[stage][0]); bn_out(S[stage + ".down"], p + stage + ".0.1")
            block_out(S[stage][0], p + stage + ".1"); block_out(S[stage][1], p + stage + ".2")
        sparse_out(S["conv_out"], p + "conv_out.0", _DOWN["conv_out"][0]); bn_out(S["conv_out"], p + "conv_out.1")

        p = "backbone_2d."
        depth, C = self.depth, cfg.out_features
        for lvl, (convs, de, u, c_up) in enumerate(self.bev_levels):
            names = ["blocks.%d.1" % lvl] + ["blocks.%d.%d" % (lvl, 4 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            bns = ["blocks.%d.2" % lvl] + ["blocks.%d.%d" % (lvl, 5 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            for i, (c, cn, bnn) in enumerate(zip(convs, names, bns)):
                conv2d_out(c, p + cn)
                if lvl == 0 and i == 0:
                    w = sd[p + cn + ".weight"]
                    sd[p + cn + ".weight"] = w.reshape(w.shape[0], depth, C, 3, 3).permute(0, 2, 1, 3, 4).reshape(
                        w.shape[0], depth * C, 3, 3).clone()
                bn_out(c, p + bnn)
            w = V(de.wn)                                     # [1, ci, u*u*co]
            sd[p + "deblocks.%d.0.weight" % lvl] = w.view(de.c_in, u, u, c_up).permute(0, 3, 1, 2).clone()
            bn_out(de, p + "deblocks.%d.1" % lvl)
        p = "dense_head."
        conv2d_out(self.shared, p + "shared_conv.0"); bn_out(self.shared, p + "shared_conv.1")
        sc = cfg.shared_conv_channel
        w1 = V(self.head1.wn)
        for hi, n in enumerate(cfg.head_names()):
            q = p + "heads_list.0.%s." % n
            sl = slice(hi * sc, (hi + 1) * sc)
            sd[q + "0.0.weight"] = w1[:, :, sl].reshape(3, 3, sc, sc).permute(3, 2, 0, 1).clone()
            sd[q + "0.0.bias"] = V(self.head1.bn_)[sl].clone()
            bn_out(self.head1, q + "0.1", sl)
            conv2d_out(self.head2[hi][0], q + "1")
        return sd
