"""SubMConv3d / SparseConv3d on cpd_gather_conv, differentiable (cpd_amd/autograd_ops.py: the input gradient is the same
kernel on the adjoint weights / rulebook, the weight gradient cpd_conv_wgrad), so the reference's `loss.backward()` loop
(tools/train_utils/train_utils.py:41) trains through these modules. Weight layout is spconv-2.x (Cout, kD, kH, kW, Cin), so
reference checkpoints load (detector3d_template.py:388-419)."""
import math
import os

import torch
import torch.nn as nn

from ... import autograd_ops, ops, train_ops
from .core import SparseConvTensor
from .modules import SparseModule


def _triple(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]


# Arithmetic of the layers with c_in % 32 == 0 when a module is built without an explicit `conv_math=`: "f32" (fp32 MFMA, the
# reference's arithmetic bit for bit as an fma chain), "f16x2" or "bf16x3" (fp32-level results on the 16-bit matrix pipe,
# include/cpd_hip.h). Set by `cpd_amd.spconv.install(conv_math=...)`, `set_default_conv_math()` or CPD_CONV_MATH in the
# environment; a module reads it at CALL time unless it was given its own.
_DEFAULT_CONV_MATH = [os.environ.get("CPD_CONV_MATH", "f32")]


def set_default_conv_math(math):
    assert math in ops.CONV_MATH, math
    _DEFAULT_CONV_MATH[0] = math


def default_conv_math():
    return _DEFAULT_CONV_MATH[0]


# Row order of the levels a strided 3 x 3 x 3 SparseConv3d produces: "canonical" = ascending (b, z, y, x) (the default: what a caller
# who reads `indices` sees is the documented order), "taps" = every chunk of 4096 canonical rows sorted by sub-manifold neighbour
# pattern (ops.order_rows_by_taps; the engine's order, DESIGN 4.1 "row order of the strided levels"): the SubM layers that follow skip
# (16-row group, tap) pairs nearly exactly. Any order is a valid sparse tensor -- spconv's own is a hash order --; features, indices,
# rulebooks, the level's site index (and through it HeightCompression, the RoI pooling's queries and the transposed rulebooks of the
# backward pass) all follow. `cpd_amd.spconv.install(row_order="taps")`, set_default_row_order() or CPD_ROW_ORDER.
_DEFAULT_ROW_ORDER = [os.environ.get("CPD_ROW_ORDER", "canonical")]
ROW_ORDER_MIN_ROWS = 65536            # below this a level does not fill the chip either way (ModelConfig.row_order_min_rows)
ROW_ORDER_CHUNK = 4096


def set_default_row_order(order):
    assert order in ("canonical", "taps"), order
    _DEFAULT_ROW_ORDER[0] = order


def default_row_order():
    return _DEFAULT_ROW_ORDER[0]


# The engine's fast forms on the fused eval path of the MODULES (round 6), f16x2 only, opt-in through
# `cpd_amd.spconv.install(fast_eval=True)` / set_fast_eval() / CPD_FAST_EVAL=1:
#   * fp16-pair rows between the fused sparse layers (the producing epilogue writes x = h + l once; up to 27 gathers of the row take the
#     bits as MFMA fragments) -- a SparseConvTensor then carries ops.PairRows and decodes `.features` only for a caller who reads it;
#   * the OPTIMISTIC range guard: layers run their unscaled kernels and only RECORD max |out| into a pool of blocks; the detector
#     (models.CenterPoint.forward) reads the pool's maximum with its results and runs the step again on the guarded kernels (fp32 rows,
#     today's path) if an activation reached 2^15 -- exactly the engine's scheme (engine._range_reset).
# Outside a `range_pass` (a module called on its own, a custom detector) every layer stays guarded, as before.
_FAST_EVAL = [os.environ.get("CPD_FAST_EVAL", "0") not in ("0", "", "false")]
import threading as _threading


class _RangeState(_threading.local):          # per thread: two batches in flight on two HIP streams (two worker threads) each own a pass
    pool, next, active = None, 0, False


_RANGE = _RangeState()


def set_fast_eval(on):
    _FAST_EVAL[0] = bool(on)


def fast_eval():
    return _FAST_EVAL[0]


class range_pass:
    """with range_pass(device, n_blocks) as rp: ... the fused layers inside run unguarded and record; rp.exceeded() -> device int32 [1]
    (1: some recorded activation >= 2^15, NaN or inf: run the step again outside the pass)."""

    def __init__(self, device, n_blocks=96):
        self.device, self.n = device, int(n_blocks)

    def __enter__(self):
        pool = _RANGE.pool
        if pool is None or pool.device != torch.device(self.device) or pool.shape[0] < self.n:
            pool = _RANGE.pool = ops.absmax_blocks(self.n, self.device)
        else:
            pool.zero_()
        _RANGE.next, _RANGE.active = 0, True
        return self

    def __exit__(self, *exc):
        _RANGE.active = False

    @staticmethod
    def exceeded():
        return (_RANGE.pool.max() >= 0x47000000).to(torch.int32).view(1)       # bits of 32768.0f; NaN / inf bits are larger


def optimistic():
    """inside a range_pass: the fused f16x2 layers record instead of guarding"""
    return _RANGE.active


def record_block():
    """the next block of the active pass's pool (a layer's `out_absmax`)"""
    if _RANGE.next >= _RANGE.pool.shape[0]:
        raise RuntimeError("range_pass: more fused conv launches than the %d absmax blocks of the pass" % _RANGE.pool.shape[0])
    b = _RANGE.pool[_RANGE.next]
    _RANGE.next += 1
    return b


def fold_batchnorm(bn, conv_bias=None):
    """Eval-mode BatchNorm (+ the preceding conv's bias) as the conv epilogue's per-channel (scale, shift), cached on the
    BatchNorm module until one of its tensors (or the bias) changes."""
    tensors = [bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([conv_bias] if conv_bias is not None else [])
    key = tuple((t._version, t.data_ptr()) for t in tensors if t is not None)
    cached = getattr(bn, "_cpd_fold", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    with torch.no_grad():
        var = bn.running_var.double()
        w = bn.weight.double() if bn.weight is not None else torch.ones_like(var)
        b = bn.bias.double() if bn.bias is not None else torch.zeros_like(var)
        scale = w / torch.sqrt(var + bn.eps)
        shift = b - bn.running_mean.double() * scale
        if conv_bias is not None:
            shift = shift + conv_bias.double() * scale
        scale, shift = scale.float().contiguous(), shift.float().contiguous()
    bn._cpd_fold = (key, scale, shift)
    return scale, shift


def fusable_eval(*mods):
    """The fused eval path applies when nothing is being trained: no autograd graph wanted and every module OF THE WHOLE TREE under
    each of `mods` in eval mode, every BatchNorm among them with running statistics to fold. (A container in eval() whose BatchNorm
    was switched back to .train(), or a BatchNorm built with track_running_stats=False, normalises with batch statistics: the
    fused launch would silently use running ones -- such a tree takes the plain module-by-module path. ADVICE r3.)"""
    if torch.is_grad_enabled():
        return False
    for m in mods:
        for s in m.modules():
            if s.training:
                return False
            if isinstance(s, nn.modules.batchnorm._BatchNorm) and (not s.track_running_stats or s.running_mean is None):
                return False
    return True


class SparseConvolution(SparseModule):
    """[SPCONV] spconv.pytorch.conv.SparseConvolution (isinstance check at spconv_utils.py:49)."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, inverse=False, indice_key=None, **kwargs):
        super().__init__()
        assert ndim == 3 and groups == 1 and _triple(dilation) == [1, 1, 1], "only what CPD uses is implemented"
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None
        self._packed_version = None
        self._conv_math = kwargs.get("conv_math")           # None: the package default at call time (set_default_conv_math)

    @property
    def conv_math(self):
        return self._conv_math if self._conv_math is not None else _DEFAULT_CONV_MATH[0]

    @conv_math.setter
    def conv_math(self, m):
        assert m is None or m in ops.CONV_MATH, m
        self._conv_math = m

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _packed_weight(self):
        key = (self.weight._version, self.weight.data_ptr(), self.weight.device)
        if self._packed is None or self._packed_version != key:
            w = self.weight.detach()
            kio = w.reshape(self.out_channels, -1, self.in_channels).permute(1, 2, 0).contiguous()
            self._packed = ops.pack_weight(kio)
            self._packed_version = key
        return self._packed

    def forward(self, x: SparseConvTensor, scale=None, shift=None, residual=None, relu=False):
        """scale / shift / residual / relu: an eval-mode epilogue fused into the launch (SparseSequential and SparseBasicBlock
        pass the folded BatchNorm1d, the block's identity and the ReLU here when nothing is being trained); without them this
        is the plain convolution (+ bias) of the reference module."""
        assert isinstance(x, SparseConvTensor)
        if self.inverse:
            raise NotImplementedError("SparseInverseConv3d is declared by the reference (spconv_backbone.py:24) "
                                      "but never instantiated by any shipped config")
        fused = scale is not None or shift is not None or residual is not None or relu
        # fp16-pair rows in (round 6): only between fused split-fp16 layers of an optimistic pass (the guarded re-run reads fp32 rows)
        fast = fused and optimistic() and self.conv_math == "f16x2" and not torch.is_grad_enabled()
        in_pairs = fast and x._pairs is not None and (self.in_channels % 32 == 0 or self.in_channels == 16)
        feats = x._pairs.rows if in_pairs else x.features.contiguous().float()
        if not feats.is_cuda:
            raise ops._lib.CpdHipError("cpd_amd.spconv runs on the GPU only (no CPU fallback)")
        kv = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        cached = x.find_indice_pair(self.indice_key)
        if self.subm:
            if cached is None:
                # (a chunk-ordered level -- row order "taps" -- carries its canonical list: the table is then built chunk-wise)
                nbr = ops.rulebook_subm(x.indices.contiguous(), x.site_index(), self.kernel_size, canonical=getattr(x, "_row_canon", None))
                cached = dict(nbr=nbr, out_indices=x.indices, out_shape=x.spatial_shape, out_index=x.site_index(),
                              canon=getattr(x, "_row_canon", None))
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = cached
        else:
            if cached is None:
                src = getattr(x, "_row_canon", None)          # the output set is marked from the canonical list where the level has one
                out_idx, out_index, out_shape = ops.conv_outset((src[0] if src is not None else x.indices).contiguous(), x.batch_size,
                                                                x.spatial_shape, self.kernel_size, self.stride, self.padding)
                canon = None
                if (_DEFAULT_ROW_ORDER[0] == "taps" and list(self.kernel_size) == [3, 3, 3] and list(self.stride) == [2, 2, 2]
                        and out_idx.shape[0] >= ROW_ORDER_MIN_ROWS):
                    out_c = out_idx
                    out_idx, _, old_to_new = ops.order_rows_by_taps(out_c, out_index, chunk_rows=ROW_ORDER_CHUNK)
                    out_index.set_order(old_to_new)
                    canon = (out_c, old_to_new, ROW_ORDER_CHUNK)
                nbr = ops.rulebook_conv(out_idx, x.site_index(), self.kernel_size, self.stride, self.padding, canonical=canon)
                cached = dict(nbr=nbr, out_indices=out_idx, out_shape=out_shape, out_index=out_index, canon=canon)
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = cached
        nbr, out_indices = cached["nbr"], cached["out_indices"]

        def adjoint():          # transposed rulebook of a regular conv (input gradient), built on first use, cached with the pairs
            if "nbr_t" not in cached:
                cached["nbr_t"] = train_ops.rulebook_conv_transpose(x.indices.contiguous(), x.batch_size, x.spatial_shape,
                                                                    self.kernel_size, self.stride, self.padding, cached["out_index"])
            return cached["nbr_t"]

        spec = autograd_ops.ConvSpec(nbr, kv, out_indices.shape[0], dense=False, math=self.conv_math,
                                     mode="same" if self.subm else "strided", adjoint=adjoint, packed=self._packed_weight())
        if fused:
            assert not torch.is_grad_enabled(), "the fused epilogue is an inference path"
            shift = ops.epilogue_shift(scale, shift, self.bias)
            if fast:
                # optimistic pass: unscaled kernels, max |out| recorded; pair rows out where the consumer is another fused sparse layer
                # (`pairs_out`, set by the containers: SparseBasicBlock, the backbone) and the kernels have the form -- the engine's
                # combinations: fp32 5-channel rows -> 16-channel pairs, pairs -> pairs, a pair residual with pair rows out
                out_pairs = bool(getattr(self, "pairs_out", False)) and (self.out_channels % 32 == 0 or self.out_channels == 16) \
                    and (in_pairs or self.in_channels < 16)
                res_pairs = isinstance(residual, ops.PairRows)
                if res_pairs and not out_pairs:
                    residual, res_pairs = residual.float_rows(), False
                out_feats = ops.gather_conv(feats, self.in_channels, spec.packed, nbr, kv, spec.n_out, self.out_channels, scale, shift,
                                            residual.rows if res_pairs else residual, relu, dense=False, math=self.conv_math,
                                            out_absmax=record_block(), in_pairs=in_pairs, out_pairs=out_pairs, res_pairs=res_pairs)
                if out_pairs:
                    out_feats = ops.PairRows(out_feats)
            else:
                if isinstance(residual, ops.PairRows):
                    residual = residual.float_rows()
                # f16x2: the range block travels with the feature tensor (ops.tag_range); a tensor without one -- or modified in place
                # since it was tagged -- is measured first
                guard = self.conv_math == "f16x2"
                rb_in = ops.range_block(x.features, self.in_channels) if guard and self.in_channels % 32 == 0 else None
                rb_out = ops.absmax_blocks(1, feats.device)[0] if guard else None
                out_feats = ops.gather_conv(feats, self.in_channels, spec.packed, nbr, kv, spec.n_out, self.out_channels, scale, shift,
                                            residual, relu, dense=False, math=self.conv_math, in_absmax=rb_in, out_absmax=rb_out)
                ops.tag_range(out_feats, rb_out)
        else:
            w_kio = self.weight.reshape(self.out_channels, kv, self.in_channels).permute(1, 2, 0)
            out_feats = autograd_ops.gather_conv(feats, w_kio, self.bias, spec)
        out = SparseConvTensor(out_feats, out_indices, cached["out_shape"], x.batch_size, x.grid, x.benchmark)
        out.indice_dict = x.indice_dict
        out._site_index = cached["out_index"]
        out._row_canon = cached.get("canon")
        return out


class SubMConv3d(SparseConvolution):
    """spconv.SubMConv3d(in, out, k, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key, **kwargs)


class SparseConv3d(SparseConvolution):
    """spconv.SparseConv3d(in, out, k, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key, **kwargs)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key, **kwargs)
