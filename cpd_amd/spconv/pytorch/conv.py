"""SubMConv3d / SparseConv3d on cpd_gather_conv, differentiable (cpd_amd/autograd_ops.py: the input gradient is the same
kernel on the adjoint weights / rulebook, the weight gradient cpd_conv_wgrad), so the reference's `loss.backward()` loop
(tools/train_utils/train_utils.py:41) trains through these modules. Weight layout is spconv-2.x (Cout, kD, kH, kW, Cin), so
reference checkpoints load (detector3d_template.py:388-419)."""
import math

import torch
import torch.nn as nn

from ... import autograd_ops, ops, train_ops
from .core import SparseConvTensor
from .modules import SparseModule


def _triple(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]


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
        self.conv_math = kwargs.get("conv_math", "f32")     # "f32" | "bf16x3" | "f16x2" (ops.CONV_MATH), layers with c_in % 32 == 0

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

    def forward(self, x: SparseConvTensor):
        assert isinstance(x, SparseConvTensor)
        if self.inverse:
            raise NotImplementedError("SparseInverseConv3d is declared by the reference (spconv_backbone.py:24) "
                                      "but never instantiated by any shipped config")
        feats = x.features.contiguous().float()
        if not feats.is_cuda:
            raise ops._lib.CpdHipError("cpd_amd.spconv runs on the GPU only (no CPU fallback)")
        kv = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        cached = x.find_indice_pair(self.indice_key)
        if self.subm:
            if cached is None:
                nbr = ops.rulebook_subm(x.indices.contiguous(), x.site_index(), self.kernel_size)
                cached = dict(nbr=nbr, out_indices=x.indices, out_shape=x.spatial_shape, out_index=x.site_index())
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = cached
        else:
            if cached is None:
                out_idx, out_index, out_shape = ops.conv_outset(x.indices.contiguous(), x.batch_size, x.spatial_shape,
                                                                self.kernel_size, self.stride, self.padding)
                nbr = ops.rulebook_conv(out_idx, x.site_index(), self.kernel_size, self.stride, self.padding)
                cached = dict(nbr=nbr, out_indices=out_idx, out_shape=out_shape, out_index=out_index)
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
        w_kio = self.weight.reshape(self.out_channels, kv, self.in_channels).permute(1, 2, 0)
        out_feats = autograd_ops.gather_conv(feats, w_kio, self.bias, spec)
        out = SparseConvTensor(out_feats, out_indices, cached["out_shape"], x.batch_size, x.grid, x.benchmark)
        out.indice_dict = x.indice_dict
        out._site_index = cached["out_index"]
        return out


class SubMConv3d(SparseConvolution):
    """spconv.SubMConv3d(in, out, k, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    """spconv.SparseConv3d(in, out, k, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)
