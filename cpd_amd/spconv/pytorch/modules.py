import torch.nn as nn


class SparseModule(nn.Module):
    """Marker base class ([SPCONV] spconv.SparseModule; spconv_backbone.py:37,100)."""


def is_spconv_module(m):
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    """[SPCONV] SparseSequential (A.6): sparse modules get the tensor, plain nn.Modules
    (BatchNorm1d, ReLU) are applied to `.features` (spconv_backbone.py:29-33,414-455)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        from .conv import SparseConvolution, fold_batchnorm, fusable_eval
        from .core import SparseConvTensor
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            i += 1
            # eval mode, no autograd: conv -> BatchNorm1d [-> ReLU] is ONE launch, the folded statistics and the ReLU in the conv
            # epilogue (post_act_block / conv_input / conv_out of the backbone, spconv_backbone.py:13-35,414-455)
            if isinstance(m, SparseConvolution) and not m.inverse and i < len(mods) and isinstance(mods[i], nn.BatchNorm1d) \
                    and isinstance(x, SparseConvTensor) and fusable_eval(m, mods[i]) and mods[i].track_running_stats:
                bn = mods[i]
                i += 1
                relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
                i += int(relu)
                scale, shift = fold_batchnorm(bn, m.bias)
                x = m(x, scale=scale, shift=shift, relu=relu)
                continue
            if is_spconv_module(m):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x
