"""`spconv.pytorch`-shaped module API on the HIP kernels: the names and argument conventions
cpd/models/backbones_3d/spconv_backbone.py, cpd/utils/spconv_utils.py and
cpd/models/backbones_2d/map_to_bev/height_compression.py:136 rely on."""
from . import conv, utils  # noqa: F401
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401
