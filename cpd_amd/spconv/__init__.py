"""Drop-in for the subset of `spconv` (spconv-cu111==2.1.22) that CPD imports, backed by the HIP
kernels of this package. `cpd/utils/spconv_utils.py:34-36` does `import spconv.pytorch as spconv`;
`cpd_amd.spconv.install()` registers these modules under the `spconv` / `cumm` names so the
reference files import unchanged."""
import sys
import types

from . import pytorch  # noqa: F401
from . import utils  # noqa: F401
from .pytorch import (SparseConv3d, SparseConvTensor, SparseInverseConv3d, SparseModule,  # noqa: F401
                      SparseSequential, SubMConv3d)

__version__ = "2.1.22+cpd_amd"


def install(conv_math=None, row_order=None, fast_eval=None):
    """conv_math: package default for modules built without their own (`"f32"` | `"f16x2"` | `"bf16x3"`,
    cpd_amd.spconv.pytorch.conv.set_default_conv_math; also CPD_CONV_MATH).
    row_order: `"canonical"` (default) | `"taps"` -- the row order of the levels strided SparseConv3d layers produce
    (cpd_amd.spconv.pytorch.conv.set_default_row_order; also CPD_ROW_ORDER).
    fast_eval: True | False -- the engine's fast forms on the fused eval path of the f16x2 modules: fp16-pair rows between fused sparse
    layers and the optimistic range guard (cpd_amd.spconv.pytorch.conv.set_fast_eval; also CPD_FAST_EVAL). Same detections.
    Make `import spconv`, `import spconv.pytorch`, `from spconv.pytorch.utils import PointToVoxel`,
    `from spconv.utils import Point2VoxelCPU3d` and `import cumm.tensorview as tv` resolve to this
    package (only if the real spconv is absent)."""
    me = sys.modules[__name__]
    if conv_math is not None:
        pytorch.conv.set_default_conv_math(conv_math)
    if row_order is not None:
        pytorch.conv.set_default_row_order(row_order)
    if fast_eval is not None:
        pytorch.conv.set_fast_eval(fast_eval)
    for name, mod in {"spconv": me, "spconv.pytorch": pytorch, "spconv.pytorch.conv": pytorch.conv,
                      "spconv.pytorch.utils": pytorch.utils, "spconv.utils": utils}.items():
        sys.modules.setdefault(name, mod)
    cumm = types.ModuleType("cumm")
    cumm.tensorview = utils.tensorview
    sys.modules.setdefault("cumm", cumm)
    sys.modules.setdefault("cumm.tensorview", utils.tensorview)
    return me
