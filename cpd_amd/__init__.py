"""cpd_amd -- MI355X-native implementation of CPD's detection hot path
(voxelize -> sparse 3D conv backbone -> BEV dense head -> rotated NMS) behind the reference's
operator API. All compute lives in csrc/libcpd_hip.so (hand-written HIP for gfx950, C-ABI in
include/cpd_hip.h); this package is the Python host side that mirrors the reference interfaces.
There is no CPU / eager fallback: ops raise if the HIP library is missing.
"""
from . import _lib  # noqa: F401
from ._lib import CpdHipError, lib  # noqa: F401

__version__ = "0.1.0"
