"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports exactly the symbols
include/cpd_hip.h declares, and its host-only entry points behave (no kernels are launched)."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(REPO, "include", "cpd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cpd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from cpd_amd import _lib
    lib = _lib.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libcpd_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in lib.cpd_version()


def test_missing_library_fails_loudly(monkeypatch):
    from cpd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcpd_hip.so")
    with pytest.raises(_lib.CpdHipError):
        _lib.lib()


def test_host_geometry_helpers(oracle):
    from cpd_amd import ops
    vs, pcr = [0.1, 0.1, 0.15], [-75.2, -75.2, -2, 75.2, 75.2, 4]
    assert ops.voxel_grid_size(vs, pcr) == oracle.grid_size(vs, pcr) == [40, 1504, 1504]
    assert ops.voxel_grid_size([0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]) == [40, 1600, 1408]
    for k, s, p in [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]), ([3, 1, 1], [2, 1, 1], [0, 0, 0])]:
        assert ops.conv_out_shape([41, 1504, 1504], k, s, p) == oracle.conv_out_shape([41, 1504, 1504], k, s, p)
    # level shapes of SURVEY section 8: W config
    sh = [41, 1504, 1504]
    for k, s, p in [([3, 3, 3], [2, 2, 2], [1, 1, 1])] * 2 + [([3, 3, 3], [2, 2, 2], [0, 1, 1]), ([3, 1, 1], [2, 1, 1], [0, 0, 0])]:
        sh = ops.conv_out_shape(sh, k, s, p)
    assert sh == [2, 188, 188]


def test_size_queries_and_error_codes():
    from cpd_amd import _lib
    lib = _lib.lib()
    vs, pcr = _lib.farr([0.1, 0.1, 0.15]), _lib.farr([-75.2, -75.2, -2, 75.2, 75.2, 4])
    assert lib.cpd_voxelize_workspace_bytes(160000, 5, 1000000, vs, pcr) > 11 * 2 ** 20
    assert lib.cpd_index_bytes(1, _lib.iarr([41, 1504, 1504]), 100000) > 16 * 2 ** 20
    assert lib.cpd_packed_weight_floats(27, 5, 16) == 27 * 1 * 4 * 16 * 4
    assert lib.cpd_nms_workspace_bytes(500) >= 500 * 8 * 8
    # bad arguments give error codes, never exit()
    assert lib.cpd_gather_conv(None, 0, 0, 0, None, None, None, 0, 0, 0, None, None, None, 0, 0, None, 0, None, 0, 0, None) == -1
    assert lib.cpd_voxelize(None, -1, 5, vs, pcr, 5, 10, 0, 4, None, None, None, None, None, None, 0, None) == -1
    o = (ctypes.c_int32 * 3)()
    assert lib.cpd_conv_out_shape(_lib.iarr([1, 1, 1]), _lib.iarr([3, 3, 3]), _lib.iarr([2, 2, 2]), _lib.iarr([0, 0, 0]), o) == -1


def test_boxes_iou_bev_cpu_entry_point(golden):
    """The extension's one CPU entry point (iou3d_cpu.cpp:232-252) against the reference golden."""
    import torch
    from cpd_amd import ops
    g = golden("iou_bev")
    got = ops.boxes_iou_bev_cpu(torch.from_numpy(g["a"]), torch.from_numpy(g["b"])).numpy()
    np.testing.assert_allclose(got, g["iou_ab"], atol=1e-6, rtol=0)
