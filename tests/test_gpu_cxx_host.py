"""The C-ABI from a host with no Python and no torch (examples/cxx_host.cpp, plain hipMalloc'ed pointers):
same voxel count and checksums as the Python host path on the same inputs."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_host_matches_python_host(hip, tmp_path):
    from cpd_amd import ops
    from cpd_amd.synthetic import WAYMO, waymo_cloud
    exe = os.path.join(REPO, "examples", "cxx_host")
    if not os.path.exists(exe):
        pytest.skip("examples/cxx_host not built (__graft_entry__.build())")
    pts = waymo_cloud(3, n_points=50000)
    rng = np.random.default_rng(0)
    w = (rng.normal(size=(27, 5, 16)) * 0.2).astype(np.float32)
    pts.tofile(tmp_path / "points.f32")
    w.tofile(tmp_path / "weights.f32")
    out = subprocess.run([exe, str(tmp_path / "points.f32"), str(tmp_path / "weights.f32")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m, s_mean, s_out, graph, us_graph, us_direct = out.stdout.split()
    # the chain between the read-backs (index build -> rulebook -> conv), captured into a hipGraph and replayed: same bits
    assert graph == "graph_bitwise_equal" and float(us_graph) > 0 and float(us_direct) > 0
    print("cxx host: %s us per graph replay, %s us per direct chain" % (us_graph, us_direct))
    vox = ops.Voxelizer(WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 5, 1000000)
    _, coords, _, mean, n = vox(torch.from_numpy(pts).cuda(), batch_idx=0, coord_cols=4, want_voxels=False, want_mean=True, sync=True)
    assert int(m) == n
    g = ops.voxel_grid_size(WAYMO["voxel_size"], WAYMO["point_cloud_range"])
    index = ops.SiteIndex.build(coords, 1, [g[0] + 1, g[1], g[2]])
    nbr = ops.rulebook_subm(coords, index)
    y = ops.gather_conv(mean, 5, ops.pack_weight(torch.from_numpy(w).cuda()), nbr, 27, n, 16, relu=True)
    assert abs(float(s_mean) - mean.double().sum().item()) <= 1e-6 * abs(mean.double().sum().item()) + 1e-3
    assert abs(float(s_out) - y.double().sum().item()) <= 1e-6 * abs(y.double().sum().item()) + 1e-3
