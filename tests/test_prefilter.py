"""Dataloader pre-filter: oracle vs the reference's compiled points_in_boxes_cpu and numpy
semantics on CPU; HIP kernels vs the oracle on the GPU."""
import numpy as np
import pytest


def _boxes_pts(rng, nb, npt, span=20.0):
    boxes = np.concatenate([rng.uniform(-span, span, (nb, 2)), rng.uniform(-1, 1, (nb, 1)), rng.uniform(1.0, 6.0, (nb, 3)),
                            rng.uniform(-np.pi, np.pi, (nb, 1))], 1).astype(np.float32)
    pts = np.concatenate([rng.uniform(-span, span, (npt, 2)), rng.uniform(-3, 3, (npt, 1))], 1).astype(np.float32)
    # half of the points are drawn inside boxes so that hits are common
    k = rng.integers(0, nb, npt // 2)
    local = rng.uniform(-0.6, 0.6, (npt // 2, 3)) * boxes[k, 3:6]
    c, s = np.cos(boxes[k, 6]), np.sin(boxes[k, 6])
    pts[:npt // 2, 0] = boxes[k, 0] + local[:, 0] * c - local[:, 1] * s
    pts[:npt // 2, 1] = boxes[k, 1] + local[:, 0] * s + local[:, 1] * c
    pts[:npt // 2, 2] = boxes[k, 2] + local[:, 2]
    return boxes, pts.astype(np.float32)


def _not_borderline(boxes, pts, margin, tol=1e-4):
    """points whose local coordinates are not within tol of any box face (float64 geometry)."""
    ok = np.ones(pts.shape[0], bool)
    for b in boxes.astype(np.float64):
        sx, sy = pts[:, 0] - b[0], pts[:, 1] - b[1]
        c, s = np.cos(-b[6]), np.sin(-b[6])
        lx, ly = sx * c - sy * s, sx * s + sy * c
        near = (np.abs(np.abs(lx) - (b[3] / 2 + margin)) < tol) | (np.abs(np.abs(ly) - (b[4] / 2 + margin)) < tol) | \
               (np.abs(np.abs(pts[:, 2] - b[2]) - b[5] / 2) < tol)
        ok &= ~near
    return ok


def test_oracle_points_in_boxes_matches_compiled_reference(oracle):
    from oracle.binding import load_reference_points_in_boxes
    ref = load_reference_points_in_boxes()
    if ref is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref; build container only)")
    rng = np.random.default_rng(3)
    boxes, pts = _boxes_pts(rng, 40, 4000)
    mask = ref(boxes, pts)                                     # (N, M) 0/1, MARGIN 1e-2 (roiaware_pool3d.cpp:131)
    first = np.where(mask.any(0), mask.argmax(0), -1).astype(np.int32)
    got = oracle.points_in_boxes(boxes[None], pts[None], margin=1e-2)[0]
    ok = _not_borderline(boxes, pts, 1e-2)
    np.testing.assert_array_equal(got[ok], first[ok])
    assert ok.mean() > 0.95 and (first >= 0).mean() > 0.3


def test_oracle_mask_points_by_range_is_boolean_indexing(oracle):
    rng = np.random.default_rng(4)
    pts = rng.uniform(-90, 90, (5000, 5)).astype(np.float32)
    rg = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    m = (pts[:, 0] >= rg[0]) & (pts[:, 0] <= rg[3]) & (pts[:, 1] >= rg[1]) & (pts[:, 1] <= rg[4])
    np.testing.assert_array_equal(oracle.mask_points_by_range(pts, rg), pts[m])


@pytest.mark.gpu
def test_hip_prefilter_matches_oracle(oracle, hip):
    import torch
    from cpd_amd import prefilter
    rng = np.random.default_rng(5)
    pts = rng.uniform(-90, 90, (200001, 5)).astype(np.float32)
    rg = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    got = prefilter.mask_points_by_range(torch.from_numpy(pts).cuda(), rg)
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.mask_points_by_range(pts, rg))
    perm = torch.from_numpy(rng.permutation(pts.shape[0]))
    np.testing.assert_array_equal(prefilter.shuffle_points(torch.from_numpy(pts).cuda(), perm).cpu().numpy(), pts[perm.numpy()])
    B = 3
    bp = [_boxes_pts(rng, 700, 30011) for _ in range(B)]       # > 512 boxes: two LDS chunks
    boxes = np.stack([b for b, _ in bp]); p3 = np.stack([p for _, p in bp])
    want = oracle.points_in_boxes(boxes, p3, margin=1e-5)
    got = prefilter.points_in_boxes_gpu(torch.from_numpy(p3).cuda(), torch.from_numpy(boxes).cuda()).cpu().numpy()
    for b in range(B):
        ok = _not_borderline(boxes[b], p3[b], 1e-5)
        np.testing.assert_array_equal(got[b][ok], want[b][ok])
        assert ok.mean() > 0.9
    assert (want >= 0).mean() > 0.3


@pytest.mark.gpu
def test_merge_sweeps_matches_the_reference_arithmetic(golden, hip):
    """cpd_merge_sweeps against get_frame / points_rigid_transform (waymo_unsupervised_dataset.py:192-202, 333-360) RUN FROM THE
    REFERENCE FILE (tests/golden/merge_sweeps.npz, make_golden.py::merge_sweeps_fixture): float32 coordinates in a float32 [N, 4]
    matrix, np.mat products with the float64 pose (sweep -> world, then inverse of the current pose), each result cast to
    float32; intensity and the last column zeroed; sweeps concatenated oldest first. Coordinates agree to float32 rounding (the
    4-term float64 dot products may be summed in another order by BLAS), everything else exactly."""
    import torch
    from cpd_amd import prefilter
    g = golden("merge_sweeps")
    poses = [g["poses"][i] for i in range(4)]
    sweeps = [g["sweep%d" % i] for i in range(4)]
    want = g["merged"]
    got = prefilter.merge_sweeps([torch.from_numpy(s).cuda() for s in sweeps], poses, poses[-1]).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, 3:], want[:, 3:])
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=0, atol=2e-5)     # world coordinates ~1e3: float32 ulp 1.2e-4 before the second product
    assert np.mean(got[:, :3] == want[:, :3]) > 0.99
    # the current sweep maps onto itself up to the two roundings
    np.testing.assert_allclose(got[-4097:, :3], sweeps[-1][:, :3], atol=2e-4)
    # what get_frame hands on after prepare_data: columns 3.. zeroed (l.491)
    np.testing.assert_array_equal(g["final_points"][:, 3:], 0)
    np.testing.assert_array_equal(g["final_points"][:, :3], want[:, :3])


PROTO_NAMES = ["Vehicle", "Pedestrian", "Cyclist", "Dis_Small", "Sign"]


def _proto_set(g, to_torch=None):
    ps = {}
    for name in PROTO_NAMES[:3]:
        ps[name] = {}
        for pid in range(3):
            pts = g["set_%s_%d_points" % (name, pid)]
            ps[name][pid] = {"points": to_torch(pts) if to_torch else pts, "box": g["set_%s_%d_box" % (name, pid)]}
    return ps


@pytest.mark.gpu
def test_sample_prototype_matches_reference_method(golden, hip):
    """proto_crop.npz = the reference's sample_prototype_cpu itself (its source run from the reference file on the reference's
    compiled points_in_boxes_cpu). Retained clouds: the same rows in the same order, bit for bit; placed prototypes: the
    reference's float64 result rounded to fp32, <= 1 ulp-level (1e-5 m)."""
    import torch
    from cpd_amd import prefilter
    g = golden("proto_crop")
    thr_max = dict(zip(PROTO_NAMES[:3], g["thr_max"].tolist()))
    thr_min = dict(zip(PROTO_NAMES[:3], g["thr_min"].tolist()))
    ps = _proto_set(g, lambda a: torch.from_numpy(a).cuda())
    for scene in range(2):
        pre = "s%d_" % scene
        pts = torch.from_numpy(g[pre + "points"]).cuda()
        names = [PROTO_NAMES[i] for i in g[pre + "names"]]
        good, proto, nb, nc, nsc, nid = prefilter.sample_prototype(pts, g[pre + "boxes"], names, g[pre + "score"], g[pre + "proto_id"], ps,
                                                                    thr_max, thr_min, coin=int(g[pre + "coin"]), permutation=g[pre + "perm"])
        assert int(g[pre + "coin"]) == scene
        np.testing.assert_array_equal(good.cpu().numpy(), g[pre + "good"])
        want = g[pre + "proto"]
        got = proto.cpu().numpy()
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
        tail = got[-1000:]
        np.testing.assert_array_equal(tail, want[-1000:])                      # the no-object rows at the end are copies: exact
        np.testing.assert_array_equal(nb, g[pre + "new_boxes"])
        np.testing.assert_array_equal([PROTO_NAMES.index(n) for n in nc], g[pre + "new_names"])
        np.testing.assert_allclose(nsc, g[pre + "new_score"], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(nid, g[pre + "new_id"])


@pytest.mark.gpu
def test_points_in_boxes_cpu_and_crop_match_compiled_reference(golden, hip):
    """roiaware_pool3d.cpp's points_in_boxes_cpu, compiled from the reference in the build container and recorded in
    tests/golden/points_in_boxes.npz (the binary itself does not travel), against the device mask, incl. points on the MARGIN
    shell and the z faces; crop_boxes against masks built from that matrix."""
    import torch
    from cpd_amd import prefilter
    g = golden("points_in_boxes")
    boxes, pts, want, discard = g["boxes"], g["points"], g["mask"], g["discard"]
    k, n = boxes.shape[0], pts.shape[0]
    got = prefilter.points_in_boxes_cpu(torch.from_numpy(pts).cuda(), torch.from_numpy(boxes).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    full = np.concatenate([pts, g["extra"]], 1)
    a, b_ = prefilter.crop_boxes(torch.from_numpy(full).cuda(), torch.from_numpy(boxes).cuda(), discard)
    np.testing.assert_array_equal(a.cpu().numpy(), full[want.sum(0) == 0])
    np.testing.assert_array_equal(b_.cpu().numpy(), full[want[discard].sum(0) == 0])
    e0, e1 = prefilter.crop_boxes(torch.from_numpy(full).cuda(), torch.zeros((0, 7)).cuda(), np.zeros(0, bool))   # no boxes: everything stays
    assert e0.shape[0] == n and e1.shape[0] == n


def test_oracle_points_in_boxes_mask_matches_reference_fixture(oracle, golden):
    """the oracle's restatement against the recorded output of the compiled reference (runs anywhere; the live comparison below
    needs oracle/_ref, i.e. the build container)"""
    g = golden("points_in_boxes")
    np.testing.assert_array_equal(oracle.points_in_boxes_mask(g["boxes"], g["points"]), g["mask"])


def test_oracle_points_in_boxes_mask_matches_compiled_reference(oracle):
    from oracle.binding import load_reference_points_in_boxes
    ref = load_reference_points_in_boxes()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    k, n = 25, 6000
    boxes = np.concatenate([rng.uniform(-20, 20, (k, 2)), rng.uniform(-1, 1, (k, 1)), rng.uniform(0.5, 6, (k, 3)), rng.uniform(-3.2, 3.2, (k, 1))], 1).astype(np.float32)
    pts = np.concatenate([rng.uniform(-24, 24, (n, 2)), rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    for i in range(k):
        b = boxes[i]
        c, s = np.cos(b[6]), np.sin(b[6])
        for j, (fx, fy, fz) in enumerate([(0.5, 0, 0), (0.5 + 0.01 / b[3], 0, 0), (0, 0.5 + 0.0099 / b[4], 0), (0, 0, 0.5), (0, 0, 0.50001)]):
            lx, ly, lz = fx * b[3], fy * b[4], fz * b[5]
            pts[i * 5 + j] = [lx * c - ly * s + b[0], lx * s + ly * c + b[1], lz + b[2]]
    np.testing.assert_array_equal(oracle.points_in_boxes_mask(boxes, pts), ref(boxes, pts))
