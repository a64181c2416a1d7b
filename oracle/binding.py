"""ctypes/numpy binding of oracle/libcpd_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcpd_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libiou3d_ref.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)


def build_oracle(force=False):
    """Compile the C restatement with gcc (seconds). Also builds oracle/_ref when the reference
    tree is present (build container only)."""
    src = os.path.join(_HERE, "cpd_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    ref_src = "/root/reference/cpd/ops/iou3d_nms/src/iou3d_cpu.cpp"
    ref_libs = [_REF, os.path.join(_HERE, "_ref", "libroiaware_ref.so")]
    if os.path.exists(ref_src) and not all(os.path.exists(f) for f in ref_libs):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB


def load_reference_iou():
    """Returns f(a[N,7], b[M,7]) -> iou[N,M] backed by the reference's compiled iou3d_cpu.cpp,
    or None when oracle/_ref has not been built."""
    if not os.path.exists(_REF):
        return None
    import torch  # noqa: F401  (libtorch must be loadable)
    lib = ctypes.CDLL(_REF)

    def f(a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        lib.ref_boxes_iou_bev_cpu(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
        return out

    return f


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_i32_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _arr3(v):
    return (ctypes.c_int32 * 3)(*[int(x) for x in v])


def _farr(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


_REF_ROIAWARE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libroiaware_ref.so")


def load_reference_points_in_boxes():
    """Returns f(boxes[N,7], pts[M,3]) -> mask[N,M] i32 backed by the reference's compiled
    roiaware_pool3d.cpp (points_in_boxes_cpu, MARGIN 1e-2), or None when oracle/_ref has not been built."""
    if not os.path.exists(_REF_ROIAWARE):
        return None
    import torch  # noqa: F401  (libtorch / libtorch_python must be loaded first)
    lib = ctypes.CDLL(_REF_ROIAWARE, mode=os.RTLD_LAZY)    # its unused CUDA launchers stay unresolved

    def f(boxes, pts):
        boxes = np.ascontiguousarray(boxes, np.float32)
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.zeros((boxes.shape[0], pts.shape[0]), np.int32)
        lib.ref_points_in_boxes_cpu(_fp(boxes), boxes.shape[0], _fp(pts), pts.shape[0], _ip(out))
        return out

    return f


class Oracle:
    """numpy-level wrappers around the C oracle. One method per C-ABI entry point of
    include/cpd_hip.h, same argument meaning, host arrays."""

    def __init__(self):
        build_oracle()
        self.lib = ctypes.CDLL(_LIB)
        self.lib.cpd_ref_box_overlap.restype = ctypes.c_float
        self.lib.cpd_ref_iou_bev.restype = ctypes.c_float

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("oracle %s failed: %d" % (what, rc))

    # ---- B1 voxelizer ------------------------------------------------------------------------
    def grid_size(self, vsize_xyz, range_xyz):
        g = (ctypes.c_int32 * 3)()
        self.lib.cpd_ref_grid_size(_farr(vsize_xyz), _farr(range_xyz), g)
        return [g[0], g[1], g[2]]

    def voxelize(self, points, vsize_xyz, range_xyz, max_points, max_voxels):
        points = _f32(points)
        n, c = points.shape
        max_voxels = max(1, min(int(max_voxels), n))    # a cloud of n points fills at most n voxels
        voxels = np.empty((max_voxels, max_points, c), np.float32)
        coords = np.empty((max_voxels, 3), np.int32)
        num = np.empty((max_voxels,), np.int32)
        m = ctypes.c_int32(0)
        self._check(self.lib.cpd_ref_voxelize(_fp(points), n, c, _farr(vsize_xyz), _farr(range_xyz),
                                               max_points, max_voxels, _fp(voxels), _ip(coords), _ip(num),
                                               ctypes.byref(m)), "voxelize")
        m = m.value
        return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()

    def mean_vfe(self, voxels, num_points):
        voxels = _f32(voxels)
        m, p, c = voxels.shape
        out = np.empty((m, c), np.float32)
        self._check(self.lib.cpd_ref_mean_vfe(_fp(voxels), _ip(_i32(num_points)), m, p, c, _fp(out)), "mean_vfe")
        return out

    # ---- B2 sparse conv ----------------------------------------------------------------------
    def subm_rulebook(self, indices, batch, shape, ksize):
        indices = _i32(indices)
        n = indices.shape[0]
        kv = int(np.prod(ksize))
        nbr = np.empty((kv, n), np.int32)
        self._check(self.lib.cpd_ref_subm_rulebook(_ip(indices), n, batch, _arr3(shape), _arr3(ksize), _ip(nbr)),
                    "subm_rulebook")
        return nbr

    def conv_out_shape(self, in_shape, ksize, stride, pad):
        o = (ctypes.c_int32 * 3)()
        self._check(self.lib.cpd_ref_conv_out_shape(_arr3(in_shape), _arr3(ksize), _arr3(stride), _arr3(pad), o),
                    "conv_out_shape")
        return [o[0], o[1], o[2]]

    def conv_outset(self, in_indices, batch, in_shape, ksize, stride, pad):
        in_indices = _i32(in_indices)
        n = in_indices.shape[0]
        kv = int(np.prod(ksize))
        cap = max(1, n * kv)
        out = np.empty((cap, 4), np.int32)
        cnt = ctypes.c_int32(0)
        self._check(self.lib.cpd_ref_conv_outset(_ip(in_indices), n, batch, _arr3(in_shape), _arr3(ksize),
                                                  _arr3(stride), _arr3(pad), _ip(out), cap, ctypes.byref(cnt)),
                    "conv_outset")
        return out[:cnt.value].copy()

    def conv_rulebook(self, in_indices, out_indices, batch, in_shape, ksize, stride, pad):
        in_indices = _i32(in_indices)
        out_indices = _i32(out_indices)
        kv = int(np.prod(ksize))
        nbr = np.empty((kv, out_indices.shape[0]), np.int32)
        self._check(self.lib.cpd_ref_conv_rulebook(_ip(in_indices), in_indices.shape[0], _ip(out_indices),
                                                    out_indices.shape[0], batch, _arr3(in_shape), _arr3(ksize),
                                                    _arr3(stride), _arr3(pad), _ip(nbr)), "conv_rulebook")
        return nbr

    def sparse_conv(self, feat_in, weight, bias, nbr):
        """weight: reference layout (Cout, kD, kH, kW, Cin)."""
        feat_in = _f32(feat_in)
        weight = _f32(weight)
        cout, cin = weight.shape[0], weight.shape[-1]
        kv, n_out = nbr.shape
        assert weight.size == cout * kv * cin and feat_in.shape[1] == cin
        out = np.empty((n_out, cout), np.float32)
        b = _f32(bias) if bias is not None else None
        self._check(self.lib.cpd_ref_sparse_conv(_fp(feat_in), cin, _fp(weight), _fp(b) if b is not None else None,
                                                  _ip(_i32(nbr)), kv, n_out, cout, _fp(out)), "sparse_conv")
        return out

    def affine_rows(self, x, scale=None, shift=None, residual=None, relu=False):
        x = _f32(x).copy()
        n, c = x.shape
        s = _f32(scale) if scale is not None else None
        t = _f32(shift) if shift is not None else None
        r = _f32(residual) if residual is not None else None
        self._check(self.lib.cpd_ref_affine_rows(_fp(x), n, c, _fp(s) if s is not None else None,
                                                  _fp(t) if t is not None else None,
                                                  _fp(r) if r is not None else None, int(relu)), "affine_rows")
        return x

    def densify(self, feat, indices, batch, shape):
        feat = _f32(feat)
        n, c = feat.shape
        out = np.empty((batch, c * shape[0], shape[1], shape[2]), np.float32)
        self._check(self.lib.cpd_ref_densify(_fp(feat), _ip(_i32(indices)), n, c, batch, _arr3(shape), _fp(out)),
                    "densify")
        return out

    # ---- RoI-head pooling ----------------------------------------------------------
    def voxel2pinds(self, indices, batch, shape):
        indices = _i32(indices)
        out = np.empty((batch, shape[0], shape[1], shape[2]), np.int32)
        self._check(self.lib.cpd_ref_voxel2pinds(_ip(indices), indices.shape[0], batch, _arr3(shape), _ip(out)), "voxel2pinds")
        return out

    def voxel_query(self, max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
        """Raw kernel semantics: idx [M, nsample] i32 pre-zeroed, idx[:,0] = -1 for empty balls."""
        xyz, new_xyz = _f32(xyz), _f32(new_xyz)
        new_coords, point_indices = _i32(new_coords), _i32(point_indices)
        m = new_coords.shape[0]
        b, z, y, x = point_indices.shape
        idx = np.zeros((m, nsample), np.int32)
        self._check(self.lib.cpd_ref_voxel_query(m, z, y, x, nsample, ctypes.c_float(radius), int(max_range[0]), int(max_range[1]),
                                                 int(max_range[2]), _fp(new_xyz), _fp(xyz), _ip(new_coords), _ip(point_indices),
                                                 _ip(idx)), "voxel_query")
        return idx

    def group_points(self, features, features_batch_cnt, idx, idx_batch_cnt):
        features = _f32(features)
        idx = _i32(idx)
        fbc, ibc = _i32(features_batch_cnt), _i32(idx_batch_cnt)
        m, ns = idx.shape
        c = features.shape[1]
        out = np.empty((m, c, ns), np.float32)
        self._check(self.lib.cpd_ref_group_points(fbc.shape[0], m, c, ns, _fp(features), _ip(fbc), _ip(idx), _ip(ibc), _fp(out)),
                    "group_points")
        return out

    # ---- dataloader pre-filter -------------------------------------------------------
    def mask_points_by_range(self, points, limit_range):
        points = _f32(points)
        out = np.empty_like(points)
        n_out = ctypes.c_int32(0)
        self._check(self.lib.cpd_ref_mask_points_by_range(_fp(points), points.shape[0], points.shape[1], _farr(limit_range),
                                                          _fp(out), ctypes.byref(n_out)), "mask_points_by_range")
        return out[:n_out.value]

    def points_in_boxes(self, boxes, pts, margin=1e-5):
        """boxes [B,N,7], pts [B,M,3] -> [B,M] i32 index of the first containing box or -1."""
        boxes, pts = _f32(boxes), _f32(pts)
        b, n, _ = boxes.shape
        m = pts.shape[1]
        out = np.empty((b, m), np.int32)
        self._check(self.lib.cpd_ref_points_in_boxes(b, n, m, _fp(boxes), _fp(pts), ctypes.c_float(margin), _ip(out)),
                    "points_in_boxes")
        return out

    # ---- anchor head ---------------------------------------------------------------
    def points_in_boxes_mask(self, boxes, pts):
        """roiaware_pool3d.cpp:128-168 (points_in_boxes_cpu): mask [K, N] int32. fp32 local coordinates (cos / sin of -rz),
        comparisons in double against d / 2.0 + (double)(float)1e-2, z against dz / 2.0 without margin."""
        f = np.float32
        b, p = _f32(boxes)[:, :7], _f32(pts)[:, :3]
        out = np.zeros((b.shape[0], p.shape[0]), np.int32)
        for i, q in enumerate(b):
            zin = ~(np.abs((p[:, 2] - q[2]).astype(f)).astype(np.float64) > np.float64(q[5]) / 2.0)
            sx, sy = (p[:, 0] - q[0]).astype(f), (p[:, 1] - q[1]).astype(f)
            ca, sa = f(np.cos(f(-q[6]))), f(np.sin(f(-q[6])))
            lx = ((sx * ca).astype(f) + (sy * f(-sa)).astype(f)).astype(f)
            ly = ((sx * sa).astype(f) + (sy * ca).astype(f)).astype(f)
            mg = np.float64(f(1e-2))
            out[i] = zin & (np.abs(lx).astype(np.float64) < np.float64(q[3]) / 2.0 + mg) & (np.abs(ly).astype(np.float64) < np.float64(q[4]) / 2.0 + mg)
        return out

    def nearest_bev_iou(self, a, b):
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self._check(self.lib.cpd_ref_nearest_bev_iou(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out)), "nearest_bev_iou")
        return out

    def residual_encode(self, boxes, anchors):
        boxes, anchors = _f32(boxes), _f32(anchors)
        out = np.empty((boxes.shape[0], 7), np.float32)
        self._check(self.lib.cpd_ref_residual_encode(_fp(boxes), _fp(anchors), boxes.shape[0], _fp(out)), "residual_encode")
        return out

    def anchor_decode(self, box_preds, anchors, dir_cls=None, dir_offset=0.78539, dir_limit_offset=0.0):
        box_preds, anchors = _f32(box_preds), _f32(anchors)
        b, n, _ = box_preds.shape
        out = np.empty((b, n, 7), np.float32)
        d = _f32(dir_cls) if dir_cls is not None else None
        self._check(self.lib.cpd_ref_anchor_decode(_fp(box_preds), _fp(anchors), _fp(d) if d is not None else None, b, n,
                                                   d.shape[-1] if d is not None else 0, ctypes.c_float(dir_offset),
                                                   ctypes.c_float(dir_limit_offset), _fp(out)), "anchor_decode")
        return out

    def anchor_assign(self, anchors, gt, gt_classes, matched, unmatched, norm_by_num_examples=False):
        anchors, gt = _f32(anchors), _f32(gt).reshape(-1, 7)
        cls = _i32(gt_classes)
        n = anchors.shape[0]
        labels = np.empty((n,), np.int32)
        tgt = np.empty((n, 7), np.float32)
        w = np.empty((n,), np.float32)
        ious = np.empty((n,), np.float32)
        self._check(self.lib.cpd_ref_anchor_assign(_fp(anchors), n, _fp(gt), gt.shape[0], _ip(cls), ctypes.c_float(matched),
                                                   ctypes.c_float(unmatched), int(norm_by_num_examples), _ip(labels), _fp(tgt),
                                                   _fp(w), _fp(ious)), "anchor_assign")
        return labels, tgt, w, ious

    def atss_assign(self, anchors, gt_with_classes, topk, match_height=False):
        """ATSSTargetAssigner.assign_targets_single (atss_target_assigner.py:76-141) on the full anchors x GT matrices, fp32
        operation by operation (numpy; the IoU matrices come from this oracle's C restatement of the reference kernels).
        gt_with_classes [M, 8] with trailing padding already trimmed (l.40-45). Where torch leaves the result unspecified
        the rule is spelled out: top-k ties -> lower anchor index; argmax ties -> first index; duplicate forced anchors ->
        the later GT. Returns labels [N] float32, reg_targets [N, 7], reg_weights [N]."""
        f = np.float32
        a = _f32(anchors)[:, :7]
        gt = _f32(gt_with_classes)
        g, cls = np.ascontiguousarray(gt[:, :7]), gt[:, -1]
        n, m, k = a.shape[0], g.shape[0], int(topk)
        ious = self.boxes_iou3d(a, g) if match_height else self.boxes_iou_bev(a, g)                 # (N, M), l.89-92
        d = a[:, None, :3] - g[None, :, :3]
        dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(f)   # l.94
        idx = np.stack([np.lexsort((np.arange(n), dist[:, j]))[:k] for j in range(m)], 1)            # (K, M), l.95
        cand = ious[idx, np.arange(m)[None, :]]                                                      # l.96
        total, mean, m2 = np.zeros(m, f), np.zeros(m, f), np.zeros(m, f)
        for j in range(k):                                         # sum / k; Welford M2 as torch.std accumulates it
            x = cand[j]
            total = (total + x).astype(f)
            delta = (x - mean).astype(f)
            mean = (mean + delta / f(j + 1)).astype(f)
            m2 = (m2 + delta * (x - mean).astype(f)).astype(f)
        with np.errstate(invalid="ignore", divide="ignore"):
            std = np.sqrt(m2 / f(k - 1)).astype(f)
        thresh = ((total / f(k)).astype(f) + std).astype(f) + f(1e-6)                                # l.97-99
        is_pos = cand >= thresh[None, :]
        ca = a[idx.reshape(-1)]                                                                      # (K*M, 7), l.103
        gg = np.tile(g, (k, 1))
        loc = (ca[:, :3] - gg[:, :3]).astype(f)
        ang = (-gg[:, 6]).astype(f)
        c, s_ = np.cos(ang).astype(f), np.sin(ang).astype(f)
        xr = ((loc[:, 0] * c).astype(f) + (loc[:, 1] * (-s_)).astype(f)).astype(f) + (loc[:, 2] * f(0)).astype(f)
        yr = ((loc[:, 0] * s_).astype(f) + (loc[:, 1] * c).astype(f)).astype(f) + (loc[:, 2] * f(0)).astype(f)
        hx, hy = (gg[:, 4] / f(2)).astype(f), (gg[:, 3] / f(2)).astype(f)                            # "lw": x vs dy, y vs dx
        inside = ((xr <= hx) & (xr >= -hx) & (yr <= hy) & (yr >= -hy)).reshape(k, m)
        is_pos &= inside
        best_v = np.full(n, -np.inf, f)
        best_g = np.zeros(n, np.int64)
        for kk in range(k):                                        # ious_inf.max(dim=1): highest IoU, then lowest GT index
            for j in range(m):
                if is_pos[kk, j]:
                    i, v = idx[kk, j], ious[idx[kk, j], j]
                    if v > best_v[i] or (v == best_v[i] and j < best_g[i]):
                        best_v[i], best_g[i] = v, j
        amax = np.argmax(ious, axis=0)                                                               # l.126, first maximum
        for j in range(m):                                                                           # l.127-128
            best_g[amax[j]], best_v[amax[j]] = j, ious[amax[j], j]
        labels = cls[best_g].astype(f)
        labels[np.isneginf(best_v)] = 0
        pos = labels > 0
        tgt, w = np.zeros((n, 7), f), np.zeros(n, f)
        if pos.any():
            tgt[pos] = self.residual_encode(g[best_g[pos]], a[pos])
            w[pos] = 1
        return labels, tgt, w

    # ---- dense BEV convs ---------------------------------------------------------------------
    def conv2d(self, x, w, bias=None, stride=1, pad=1):
        x = _f32(x)
        w = _f32(w)
        b, cin, h, wd = x.shape
        cout, _, kh, kw = w.shape
        ho = (h + 2 * pad - kh) // stride + 1
        wo = (wd + 2 * pad - kw) // stride + 1
        out = np.empty((b, cout, ho, wo), np.float32)
        bb = _f32(bias) if bias is not None else None
        self._check(self.lib.cpd_ref_conv2d(_fp(x), b, cin, h, wd, _fp(w), _fp(bb) if bb is not None else None,
                                             cout, kh, kw, stride, pad, _fp(out)), "conv2d")
        return out

    def deconv2d(self, x, w, k):
        x = _f32(x)
        w = _f32(w)
        b, cin, h, wd = x.shape
        cout = w.shape[1]
        out = np.empty((b, cout, h * k, wd * k), np.float32)
        self._check(self.lib.cpd_ref_deconv2d(_fp(x), b, cin, h, wd, _fp(w), cout, k, _fp(out)), "deconv2d")
        return out

    def bn_relu(self, x, gamma, beta, mean, var, eps, relu=True):
        x = _f32(x).copy()
        b, c = x.shape[:2]
        hw = int(np.prod(x.shape[2:]))
        self._check(self.lib.cpd_ref_bn_relu_nchw(_fp(x), b, c, hw, _fp(_f32(gamma)), _fp(_f32(beta)),
                                                   _fp(_f32(mean)), _fp(_f32(var)), ctypes.c_float(eps), int(relu)),
                    "bn_relu")
        return x

    # ---- decode ------------------------------------------------------------------------------
    def topk(self, v, k):
        v = _f32(v).ravel()
        s = np.empty(k, np.float32)
        i = np.empty(k, np.int32)
        self._check(self.lib.cpd_ref_topk(_fp(v), v.size, k, _fp(s), _ip(i)), "topk")
        return s, i

    def center_decode(self, hm, center, center_z, dim, rot, K, stride, voxel_xy, range_lo_xy, limit_range,
                      score_thresh):
        hm = _f32(hm)
        nc, h, w = hm.shape
        boxes = np.empty((K, 7), np.float32)
        scores = np.empty(K, np.float32)
        labels = np.empty(K, np.int32)
        n = ctypes.c_int32(0)
        self._check(self.lib.cpd_ref_center_decode(
            _fp(hm), _fp(_f32(center)), _fp(_f32(center_z)), _fp(_f32(dim)), _fp(_f32(rot)), nc, h, w, K,
            ctypes.c_float(stride), _farr(voxel_xy), _farr(range_lo_xy), _farr(limit_range),
            ctypes.c_float(score_thresh), _fp(boxes), _fp(scores), _ip(labels), ctypes.byref(n)), "center_decode")
        return boxes[:n.value].copy(), scores[:n.value].copy(), labels[:n.value].copy()

    # ---- B3 iou / nms ------------------------------------------------------------------------
    def _pair(self, fn, a, b):
        a = _f32(a)
        b = _f32(b)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self._check(fn(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out)), "pairwise")
        return out

    def boxes_overlap_bev(self, a, b):
        return self._pair(self.lib.cpd_ref_boxes_overlap_bev, a, b)

    def boxes_iou_bev(self, a, b):
        return self._pair(self.lib.cpd_ref_boxes_iou_bev, a, b)

    def boxes_iou3d(self, a, b):
        return self._pair(self.lib.cpd_ref_boxes_iou3d, a, b)

    def _nms(self, fn, boxes, thr):
        boxes = _f32(boxes)
        n = boxes.shape[0]
        keep = np.empty(max(n, 1), np.int64)
        k = ctypes.c_int32(0)
        self._check(fn(_fp(boxes), n, ctypes.c_float(thr), keep.ctypes.data_as(c_i64_p), ctypes.byref(k)), "nms")
        return keep[:k.value].copy()

    def nms(self, boxes_sorted, thr):
        return self._nms(self.lib.cpd_ref_nms, boxes_sorted, thr)

    def nms_normal(self, boxes_sorted, thr):
        return self._nms(self.lib.cpd_ref_nms_normal, boxes_sorted, thr)
