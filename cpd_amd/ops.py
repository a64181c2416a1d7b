"""Tensor-level wrappers of the C-ABI (include/cpd_hip.h): allocate outputs / workspaces with torch,
pass raw device pointers and the current HIP stream. PyTorch is plumbing here (device memory,
streams); every computation happens in libcpd_hip.so.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, farr, iarr, lib, ptr, stream


def _dev(t):
    return t.device


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.CpdHipError("%s must be a device (HIP) tensor" % name)


# ---------------------------------------------------------------------------------------- B1
def voxel_grid_size(vsize_xyz, range_xyz):
    g = (ctypes.c_int32 * 3)()
    check(lib().cpd_voxel_grid_size(farr(vsize_xyz), farr(range_xyz), g), "cpd_voxel_grid_size")
    return [g[0], g[1], g[2]]


class Voxelizer:
    """Device voxelizer with a persistent workspace (sized for `max_points_in` points)."""

    def __init__(self, vsize_xyz, range_xyz, num_point_features, max_points_per_voxel, max_voxels, device="cuda"):
        self.vs = [float(v) for v in vsize_xyz]
        self.rg = [float(v) for v in range_xyz]
        self.c = int(num_point_features)
        self.P = int(max_points_per_voxel)
        self.max_voxels = int(max_voxels)
        self.device = torch.device(device)
        self.grid_zyx = voxel_grid_size(self.vs, self.rg)
        self._ws = None
        self._wsb = None
        self._ws_n = -1

    def _workspace(self, n):
        if self._ws is None or n > self._ws_n:
            nbytes = lib().cpd_voxelize_workspace_bytes(n, self.P, self.max_voxels, farr(self.vs), farr(self.rg))
            if nbytes == 0:
                raise _lib.CpdHipError("cpd_voxelize_workspace_bytes: unsupported geometry")
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_n = n
        return self._ws

    def __call__(self, points, batch_idx=0, coord_cols=4, want_voxels=True, want_mean=True, sync=True):
        """points [N, C] f32 device tensor. Returns (voxels|None, coords, num_points, mean|None, n)
        where n is a python int (sync=True, rows sliced) or the device scalar (sync=False, capacity
        sized rows)."""
        _need_cuda(points, "points")
        points = points.contiguous()
        n, c = points.shape
        assert c == self.c and points.dtype == torch.float32
        cap = max(1, min(self.max_voxels, n))
        dev = points.device
        voxels = torch.empty((cap, self.P, c), dtype=torch.float32, device=dev) if want_voxels else None
        coords = torch.empty((cap, coord_cols), dtype=torch.int32, device=dev)
        num = torch.empty((cap,), dtype=torch.int32, device=dev)
        mean = torch.empty((cap, c), dtype=torch.float32, device=dev) if want_mean else None
        nvox = torch.zeros((1,), dtype=torch.int32, device=dev)
        ws = self._workspace(n)
        check(lib().cpd_voxelize(ptr(points), n, c, farr(self.vs), farr(self.rg), self.P, self.max_voxels,
                                 int(batch_idx), int(coord_cols), ptr(voxels), ptr(coords), ptr(num), ptr(mean),
                                 ptr(nvox), ptr(ws), ws.numel(), stream()), "cpd_voxelize")
        if not sync:
            return voxels, coords, num, mean, nvox
        m = int(nvox.item())
        return (voxels[:m] if voxels is not None else None, coords[:m], num[:m],
                mean[:m] if mean is not None else None, m)


    def batch_supported(self, n_frames, z_extra=0):
        g = voxel_grid_size(self.vs, self.rg)
        cells = (g[0] + z_extra) * g[1] * g[2]
        return 0 < n_frames <= 64 and cells < (1 << 31) and n_frames * cells < (1 << 40)

    def batch(self, points_list, want_voxels=False, want_mean=True, index_z_extra=None, canonical=False):
        """All frames in one set of launches (cpd_voxelize_batch). Returns capacity-sized device tensors
        (voxels|None, coords [cap,4] (b,z,y,x), num_points, mean|None, n_voxels [B+1] = per frame + total);
        rows of frame f follow those of frame f-1. No host synchronisation.
        index_z_extra = k: also returns, as a sixth value, the SiteIndex of the voxel list over the grid with k more
        z-levels (the backbone's sparse_shape for k = 1), built by the voxelizer itself (cpd_voxelize_batch_index).
        canonical = True (with index_z_extra): rows in ascending (frame, z, y, x) order instead of first appearance, canonical
        index, max_voxels cap not applied (cpd_voxelize_batch_canonical; the caller checks n_voxels against the cap)."""
        for p in points_list:
            _need_cuda(p, "points")
        # the frames as they lie (cpd_voxelize_batch_frames: one device pointer per frame) -- no concatenated copy of the batch's points
        frames = [p.contiguous() for p in points_list]
        nf = len(frames)
        offs = [0]
        for p in frames:
            assert p.dim() == 2 and p.shape[1] == self.c and p.dtype == torch.float32
            offs.append(offs[-1] + p.shape[0])
        n, c = offs[-1], self.c
        cap = max(1, min(self.max_voxels * nf, n))
        dev = frames[0].device
        fptrs = (ctypes.c_void_p * nf)(*[ctypes.c_void_p(p.data_ptr()) for p in frames])
        voxels = torch.empty((cap, self.P, c), dtype=torch.float32, device=dev) if want_voxels else None
        coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        num = torch.empty((cap,), dtype=torch.int32, device=dev)
        mean = torch.empty((cap, c), dtype=torch.float32, device=dev) if want_mean else None
        nvox = torch.zeros((nf + 1,), dtype=torch.int32, device=dev)
        nbytes = lib().cpd_voxelize_batch_workspace_bytes(n, nf, self.P, self.max_voxels, farr(self.vs), farr(self.rg))
        if nbytes == 0:
            raise _lib.CpdHipError("cpd_voxelize_batch: unsupported batch (%d frames)" % nf)
        if self._wsb is None or self._wsb.numel() < nbytes:
            self._wsb = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if index_z_extra is not None:
            g = self.grid_zyx
            index = SiteIndex(nf, [g[0] + int(index_z_extra), g[1], g[2]], max(n, 1), dev)
            check(lib().cpd_voxelize_batch_frames(fptrs, iarr(offs), nf, c, farr(self.vs), farr(self.rg), self.P, self.max_voxels,
                                                  ptr(voxels), ptr(coords), ptr(num), ptr(mean), ptr(nvox), ptr(self._wsb),
                                                  self._wsb.numel(), ptr(index.buf), index.buf.numel(), int(index_z_extra),
                                                  1 if canonical else 0, stream()), "cpd_voxelize_batch_frames")
            return voxels, coords, num, mean, nvox, index
        assert not canonical, "canonical rows come with the in-place index (index_z_extra)"
        check(lib().cpd_voxelize_batch_frames(fptrs, iarr(offs), nf, c, farr(self.vs), farr(self.rg), self.P, self.max_voxels,
                                              ptr(voxels), ptr(coords), ptr(num), ptr(mean), ptr(nvox), ptr(self._wsb),
                                              self._wsb.numel(), None, 0, 0, 0, stream()), "cpd_voxelize_batch_frames")
        return voxels, coords, num, mean, nvox


# ---------------------------------------------------------------------------------------- B2
class SiteIndex:
    """Occupancy bitmap + popcount prefix of a site set (see csrc/site_index.hip)."""

    def __init__(self, batch, shape_zyx, capacity, device):
        self.batch = int(batch)
        self.shape = [int(s) for s in shape_zyx]
        nbytes = lib().cpd_index_bytes(self.batch, iarr(self.shape), int(capacity))
        if nbytes == 0:
            raise _lib.CpdHipError("cpd_index_bytes: unsupported grid %s x batch %d" % (self.shape, self.batch))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)

    @staticmethod
    def build(indices, batch, shape_zyx):
        _need_cuda(indices, "indices")
        indices = indices.contiguous()
        assert indices.dtype == torch.int32 and indices.shape[1] == 4
        idx = SiteIndex(batch, shape_zyx, indices.shape[0], indices.device)
        check(lib().cpd_index_build(ptr(indices), indices.shape[0], idx.batch, iarr(idx.shape), ptr(idx.buf),
                                    idx.buf.numel(), stream()), "cpd_index_build")
        return idx


    def set_order(self, rank_to_row):
        """Install a caller-owned rank -> row map (cpd_index_set_order; None = canonical again): lookups through this index
        then return rows in that order. The tensor is kept alive here."""
        self.order = rank_to_row.contiguous() if rank_to_row is not None else None
        check(lib().cpd_index_set_order(ptr(self.buf), ptr(self.order), stream()), "cpd_index_set_order")
        return self


def order_rows_by_taps(indices, index, ksize=(3, 3, 3), chunk_rows=4096):
    """For a level's canonical site list [n, 4] and its SiteIndex: (indices in the new order, new_to_old, old_to_new) -- the rows
    of every `chunk_rows` consecutive rows sorted by their sub-manifold neighbour pattern (cpd_order_rows_by_taps)."""
    indices = indices.contiguous()
    n = indices.shape[0]
    dev = indices.device
    new_to_old = torch.empty((n,), dtype=torch.int32, device=dev)
    old_to_new = torch.empty((n,), dtype=torch.int32, device=dev)
    out = torch.empty_like(indices)
    ws = torch.empty((max(n, 1) * 4,), dtype=torch.uint8, device=dev)
    check(lib().cpd_order_rows_by_taps(ptr(indices), n, index.batch, iarr(index.shape), iarr(ksize), ptr(index.buf), int(chunk_rows),
                                       ptr(new_to_old), ptr(old_to_new), ptr(out), ptr(ws), ws.numel(), stream()), "cpd_order_rows_by_taps")
    return out, new_to_old, old_to_new


def order_rows_bricks(indices, index, brick=(8, 8), tile_rows=128):
    """For a level's CANONICAL site list [n, 4] and its canonical SiteIndex: (indices in brick order, new_to_old, old_to_new) --
    rows sorted by (b, z, y / brick[0], x / brick[1], y, x), then by neighbour pattern inside every 128-row tile
    (cpd_order_rows_bricks): the row order the staged row-wave kernel (gather_conv with a planned rulebook) wants."""
    indices = indices.contiguous()
    n = indices.shape[0]
    dev = indices.device
    new_to_old = torch.empty((n,), dtype=torch.int32, device=dev)
    old_to_new = torch.empty((n,), dtype=torch.int32, device=dev)
    out = torch.empty_like(indices)
    a = lambda b: (b + 255) // 256 * 256
    ws = torch.empty((2 * a(4 * max(n, 1)) + a(16 * max(n, 1)),), dtype=torch.uint8, device=dev)
    check(lib().cpd_order_rows_bricks(ptr(indices), n, index.batch, iarr(index.shape), ptr(index.buf), int(brick[0]), int(brick[1]), int(tile_rows),
                                      ptr(new_to_old), ptr(old_to_new), ptr(out), ptr(ws), ws.numel(), stream()), "cpd_order_rows_bricks")
    return out, new_to_old, old_to_new


def rulebook_plan(nbr, tile_rows=128):
    """Attach the row plan of a 27-tap sub-manifold rulebook (cpd_rulebook_plan; tiles of 128 or 256 rows) to the table: gather_conv
    then runs the staged row-wave kernel on it where that kernel applies. Returns nbr."""
    kv, n = nbr.shape
    dev = nbr.device
    t = int(tile_rows)
    slots = torch.empty((lib().cpd_rulebook_plan_bytes(n, t, 0) // 2,), dtype=torch.int16, device=dev)
    ulist = torch.empty((lib().cpd_rulebook_plan_bytes(n, t, 1) // 4,), dtype=torch.int32, device=dev)
    count = torch.empty((lib().cpd_rulebook_plan_bytes(n, t, 2) // 4,), dtype=torch.int32, device=dev)
    check(lib().cpd_rulebook_plan(ptr(nbr), kv, n, t, ptr(slots), ptr(ulist), ptr(count), stream()), "cpd_rulebook_plan")
    nbr.plan = (slots, ulist, count, t)
    return nbr


def _new_tapmask(n, kv, device):
    if kv > 32 or n == 0:
        return None
    return torch.empty(((n + 15) // 16,), dtype=torch.int32, device=device)


# one 1024-thread workgroup per 4096-row chunk: below ~200 chunks (one- and four-frame batches: 28 ... 112 chunks on the largest level)
# the chunk-wise builder leaves most of the chip idle and the row-per-lane kernel is faster (one frame 3.23 vs 3.10 ms)
CHUNKED_MIN_ROWS = 200 * 4096


def _rulebook_chunked(canonical, in_index, ksize, stride, pad, n):
    """the table of a chunk-ordered level, built chunk by chunk in canonical order (cpd_rulebook_chunk_ordered); `canonical` =
    (canonical site list, old_to_new or None, chunk_rows)"""
    out_c, o2n, chunk = canonical
    dev = out_c.device
    nbr = torch.empty((27, n), dtype=torch.int32, device=dev)
    tapmask = _new_tapmask(n, 27, dev)
    check(lib().cpd_rulebook_chunk_ordered(ptr(out_c.contiguous()), ptr(o2n), n, in_index.batch, iarr(in_index.shape), iarr(ksize), iarr(stride),
                                           iarr(pad), ptr(in_index.buf), int(chunk), ptr(nbr), ptr(tapmask), stream()), "cpd_rulebook_chunk_ordered")
    nbr.tapmask = tapmask
    return nbr


def rulebook_subm(indices, index, ksize=(3, 3, 3), canonical=None):
    """`canonical` = (canonical site list, old_to_new, chunk_rows) of a level whose `indices` are chunk-ordered (order_rows_by_taps,
    chunk_rows = 4096): same table, built by the chunk-wise kernel (3 x 3 x 3 only)."""
    if canonical is not None and tuple(ksize) == (3, 3, 3) and canonical[2] == 4096 and indices.shape[0] >= CHUNKED_MIN_ROWS:
        return _rulebook_chunked(canonical, index, ksize, (1, 1, 1), (1, 1, 1), indices.shape[0])
    indices = indices.contiguous()
    n = indices.shape[0]
    kv = int(ksize[0] * ksize[1] * ksize[2])
    nbr = torch.empty((kv, n), dtype=torch.int32, device=indices.device)
    tapmask = _new_tapmask(n, kv, indices.device)
    check(lib().cpd_rulebook_subm(ptr(indices), n, index.batch, iarr(index.shape), iarr(ksize), ptr(index.buf),
                                  ptr(nbr), ptr(tapmask), stream()), "cpd_rulebook_subm")
    nbr.tapmask = tapmask      # travels with the table; cpd_gather_conv skips empty (row group, tap) pairs with it
    return nbr


def conv3x3_rows_tile(frames, h, w, c_in, c_out, math="f16x2"):
    """(bm, bn) of the rulebook-free window kernel for a 3 x 3 / stride 1 / pad 1 layer over frames x h x w pixel rows, or None when the
    layer takes the table path (cpd_conv3x3_rows_tile)"""
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().cpd_conv3x3_rows_tile(int(frames), int(h), int(w), int(c_in), int(c_out), CONV_MATH[math], ctypes.byref(bm), ctypes.byref(bn))
    return (bm.value, bn.value) if rc == 0 else None


def conv_out_shape(in_shape, ksize, stride, pad):
    o = (ctypes.c_int32 * 3)()
    check(lib().cpd_conv_out_shape(iarr(in_shape), iarr(ksize), iarr(stride), iarr(pad), o), "cpd_conv_out_shape")
    return [o[0], o[1], o[2]]


def conv_outset_begin(in_indices, batch, in_shape, ksize, stride, pad):
    """First half of conv_outset: the launches that mark and count a SparseConv3d's active output sites (cpd_conv_outset). Returns a
    handle for conv_outset_end -- which reads the count back; queueing other work between the two hides the read-back's wait."""
    in_indices = in_indices.contiguous()
    out_shape = conv_out_shape(in_shape, ksize, stride, pad)
    out_index = SiteIndex(batch, out_shape, 0, in_indices.device)
    n_out_dev = torch.zeros((1,), dtype=torch.int32, device=in_indices.device)
    check(lib().cpd_conv_outset(ptr(in_indices), in_indices.shape[0], batch, iarr(in_shape), iarr(ksize),
                                iarr(stride), iarr(pad), ptr(out_index.buf), out_index.buf.numel(), ptr(n_out_dev),
                                stream()), "cpd_conv_outset")
    return (in_indices, batch, out_shape, out_index, n_out_dev)


def conv_outset_end(handle):
    """Second half: (out_indices [n_out,4] i32 in canonical (b,z,y,x) order, out_index SiteIndex, out_shape)."""
    in_indices, batch, out_shape, out_index, n_out_dev = handle
    n_out = int(n_out_dev.item())  # the one host sync of a strided layer: sizes the output tensors
    out_indices = torch.empty((n_out, 4), dtype=torch.int32, device=in_indices.device)
    check(lib().cpd_index_emit(ptr(out_index.buf), batch, iarr(out_shape), ptr(out_indices), n_out, stream()),
          "cpd_index_emit")
    return out_indices, out_index, out_shape


def conv_outset(in_indices, batch, in_shape, ksize, stride, pad):
    """Active output sites of a SparseConv3d in canonical (b,z,y,x) order.
    Returns (out_indices [n_out,4] i32, out_index SiteIndex, out_shape)."""
    return conv_outset_end(conv_outset_begin(in_indices, batch, in_shape, ksize, stride, pad))


def rulebook_conv(out_indices, in_index, ksize, stride, pad, canonical=None):
    if canonical is not None and tuple(ksize) == (3, 3, 3) and canonical[2] == 4096 and out_indices.shape[0] >= CHUNKED_MIN_ROWS:
        return _rulebook_chunked(canonical, in_index, ksize, stride, pad, out_indices.shape[0])
    out_indices = out_indices.contiguous()
    n_out = out_indices.shape[0]
    kv = int(ksize[0] * ksize[1] * ksize[2])
    nbr = torch.empty((kv, n_out), dtype=torch.int32, device=out_indices.device)
    tapmask = _new_tapmask(n_out, kv, out_indices.device)
    check(lib().cpd_rulebook_conv(ptr(out_indices), n_out, in_index.batch, iarr(in_index.shape), iarr(ksize),
                                  iarr(stride), iarr(pad), ptr(in_index.buf), ptr(nbr), ptr(tapmask), stream()),
          "cpd_rulebook_conv")
    nbr.tapmask = tapmask
    return nbr


def pack_weight(w_kio, out=None):
    """w_kio: [kv, c_in, c_out] f32 device tensor -> packed MFMA-fragment weights."""
    _need_cuda(w_kio, "weight")
    w_kio = w_kio.contiguous().float()
    kv, cin, cout = w_kio.shape
    n = lib().cpd_packed_weight_floats(kv, cin, cout)
    packed = out if out is not None else torch.empty((n,), dtype=torch.float32, device=w_kio.device)
    assert packed.numel() == n and packed.is_contiguous()
    check(lib().cpd_pack_weight(ptr(w_kio), kv, cin, cout, ptr(packed), stream()), "cpd_pack_weight")
    return packed


CONV_MATH = {"f32": 0, "bf16x3": 2, "f16x2": 4}     # CPD_GC_BF16X3 / CPD_GC_F16X2 of include/cpd_hip.h


def _gc_flags(dense, bf16x3, math):
    if math is not None:
        return (1 if dense else 0) | CONV_MATH[math]
    return (1 if dense else 0) | (2 if bf16x3 else 0)


def gather_conv(inp, c_in, packed_w, nbr, kv, n_out, c_out, scale=None, shift=None, residual=None, relu=False,
                out=None, out_row_map=None, out_col_group=0, dense=False, bf16x3=False, math=None, in_absmax=None, out_absmax=None,
                guard=False, in_pairs=False, out_pairs=False, res_pairs=False):
    """out[j,:c_out] = act((sum_t in[nbr[t][j]] . W[t]) * scale + shift + residual[j]).
    `inp` / `out` / `residual` are 2-D row tensors whose row stride may exceed the channel count.
    `math`: "f32" | "bf16x3" | "f16x2" (overrides the older `bf16x3` switch).
    `in_absmax`: the absmax block of `inp` (int32 device tensor: absmax_blocks / absmax_rows here, train_ops.bn_backward for
    gradients): the split-fp16 kernels then pre-scale `inp` into fp16's range by a power of two (exact) -- the range guard of
    the f16x2 inference path, and how gradients take that path. `out_absmax`: a block this launch raises to max |out| (zeroed by
    the caller; the next layer's `in_absmax`). `guard=True`: f16x2 with no `in_absmax` given measures the input first
    (one extra pass over `inp`) instead of trusting it to stay below 65504.
    `in_pairs` / `out_pairs` / `res_pairs`: the rows of `inp` / `out` / `residual` are fp16-PAIR rows (CPD_GC_*_PAIRS of
    include/cpd_hip.h: per 32-channel block the fp16 high terms, then the fp16 low terms -- rows_to_pairs / pairs_to_rows here):
    storage between the engine's f16x2 sparse layers; with `dense` (round 5) the BEV maps between split-fp16 dense layers: the window
    kernel's 128 x 128 / 256 x 64 / 256 x 16 tiles and the 128 x 128 tile kernel (window_conv_f16p_kernel / tile_conv_f16p_kernel) --
    a shape they do not take raises (CPD_ERR_UNSUPPORTED): nothing else reads dense pair rows."""
    _need_cuda(inp, "inp")
    assert inp.dim() == 2 and inp.stride(1) == 1
    if guard and in_absmax is None and (math == "f16x2") and c_in % 32 == 0:
        in_absmax = absmax_rows(inp, c_in)
    if out is None:
        out = torch.empty((n_out, c_out), dtype=torch.float32, device=inp.device)
    assert out.dim() == 2 and out.stride(1) == 1
    res_ld = 0
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1
        res_ld = residual.stride(0)
    flags = _gc_flags(dense, bf16x3, math) | (16 if in_pairs else 0) | (32 if out_pairs else 0) | (64 if res_pairs else 0)
    image = getattr(nbr, "image", None)             # 3x3 / stride 1 / pad 1 pixel table: the rulebook-free window kernel
    # (pair rows: dense maps only -- cpd_conv3x3_rows takes CPD_GC_IN_PAIRS / OUT_PAIRS on the tiles of large batches)
    if image is not None and (dense or not (flags & 0x70)) and out_row_map is None and kv == 9 and n_out == image[0] * image[1] * image[2] and inp.shape[0] == n_out:
        rc = lib().cpd_conv3x3_rows_ranged(
            ctypes.c_void_p(inp.data_ptr()), inp.stride(0), image[0], image[1], image[2], c_in, ptr(packed_w), c_out,
            ptr(scale), ptr(shift), ctypes.c_void_p(residual.data_ptr()) if residual is not None else None, res_ld,
            int(bool(relu)), ctypes.c_void_p(out.data_ptr()), out.stride(0), flags, ptr(in_absmax), ptr(out_absmax), stream())
        if rc != -4:                                # CPD_ERR_UNSUPPORTED: shape / alignment / size -> the table path below
            check(rc, "cpd_conv3x3_rows")
            return out
    plan = getattr(nbr, "plan", None)               # a planned sub-manifold rulebook: the staged row-wave kernel, where it applies
    if plan is not None and in_absmax is None and out_row_map is None and not out_col_group and lib().cpd_gather_conv_planned_supported(
            int(inp.shape[0]), int(n_out), int(c_in), int(c_out), int(inp.stride(0)), int(kv), flags):
        check(lib().cpd_gather_conv_planned(
            ctypes.c_void_p(inp.data_ptr()), inp.stride(0), inp.shape[0], c_in, ptr(packed_w), ptr(getattr(nbr, "tapmask", None)),
            ptr(plan[0]), ptr(plan[1]), ptr(plan[2]), plan[3], kv, n_out, c_out, ptr(scale), ptr(shift),
            ctypes.c_void_p(residual.data_ptr()) if residual is not None else None, res_ld, int(bool(relu)),
            ctypes.c_void_p(out.data_ptr()), out.stride(0), flags, ptr(out_absmax), stream()), "cpd_gather_conv_planned")
        return out
    ws, ws_bytes = None, 0
    if (flags & 6) and c_in % 32 == 0 and not out_col_group:
        # a small launch (one frame, the train step) deals its taps / stages to several workgroups through a workspace (cpd_gather_conv_ws)
        ws_bytes = int(lib().cpd_gather_conv_split_bytes(int(n_out), int(c_in), int(c_out), int(inp.stride(0)), int(kv), flags))
        if ws_bytes:
            ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=inp.device)
    check(lib().cpd_gather_conv_ws(
        ctypes.c_void_p(inp.data_ptr()), inp.stride(0), inp.shape[0], c_in, ptr(packed_w),
        ptr(nbr), ptr(getattr(nbr, "tapmask", None)), kv, n_out, c_out, ptr(scale), ptr(shift),
        ctypes.c_void_p(residual.data_ptr()) if residual is not None else None, res_ld, int(bool(relu)),
        ctypes.c_void_p(out.data_ptr()), out.stride(0), ptr(out_row_map), int(out_col_group), flags,
        ptr(in_absmax), ptr(out_absmax), ptr(ws), ws_bytes, stream()),
        "cpd_gather_conv")
    return out


def rows_to_pairs(x):
    """fp32 rows [n, C] (C % 32 == 0) -> the same bytes as fp16-pair rows (a float32 tensor of the same shape holding, per 32-channel
    block, 32 fp16 high terms then 32 fp16 low terms of x = h + l; CPD_GC_*_PAIRS). torch ops: tests and tools, not the hot path."""
    n, c = x.shape
    blk = 4 if c == 16 else 32          # 16-channel rows: per k-group of 4 channels [hi 4 | lo 4] (the K = 16 MFMA's pieces)
    assert c % blk == 0 and (c == 16 or c % 32 == 0) and x.dtype == torch.float32
    h = x.to(torch.float16)
    l = (x - h.float()).to(torch.float16)
    pairs = torch.stack([h.view(n, c // blk, blk), l.view(n, c // blk, blk)], dim=2)   # [n, blocks, 2, blk] fp16
    return pairs.contiguous().view(n, c * 2).view(torch.float32)


def pairs_to_rows(x, c=None):
    """fp16-pair rows -> fp32 rows (h + l, exact in fp32)."""
    n = x.shape[0]
    c = x.shape[1] if c is None else c
    blk = 4 if c == 16 else 32
    assert (c == 16 or c % 32 == 0) and x.dtype == torch.float32
    halves = x[:, :c].contiguous().view(torch.float16).view(n, c // blk, 2, blk).float()
    return (halves[:, :, 0] + halves[:, :, 1]).reshape(n, c)


class PairRows:
    """fp16-pair rows handed from an engine to a consumer that reads the stored bits (the RoI pooling's first GEMM): `rows` is a
    contiguous [N, C] float32-typed tensor whose BYTES are, per 32-channel block, [32 high halves | 32 low halves] -- not fp32 values.
    The layout travels as this type, not as an attribute on the tensor (.contiguous() / slicing / .to() return a new
    tensor and silently dropped the attribute; the bytes would then have been read as fp32)."""
    __slots__ = ("rows",)

    def __init__(self, rows):
        assert rows.dtype == torch.float32 and rows.dim() == 2 and rows.is_contiguous() and (rows.shape[1] % 32 == 0 or rows.shape[1] == 16), \
            "PairRows: contiguous [N, C] fp16-pair rows, C a multiple of 32 (or the 16-channel level's [hi 4 | lo 4] k-group form)"
        self.rows = rows

    @property
    def shape(self):
        return self.rows.shape

    @property
    def device(self):
        return self.rows.device

    def float_rows(self):
        """the fp32 values (h + l, exact)"""
        return pairs_to_rows(self.rows)


class launch_log:
    """with ops.launch_log() as log: ...; log.counts -> {kernel instantiation: launches} of the conv / weight-gradient calls made
    inside (cpd_launch_log_*; diagnostics for tests and tools)."""

    def __enter__(self):
        lib().cpd_launch_log_enable(1)
        self.counts = {}
        return self

    def __exit__(self, *exc):
        n = lib().cpd_launch_log_dump(None, 0)
        buf = ctypes.create_string_buffer(int(n))
        lib().cpd_launch_log_dump(buf, n)
        lib().cpd_launch_log_enable(0)
        for ln in buf.value.decode().splitlines():
            name, cnt = ln.rsplit(" ", 1)
            self.counts[name] = int(cnt)


ABSMAX_WORDS = 16 * 32       # CPD_ABSMAX_WORDS of include/cpd_hip.h: 16 words, one per 128-byte line


def absmax_blocks(n, device):
    """n zeroed absmax blocks [n, ABSMAX_WORDS] int32: row i is the block of one activation tensor (bits of its max |value|)."""
    return torch.zeros((n, ABSMAX_WORDS), dtype=torch.int32, device=device)


def absmax_rows(x, c=None, block=None):
    """The absmax block of a row tensor no kernel of this library produced (cpd_absmax_rows: one pass over x[:, :c])."""
    _need_cuda(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    if block is None:
        block = torch.zeros((ABSMAX_WORDS,), dtype=torch.int32, device=x.device)
    check(lib().cpd_absmax_rows(ctypes.c_void_p(x.data_ptr()), x.stride(0), x.shape[0], int(c if c is not None else x.shape[1]), ptr(block),
                                stream()), "cpd_absmax_rows")
    return block


def tag_range(t, block):
    """Attach the absmax block a kernel filled for tensor `t` (attribute `_cpd_rb`) together with the tensor's version counter: an
    in-place operation on `t` (or on a view sharing its storage) afterwards bumps the counter and retires the tag."""
    if block is not None:
        t._cpd_rb = (block, t._version)
    return t


def tagged_range(t):
    """the block attached by tag_range, or None when there is none or `t` was written in place since (stale maximum: ADVICE r3)"""
    tag = getattr(t, "_cpd_rb", None)
    if tag is None:
        return None
    block, ver = tag
    return block if ver == t._version else None


def epilogue_shift(scale, shift, bias):
    """The conv epilogue computes acc * scale + shift. A caller that gives no shift but whose conv has a bias means
    (acc + bias) * scale: shift = bias * scale (bias alone when there is no scale)."""
    if shift is not None or bias is None:
        return shift
    b = bias.detach().float()
    return b * scale if scale is not None else b


def range_block(t, c=None):
    """The absmax block travelling with tensor `t` (tag_range, set by the layer that produced it) when still valid, else measured now."""
    b = tagged_range(t)
    if b is not None:
        return b
    t2 = t if t.dim() == 2 else t.reshape(-1, t.shape[-1])
    return absmax_rows(t2.contiguous().float(), c)


def absmax_value(block):
    """max |value| recorded in a block, as a Python float (a host read: tests and diagnostics)."""
    return float(block.view(-1)[::32][:16].max().view(torch.float32) if block.dtype == torch.int32 else block.max())


def densify_nchw(feat, indices, batch, shape_zyx):
    feat = feat.contiguous()
    n, c = feat.shape
    out = torch.empty((batch, c * shape_zyx[0], shape_zyx[1], shape_zyx[2]), dtype=torch.float32, device=feat.device)
    check(lib().cpd_densify_nchw(ptr(feat), ptr(indices.contiguous()), n, c, batch, iarr(shape_zyx), ptr(out),
                                 stream()), "cpd_densify_nchw")
    return out


def densify_nhwc(feat, indices, batch, shape_zyx, out=None):
    feat = feat.contiguous()
    n, c = feat.shape
    if out is None:
        out = torch.empty((batch, shape_zyx[1], shape_zyx[2], shape_zyx[0] * c), dtype=torch.float32,
                          device=feat.device)
    check(lib().cpd_densify_nhwc(ptr(feat), ptr(indices.contiguous()), n, c, batch, iarr(shape_zyx), ptr(out),
                                 stream()), "cpd_densify_nhwc")
    return out


class DenseMap:
    """A persistent, pre-zeroed (B, H, W, D * C) map for densify_nhwc: `scatter` writes the occupied rows into it, `clear` -- queued after
    the map's last reader -- zeroes the same rows again (cpd_densify_nhwc_rows / _clear). A map left dirty (an exception between the
    two) is cleared in full on its next use. ONE map serves every batch size up to its capacity: the leading `batch` frames of the
    (B, H, W, D * C) layout are a valid map of that batch, and scatter / clear touch occupied rows only (a map per batch
    size pinned the sum of the sizes seen -- 36 MB per frame)."""

    def __init__(self, batch, shape_zyx, c, device):
        self.batch, self.shape, self.c = int(batch), [int(v) for v in shape_zyx], int(c)       # batch = the capacity in frames
        self.buf = torch.zeros((self.batch, self.shape[1], self.shape[2], self.shape[0] * self.c), dtype=torch.float32, device=device)
        self.dirty = None                # (index list, frames) scattered and not yet cleared

    def scatter(self, feat, indices, batch=None):
        """-> the (batch, H, W, D * C) map (a view of the leading frames); batch <= capacity, default the capacity"""
        batch = self.batch if batch is None else int(batch)
        assert 1 <= batch <= self.batch
        if self.dirty is not None:
            self.buf.zero_()
        feat, indices = feat.contiguous(), indices.contiguous()
        assert feat.shape[1] == self.c
        self.dirty = (indices, batch)
        check(lib().cpd_densify_nhwc_rows(ptr(feat), ptr(indices), feat.shape[0], self.c, batch, iarr(self.shape), ptr(self.buf), stream()),
              "cpd_densify_nhwc_rows")
        return self.buf[:batch]

    def clear(self):
        if self.dirty is None:
            return
        indices, batch = self.dirty
        check(lib().cpd_densify_nhwc_clear(ptr(indices), indices.shape[0], self.c, batch, iarr(self.shape), ptr(self.buf), stream()),
              "cpd_densify_nhwc_clear")
        self.dirty = None


def densify_nhwc_cd(feat, indices, batch, shape_zyx):
    """(B, H, W, C*D) with channel = c*D + z: the reference's (B, C*D, H, W) map in channels_last memory."""
    feat = feat.contiguous()
    n, c = feat.shape
    out = torch.empty((batch, shape_zyx[1], shape_zyx[2], c * shape_zyx[0]), dtype=torch.float32, device=feat.device)
    check(lib().cpd_densify_nhwc_cd(ptr(feat), ptr(indices.contiguous()), n, c, batch, iarr(shape_zyx), ptr(out), stream()),
          "cpd_densify_nhwc_cd")
    return out


_PIXEL_TABLES = {}


def rulebook_conv2d_cached(batch, h, w, device, k=3, stride=1, pad=1):
    """The pixel table of a k x k conv over (batch, h, w) maps, built once per shape and device."""
    key = (batch, h, w, k, stride, pad, str(device))
    if key not in _PIXEL_TABLES:
        _PIXEL_TABLES[key] = rulebook_conv2d(batch, h, w, k, k, stride, pad, device)
    return _PIXEL_TABLES[key]


def rulebook_conv2d(batch, h, w, kh, kw, stride, pad, device):
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (w + 2 * pad - kw) // stride + 1
    nbr = torch.empty((kh * kw, batch * ho * wo), dtype=torch.int32, device=device)
    check(lib().cpd_rulebook_conv2d(batch, h, w, kh, kw, stride, pad, ptr(nbr), stream()), "cpd_rulebook_conv2d")
    if (kh, kw, stride, pad) == (3, 3, 1, 1):
        nbr.image = (batch, h, w)       # gather_conv may then run the rulebook-free kernel (cpd_conv3x3_rows)
    return nbr, ho, wo


# ------------------------------------------------------------------------------------ decode
def center_decode(hm, center, center_z, dim, rot, pix_stride, ch_stride, num_class, h, w, k, feature_map_stride,
                  voxel_xy, range_lo_xy, limit_range, score_thresh, sync=True, batch=1, sample_stride=0):
    """Decode `batch` samples in one call (sample b's maps start b*sample_stride floats later).
    sync=True and batch == 1: returns (boxes[n,7], scores[n], labels[n], n) sliced on the host.
    Otherwise: capacity-sized (boxes[batch,k,7], scores[batch,k], labels[batch,k], counts[batch])
    device tensors and no host synchronisation."""
    dev = hm.device
    boxes = torch.empty((batch, k, 7), dtype=torch.float32, device=dev)
    scores = torch.empty((batch, k), dtype=torch.float32, device=dev)
    labels = torch.empty((batch, k), dtype=torch.int32, device=dev)
    n_out = torch.zeros((batch,), dtype=torch.int32, device=dev)
    ws = torch.empty(lib().cpd_center_decode_workspace_bytes(batch, num_class, h * w, k), dtype=torch.uint8, device=dev)
    check(lib().cpd_center_decode(
        ctypes.c_void_p(hm.data_ptr()), ctypes.c_void_p(center.data_ptr()), ctypes.c_void_p(center_z.data_ptr()),
        ctypes.c_void_p(dim.data_ptr()), ctypes.c_void_p(rot.data_ptr()), int(batch), int(sample_stride), pix_stride,
        ch_stride, num_class, h, w, k, float(feature_map_stride), farr(voxel_xy), farr(range_lo_xy), farr(limit_range),
        float(score_thresh), ptr(boxes), ptr(scores), ptr(labels), ptr(n_out), ptr(ws), ws.numel(), stream()),
        "cpd_center_decode")
    if not sync or batch != 1:
        return boxes, scores, labels, n_out
    n = int(n_out.item())
    return boxes[0, :n], scores[0, :n], labels[0, :n], n


def nms_batch(boxes, counts, thresh, normal=False):
    """boxes [batch, cap, 7] (descending score per sample), counts [batch] i32 on the device.
    Returns (keep [batch, cap] i64, num_keep [batch] i32), no host synchronisation."""
    boxes = boxes.contiguous()
    batch, cap = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((batch, cap), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((batch,), dtype=torch.int32, device=boxes.device)
    ws = torch.empty(batch * lib().cpd_nms_workspace_bytes(cap), dtype=torch.uint8, device=boxes.device)
    check(lib().cpd_nms_batch(ptr(boxes), ptr(counts), batch, cap, float(thresh), 1 if normal else 0, ptr(keep), ptr(num),
                              ptr(ws), ws.numel(), stream()), "cpd_nms_batch")
    return keep, num


def nms_batch_first(boxes, counts, thresh, max_keep, row_limit, normal=False):
    """nms_batch for callers that keep the first `max_keep` survivors only (cpd_nms_batch_first): the mask covers a sample's first
    `row_limit` boxes, the scan stops once max_keep survived. Returns (keep, num_keep, incomplete [batch] i32): keep[b][:min(num_keep[b],
    max_keep)] equal nms_batch's first entries; incomplete[b] = 1 -> the sample needs the full nms_batch."""
    boxes = boxes.contiguous()
    batch, cap = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((batch, cap), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((batch,), dtype=torch.int32, device=boxes.device)
    inc = torch.zeros((batch,), dtype=torch.int32, device=boxes.device)
    ws = torch.empty(batch * lib().cpd_nms_workspace_bytes(min(int(row_limit), cap)), dtype=torch.uint8, device=boxes.device)
    check(lib().cpd_nms_batch_first(ptr(boxes), ptr(counts), batch, cap, float(thresh), 1 if normal else 0, int(max_keep), int(row_limit),
                                    ptr(keep), ptr(num), ptr(inc), ptr(ws), ws.numel(), stream()), "cpd_nms_batch_first")
    return keep, num, inc


_NMS_WHERE_WS = {}


def nms_batch_where(boxes, counts, where, thresh, keep, num_keep, normal=False):
    """nms_batch for the samples whose `where` word (device i32 [batch]) is non-zero, written into the given keep / num_keep; the other
    samples keep theirs (cpd_nms_batch_where): the device-side fallback of nms_batch_first, no read-back in between."""
    boxes = boxes.contiguous()
    batch, cap = boxes.shape[0], boxes.shape[1]
    # the full-capacity mask workspace (2 MB per sample at 4096 boxes, 10 MB at TRAIN's 9000) is only TOUCHED by flagged samples:
    # one grow-only buffer per (device, stream) instead of an allocation per call (kernels of one stream run in order)
    need = batch * lib().cpd_nms_workspace_bytes(cap)
    key = (str(boxes.device), stream().value)
    ws = _NMS_WHERE_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _NMS_WHERE_WS[key] = torch.empty(need, dtype=torch.uint8, device=boxes.device)
    check(lib().cpd_nms_batch_where(ptr(boxes), ptr(counts), ptr(where), batch, cap, float(thresh), 1 if normal else 0, ptr(keep), ptr(num_keep),
                                    ptr(ws), ws.numel(), stream()), "cpd_nms_batch_where")
    return keep, num_keep


def rank_scores(cls, boxes, labels, score_thresh, pre_max, normalized=False):
    """post_processing's score pipeline for a batch in one launch (cpd_rank_scores): cls [B, R, C] logits (scores when `normalized`),
    boxes [B, R, 7], labels [B, R] i64 -> (boxes, scores, labels i32) ranked by max-class sigmoid score descending (ties: lower index),
    rows below the threshold last with score -1, and n_ok [B] i32 = min(rows above the threshold, pre_max)."""
    b, r, c = cls.shape
    cls, boxes, labels = cls.contiguous().float(), boxes.contiguous().float(), labels.contiguous().long()
    ob, osc = torch.empty_like(boxes), torch.empty((b, r), dtype=torch.float32, device=cls.device)
    ol = torch.empty((b, r), dtype=torch.int32, device=cls.device)
    n_ok = torch.empty((b,), dtype=torch.int32, device=cls.device)
    check(lib().cpd_rank_scores(ptr(cls), c, ptr(boxes), ctypes.c_void_p(labels.data_ptr()), b, r, float(score_thresh), int(pre_max),
                                int(bool(normalized)), ptr(ob), ptr(osc), ptr(ol), ptr(n_ok), stream()), "cpd_rank_scores")
    return ob, osc, ol, n_ok


def select_boxes(boxes, scores, labels, keep, num_keep, post_max, label_offset=0, packed=False, extra_ints=0):
    """out[b][k] = in[b][keep[b][k]], k < min(num_keep[b], post_max). Returns padded
    (boxes [batch,post_max,7], scores, labels i64, counts [batch] i32).
    packed=True: the four outputs are views of ONE allocation, returned as a fifth value (a flat uint8 tensor: counts -- `batch` int32
    followed by `extra_ints` more words the caller may fill --, then boxes, scores, labels): one device-to-host copy moves a step's
    results, `unpack_boxes` cuts the host copy up again."""
    batch, cap = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    if packed:
        lay = _box_block_layout(batch, int(post_max), int(extra_ints))
        blk = torch.empty((lay["bytes"],), dtype=torch.uint8, device=dev)
        hdr, ob, os_, ol = _box_block_views(blk, lay)
        hdr.zero_()
        on = hdr[:batch]
    else:
        ob = torch.empty((batch, post_max, 7), dtype=torch.float32, device=dev)
        os_ = torch.empty((batch, post_max), dtype=torch.float32, device=dev)
        ol = torch.empty((batch, post_max), dtype=torch.int64, device=dev)
        on = torch.zeros((batch,), dtype=torch.int32, device=dev)
    check(lib().cpd_select_boxes(ptr(boxes), ptr(scores), ptr(labels), ptr(keep), ptr(num_keep), batch, cap, int(post_max),
                                 int(label_offset), ptr(ob), ptr(os_), ptr(ol), ptr(on), stream()), "cpd_select_boxes")
    if packed:
        blk._cpd_layout = lay
        return ob, os_, ol, on, blk
    return ob, os_, ol, on


def _box_block_layout(batch, post_max, extra_ints):
    a = lambda n: (n + 15) // 16 * 16
    n_hdr = batch + extra_ints
    o_box = a(4 * n_hdr)
    o_sc = a(o_box + 4 * batch * post_max * 7)
    o_lab = a(o_sc + 4 * batch * post_max)
    return dict(batch=batch, post_max=post_max, n_hdr=n_hdr, o_box=o_box, o_sc=o_sc, o_lab=o_lab, bytes=a(o_lab + 8 * batch * post_max))


def _box_block_views(blk, lay):
    b, pm = lay["batch"], lay["post_max"]
    hdr = blk[:4 * lay["n_hdr"]].view(torch.int32)
    ob = blk[lay["o_box"]:lay["o_box"] + 4 * b * pm * 7].view(torch.float32).view(b, pm, 7)
    os_ = blk[lay["o_sc"]:lay["o_sc"] + 4 * b * pm].view(torch.float32).view(b, pm)
    ol = blk[lay["o_lab"]:lay["o_lab"] + 8 * b * pm].view(torch.int64).view(b, pm)
    return hdr, ob, os_, ol


def unpack_boxes(host_blk, lay):
    """(header int32 words, boxes, scores, labels) of a host copy of select_boxes(packed=True)'s block"""
    return _box_block_views(host_blk, lay)


# ---------------------------------------------------------------------------------------- B3
def _pairwise(fn, name, a, b):
    _need_cuda(a, "boxes_a")
    a = a.contiguous().float()
    b = b.contiguous().float()
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(fn(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), stream()), name)
    return out


def boxes_overlap_bev(a, b):
    return _pairwise(lib().cpd_boxes_overlap_bev, "cpd_boxes_overlap_bev", a, b)


def boxes_iou_bev(a, b):
    return _pairwise(lib().cpd_boxes_iou_bev, "cpd_boxes_iou_bev", a, b)


def boxes_iou3d(a, b):
    return _pairwise(lib().cpd_boxes_iou3d, "cpd_boxes_iou3d", a, b)


def nms(boxes_sorted, thresh, normal=False, sync=True):
    """boxes sorted by descending score. Returns kept row ids (device i64 [num]) (sync=True) or
    (keep [n], num_keep [1]) device tensors."""
    _need_cuda(boxes_sorted, "boxes")
    boxes_sorted = boxes_sorted.contiguous().float()
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty(max(256, lib().cpd_nms_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    fn = lib().cpd_nms_normal if normal else lib().cpd_nms_rotated
    check(fn(ptr(boxes_sorted), n, float(thresh), ptr(keep), ptr(num), ptr(ws), ws.numel(), stream()), "cpd_nms")
    if not sync:
        return keep, num
    return keep[:int(num.item())]


def boxes_iou_bev_cpu(a, b):
    """Host tensors (iou3d_nms_utils.boxes_bev_iou_cpu contract)."""
    a = a.contiguous().float()
    b = b.contiguous().float()
    assert not a.is_cuda and not b.is_cuda
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32)
    check(lib().cpd_boxes_iou_bev_cpu(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out)), "cpd_boxes_iou_bev_cpu")
    return out


def gather_conv_tile(n_out, c_in, c_out, in_ld, dense=False, bf16x3=False, nbr=None, math=None, scaled=False, in_pairs=False):
    """Name of the kernel instantiation gather_conv will run for this problem (`nbr`: the table it would be given; `scaled`: an
    `in_absmax` block comes with the input -- the split-fp16 kernels then run as their pre-scaling `f16s` instantiations)."""
    if nbr is not None and getattr(nbr, "plan", None) is not None and not scaled and lib().cpd_gather_conv_planned_supported(
            int(nbr.shape[1]), int(n_out), int(c_in), int(c_out), int(in_ld), int(nbr.shape[0]),
            _gc_flags(dense, bf16x3, math) | (16 if in_pairs else 0) | 32):      # (the staged kernel writes pair rows)
        return "rowplan_conv_f16p_kernel<%d%s>" % (c_out, ",256" if nbr.plan[3] == 256 else "")
    name = _gather_conv_tile(n_out, c_in, c_out, in_ld, dense, bf16x3, nbr, math, in_pairs)
    return name.replace("_f16_kernel", "_f16s_kernel") if scaled else name


def _gather_conv_tile(n_out, c_in, c_out, in_ld, dense=False, bf16x3=False, nbr=None, math=None, in_pairs=False):
    image = getattr(nbr, "image", None)
    flags = _gc_flags(dense, bf16x3, math) | (16 if in_pairs else 0)
    if image is not None and n_out == image[0] * image[1] * image[2] and in_ld % 4 == 0 and lib().cpd_conv3x3_rows_supported(
            image[0], image[1], image[2], int(c_in), int(c_out), flags):
        bm, bn = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().cpd_conv3x3_rows_tile(image[0], image[1], image[2], int(c_in), int(c_out), flags, ctypes.byref(bm), ctypes.byref(bn)),
              "cpd_conv3x3_rows_tile")
        return "window_conv_%s_kernel<%s>" % ("f16" if flags & 4 else "bf16", "%d" % bn.value if bm.value == 128 else "%d,%d" % (bn.value, bm.value))
    wg, a, b, vec = (ctypes.c_int(0) for _ in range(4))
    check(lib().cpd_gather_conv_tile(int(n_out), int(c_in), int(c_out), int(in_ld), flags,
                                     ctypes.byref(wg), ctypes.byref(a), ctypes.byref(b), ctypes.byref(vec)),
          "cpd_gather_conv_tile")
    if wg.value == 13 and a.value > 128:
        return "rowwave_conv_f16pw_kernel<%d,%d>" % (b.value, a.value // 32)            # wide workgroups: <column tile, waves>
    if wg.value in (3, 13):
        return "rowwave_conv_%s_kernel<%d,%d>" % (("f16p" if flags & 16 else "f16") if wg.value == 13 else "bf16", b.value, a.value // 64)   # <column tile, row sub-tiles per wave>
    if wg.value in (2, 12):
        return "tile_conv_%s_kernel<%d,%d>" % ("f16" if wg.value == 12 else "bf16", a.value, b.value)
    if wg.value:
        return "tile_conv_kernel<%d,%d>" % (a.value, b.value)
    if in_pairs and c_in == 16:
        return "gather_conv_h16_kernel<%d,%d>" % (a.value, b.value)                   # 16-channel pair rows: the K = 16 MFMA form
    return "gather_conv_kernel<%d,%d,%s>" % (a.value, b.value, "true" if vec.value else "false")
