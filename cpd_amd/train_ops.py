"""Tensor-level wrappers of the training entry points of the C-ABI (include/cpd_hip.h, "Training step")."""
import ctypes

import torch

from ._lib import check, farr, iarr, lib, ptr, stream

_ws_cache = {}
ABSMAX_WORDS = 16 * 32       # CPD_ABSMAX_WORDS of include/cpd_hip.h: an absmax block (16 words, one per 128-byte line)


def absmax_block(t):
    """The absmax block of a tensor whose maximum the caller computes itself (what bn_backward leaves behind for its dx)."""
    blk = torch.zeros(ABSMAX_WORDS, dtype=torch.int32, device=t.device)
    blk[:1] = t.abs().max().reshape(1).contiguous().view(torch.int32)
    return blk


def _ws(nbytes, device):
    """Grow-only scratch buffer per (device, stream) -- the kernels on one stream run in order."""
    key = (device, stream().value)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def col_sum(x, out=None):
    n, c = x.shape
    if out is None:
        out = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().cpd_col_reduce_workspace_bytes(n, c), x.device)
    check(lib().cpd_col_sum(_p(x), _ld(x), n, c, ptr(out), ptr(ws), ws.numel(), stream()), "cpd_col_sum")
    return out


def bn_stats(x):
    """(sum[c], sumsq[c]) over the rows of x [n, c]."""
    n, c = x.shape
    s1 = torch.empty((c,), dtype=torch.float32, device=x.device)
    s2 = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _ws(lib().cpd_col_reduce_workspace_bytes(n, c), x.device)
    check(lib().cpd_bn_stats(_p(x), _ld(x), n, c, ptr(s1), ptr(s2), ptr(ws), ws.numel(), stream()), "cpd_bn_stats")
    return s1, s2


def bn_finalize(s1, s2, n, eps, momentum, gamma, beta, running_mean=None, running_var=None):
    """(mean, invstd, scale, shift) from the batch sums; updates the running statistics in place."""
    c = s1.numel()
    out = torch.empty((4, c), dtype=torch.float32, device=s1.device)
    check(lib().cpd_bn_finalize(ptr(s1), ptr(s2), int(n), c, float(eps), float(momentum), ptr(gamma), ptr(beta), _p(out[0]),
                                _p(out[1]), _p(out[2]), _p(out[3]), ptr(running_mean), ptr(running_var), stream()),
          "cpd_bn_finalize")
    return out[0], out[1], out[2], out[3]


def bn_stats_finalize(x, eps, momentum, gamma, beta, running_mean=None, running_var=None):
    """Batch statistics of x [n, c] -> (mean, invstd, scale, shift); running stats updated in place."""
    n, c = x.shape
    out = torch.empty((4, c), dtype=torch.float32, device=x.device)
    ws = _ws(lib().cpd_col_reduce_workspace_bytes(n, c), x.device)
    check(lib().cpd_bn_stats_finalize(_p(x), _ld(x), n, c, float(eps), float(momentum), ptr(gamma), ptr(beta), _p(out[0]),
                                      _p(out[1]), _p(out[2]), _p(out[3]), ptr(running_mean), ptr(running_var), ptr(ws),
                                      ws.numel(), stream()), "cpd_bn_stats_finalize")
    return out[0], out[1], out[2], out[3]


def bn_stats_finalize_sync(x, eps, momentum, gamma, beta, running_mean=None, running_var=None, group=None):
    """SyncBatchNorm forward bookkeeping (tools/train.py:117 convert_sync_batchnorm): this rank's (sum, sumsq, rows) all-reduced over the
    data-parallel group, then mean / invstd / scale / shift from the totals (cpd_bn_finalize_sync: the total row count is a device value).
    -> (mean, invstd, scale, shift, n_total [1] device float -- bn_backward(sync=...) needs it)."""
    import torch.distributed as dist
    n, c = x.shape
    if n > 0:
        s1, s2 = bn_stats(x)
        buf = torch.cat([s1, s2, torch.full((1,), float(n), dtype=torch.float32, device=x.device)])
    else:                                        # a rank without rows at this level still takes part in the collective
        buf = torch.zeros((2 * c + 1,), dtype=torch.float32, device=x.device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = torch.empty((4, c), dtype=torch.float32, device=x.device)
    n_total = buf[2 * c:]
    check(lib().cpd_bn_finalize_sync(_p(buf[:c]), _p(buf[c:2 * c]), _p(n_total), c, float(eps), float(momentum), ptr(gamma), ptr(beta),
                                     _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), ptr(running_mean), ptr(running_var), stream()),
          "cpd_bn_finalize_sync")
    return out[0], out[1], out[2], out[3], n_total


def pack_weight_adjoint(w_kio, flip_taps, out=None):
    kv, cin, cout = w_kio.shape
    n = lib().cpd_packed_weight_floats(kv, cout, cin)
    packed = out if out is not None else torch.empty((n,), dtype=torch.float32, device=w_kio.device)
    assert packed.numel() == n
    check(lib().cpd_pack_weight_adjoint(ptr(w_kio), kv, cin, cout, int(bool(flip_taps)), ptr(packed), stream()),
          "cpd_pack_weight_adjoint")
    return packed


class _PackJob(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("kv", ctypes.c_int32), ("c_in", ctypes.c_int32),
                ("c_out", ctypes.c_int32), ("adjoint", ctypes.c_int32), ("flip_taps", ctypes.c_int32)]


class PackBatch:
    """Every packed weight image of a model rebuilt in three launches (cpd_pack_batch_*): `jobs` = [(w_kio, packed, adjoint,
    flip_taps)] with w_kio [kv, c_in, c_out] views whose storage never moves (the trainer's flat parameter buffer) and
    `packed` buffers sized by packed_floats(). images: 1 = split-bf16, 2 = split-fp16, 3 = both (the fp32 image always)."""

    def __init__(self, jobs):
        self.n = len(jobs)
        self.keep = jobs
        arr = (_PackJob * self.n)()
        for i, (w, packed, adjoint, flip) in enumerate(jobs):
            assert w.is_contiguous() and w.dtype == torch.float32 and w.is_cuda and packed.is_contiguous()
            kv, cin, cout = w.shape
            need = lib().cpd_packed_weight_floats(kv, cout if adjoint else cin, cin if adjoint else cout)
            assert packed.numel() == need, (packed.numel(), need)
            arr[i] = _PackJob(w.data_ptr(), packed.data_ptr(), kv, cin, cout, int(bool(adjoint)), int(bool(flip)))
        self.table = torch.empty((lib().cpd_pack_batch_table_bytes(self.n),), dtype=torch.uint8, device=jobs[0][0].device)
        self.grid = (ctypes.c_int32 * 3)()
        check(lib().cpd_pack_batch_prepare(ctypes.cast(arr, ctypes.c_void_p), self.n, ptr(self.table), self.table.numel(),
                                           ctypes.cast(self.grid, ctypes.c_void_p)), "cpd_pack_batch_prepare")

    def run(self, images=3):
        check(lib().cpd_pack_batch_run(ptr(self.table), self.n, ctypes.cast(self.grid, ctypes.c_void_p), int(images), stream()),
              "cpd_pack_batch_run")


def packed_floats(kv, c_in, c_out):
    return lib().cpd_packed_weight_floats(kv, c_in, c_out)


def affine_rows(x, scale=None, shift=None, residual=None, relu=False, out=None):
    n, c = x.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib().cpd_affine_rows(_p(x), _ld(x), n, c, ptr(scale), ptr(shift), _p(residual),
                                _ld(residual) if residual is not None else 0, int(bool(relu)), _p(out), _ld(out), stream()),
          "cpd_affine_rows")
    return out


def bn_backward(dy, y, x, mean, invstd, gamma, want_dres=False, dgamma=None, dbeta=None, dx_absmax=None, sync=None):
    """BatchNorm(+ReLU when y is given) backward. Returns (dx, dgamma, dbeta, dres|None);
    dgamma / dbeta may be caller-provided (views of a flat gradient buffer). `dx_absmax`: a ZEROED int32 device
    tensor of ABSMAX_WORDS words (an absmax block) that receives the bits of max |dx| (for the split-fp16 gradient convolutions).
    `sync` = (n_total, group) from bn_stats_finalize_sync: SyncBatchNorm backward -- the input gradient uses the ALL-REDUCED
    (sum dy, sum dy * xhat) and the total row count; dgamma / dbeta stay this rank's sums (the gradient all-reduce averages them)."""
    n, c = x.shape
    dev = x.device
    if dbeta is None:
        dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
    if dgamma is None:
        dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
    ws = _ws(lib().cpd_col_reduce_workspace_bytes(n, c), dev)
    check(lib().cpd_bn_bwd_reduce(_p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(x), _ld(x), ptr(mean), ptr(invstd),
                                  n, c, ptr(dbeta), ptr(dgamma), ptr(ws), ws.numel(), stream()), "cpd_bn_bwd_reduce")
    dx = torch.empty((n, c), dtype=torch.float32, device=dev)
    dres = torch.empty((n, c), dtype=torch.float32, device=dev) if want_dres else None
    if sync is not None:
        import torch.distributed as dist
        n_total, group = sync
        tot = torch.cat([dbeta, dgamma])
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        check(lib().cpd_bn_bwd_apply_sync(_p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(x), _ld(x), n, c, ptr(mean),
                                          ptr(invstd), ptr(gamma), _p(tot[:c]), _p(tot[c:]), _p(n_total), _p(dx), _ld(dx), _p(dres),
                                          _ld(dres) if dres is not None else 0, ptr(dx_absmax), stream()), "cpd_bn_bwd_apply_sync")
        return dx, dgamma, dbeta, dres
    check(lib().cpd_bn_bwd_apply(_p(dy), _ld(dy), _p(y), _ld(y) if y is not None else 0, _p(x), _ld(x), n, c, ptr(mean),
                                 ptr(invstd), ptr(gamma), ptr(dbeta), ptr(dgamma), _p(dx), _ld(dx), _p(dres),
                                 _ld(dres) if dres is not None else 0, ptr(dx_absmax), stream()), "cpd_bn_bwd_apply")
    return dx, dgamma, dbeta, dres


def relu_backward(dy, y):
    n, c = y.shape
    dx = torch.empty((n, c), dtype=torch.float32, device=y.device)
    check(lib().cpd_relu_bwd(_p(dy), _ld(dy), _p(y), _ld(y), n, c, _p(dx), _ld(dx), stream()), "cpd_relu_bwd")
    return dx


def conv_wgrad(inp, c_in, dy, c_out, nbr, kv, n_out, dw=None, accumulate=False, bf16x3=False, math=None, in_absmax=None,
               dy_absmax=None):
    """dw[kv, c_in, c_out] (+)= sum_j inp[nbr[t][j]]^T dy[j]; bf16x3: split-bf16 arithmetic (fp32-equivalent).
    math = "f16x2": split-fp16 (half the matrix work); an operand that is a gradient then needs its `*_absmax` word
    (an absmax block, see bn_backward / absmax_block) so that the kernel can bring it into fp16's range."""
    if math is not None:
        bf16x3 = math == "bf16x3"
    if dw is None:
        dw = torch.empty((kv, c_in, c_out), dtype=torch.float32, device=inp.device)
        accumulate = False
    ws = _ws(lib().cpd_conv_wgrad_workspace_bytes(n_out, c_in, c_out, kv), inp.device)
    check(lib().cpd_conv_wgrad_scaled(_p(inp), _ld(inp), c_in, _p(dy), _ld(dy), c_out, ptr(nbr), kv, n_out, ptr(dw),
                               int(bool(accumulate)) | (2 if bf16x3 else 0) | (4 if math == "f16x2" else 0), ptr(in_absmax),
                                      ptr(dy_absmax), ptr(ws), ws.numel(), stream()), "cpd_conv_wgrad")
    return dw


def rulebook_conv_transpose(in_indices, batch, in_shape, ksize, stride, pad, out_index):
    n_in = in_indices.shape[0]
    kv = int(ksize[0] * ksize[1] * ksize[2])
    nbr_t = torch.empty((kv, n_in), dtype=torch.int32, device=in_indices.device)
    check(lib().cpd_rulebook_conv_transpose(ptr(in_indices.contiguous()), n_in, batch, iarr(in_shape), iarr(ksize), iarr(stride),
                                            iarr(pad), ptr(out_index.buf), ptr(nbr_t), stream()), "cpd_rulebook_conv_transpose")
    return nbr_t


def rulebook_conv2d_transpose(batch, h, w, kh, kw, stride, pad, device):
    nbr_t = torch.empty((kh * kw, batch * h * w), dtype=torch.int32, device=device)
    check(lib().cpd_rulebook_conv2d_transpose(batch, h, w, kh, kw, stride, pad, ptr(nbr_t), stream()),
          "cpd_rulebook_conv2d_transpose")
    return nbr_t


def center_targets(gt_boxes, feature_map_size, point_cloud_range, voxel_size, num_classes, feature_map_stride=8, num_max_objs=500,
                   gaussian_overlap=0.1, min_radius=2):
    """CenterHead.assign_targets on the device (cpd_center_targets; same arguments and returns as center_loss.assign_targets: heat
    [B, nc, H, W], target_boxes [B, K, 8], inds [B, K] i64, masks [B, K] i64) -- three launches, no host read-back."""
    if gt_boxes.shape[-1] != 8:
        # cpd_center_targets reads rows of exactly 8 floats (class in column 7). Ground truth with extra columns (velocity ...: the
        # reference keeps them, center_head.py:149-155) or fewer goes through the torch restatement, which honours shape[-1]
        from . import center_loss as _cl
        return _cl.assign_targets(gt_boxes, feature_map_size, point_cloud_range, voxel_size, num_classes, feature_map_stride, num_max_objs,
                                  gaussian_overlap, min_radius)
    gt = gt_boxes.contiguous().float()
    B, M, _ = gt.shape
    H, W = int(feature_map_size[0]), int(feature_map_size[1])
    K, dev = int(num_max_objs), gt.device
    heat = torch.empty((B, num_classes, H, W), dtype=torch.float32, device=dev)
    target = torch.empty((B, K, 8), dtype=torch.float32, device=dev)
    inds = torch.empty((B, K), dtype=torch.int64, device=dev)
    masks = torch.empty((B, K), dtype=torch.int64, device=dev)
    ws = _ws(lib().cpd_center_targets_workspace_bytes(B, K), dev)
    check(lib().cpd_center_targets(_p(gt) if M > 0 else None, B, M, int(num_classes), H, W, K, farr(point_cloud_range[:2]), farr(voxel_size[:2]),
                                   int(feature_map_stride), float(gaussian_overlap), int(min_radius), _p(heat), _p(target), _p(inds), _p(masks),
                                   ptr(ws), ws.numel(), stream()), "cpd_center_targets")
    return heat, target, inds, masks


def center_loss(rows, batch, hw, num_classes, hm_col, heat, target, inds, masks, code_weights, loc_weight=2.0, cls_weight=1.0):
    """Fused CenterHead loss + gradient (cpd_center_loss) on contiguous head rows [batch*hw, ld].
    Returns (losses[3] = total, hm, loc on the device, d_rows [batch*hw, ld])."""
    assert rows.is_contiguous() and rows.dtype == torch.float32
    ld = rows.shape[1]
    heat = heat.contiguous().float(); target = target.contiguous().float()
    inds = inds.contiguous().long(); masks = masks.contiguous().long()
    k = inds.shape[1]
    d_rows = torch.empty_like(rows)
    losses = torch.empty(3, dtype=torch.float32, device=rows.device)
    ws = _ws(lib().cpd_center_loss_workspace_bytes(batch, hw, ld), rows.device)
    check(lib().cpd_center_loss(_p(rows), ld, batch, hw, num_classes, hm_col, _p(heat), _p(target), _p(inds), _p(masks), k,
                                farr(code_weights if code_weights is not None else [1.0] * 8), float(loc_weight), float(cls_weight),
                                _p(d_rows), _p(losses), ptr(ws), ws.numel(), stream()), "cpd_center_loss")
    return losses, d_rows


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, grad_scale_dev=None):
    """grad_scale_dev: optional 1-element float32 device tensor multiplied into grad_scale inside the kernel."""
    check(lib().cpd_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr), float(beta1),
                              float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), ptr(grad_scale_dev),
                              stream()),
          "cpd_adam_step")
