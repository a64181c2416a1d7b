"""CenterHead decode and B3 (iou3d_nms) parity on the GPU against the reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from cpd_amd import iou3d_nms_utils, ops
from cpd_amd.synthetic import random_boxes

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_decode_matches_reference_golden(hip, golden):
    g = golden("decode")
    K = int(g["K"])
    hm, center, cz, dim, rot = (dev(g[k][0]) for k in ("hm", "center", "center_z", "dim", "rot"))
    nc, h, w = hm.shape
    boxes, scores, labels, n = ops.center_decode(hm, center, cz, dim, rot, 1, h * w, nc, h, w, K, float(g["stride"]),
                                                 g["vs"][:2], g["pcr"][:2], g["limit"], float(g["score_thresh"]))
    assert n == g["boxes"].shape[0]
    np.testing.assert_array_equal(labels.cpu().numpy(), g["labels"])
    np.testing.assert_allclose(scores.cpu().numpy(), g["scores"], atol=1e-6)
    np.testing.assert_allclose(boxes.cpu().numpy(), g["boxes"], atol=1e-5, rtol=1e-5)


def test_decode_channels_last_and_full_size(oracle, hip):
    """188x188 maps, K=500, channels-last rows exactly as the engine lays the head output out."""
    rng = np.random.default_rng(12)
    h = w = 188
    K = 500
    rows = rng.normal(size=(h * w, 16)).astype(np.float32)
    rows[:, 8:11] = rows[:, 8:11] * 2.0 - 1.0
    sl = dict(center=0, center_z=2, dim=3, rot=6, hm=8)
    t = dev(rows)
    pcr, vs = [-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15]
    boxes, scores, labels, n = ops.center_decode(t[:, 8:], t[:, 0:], t[:, 2:], t[:, 3:], t[:, 6:], 16, 1, 3, h, w, K, 8.0,
                                                 vs[:2], pcr[:2], pcr, 0.1)
    planes = rows.T.reshape(16, h, w)
    b0, s0, l0 = oracle.center_decode(planes[8:11], planes[0:2], planes[2:3], planes[3:6], planes[6:8], K, 8.0, vs[:2],
                                      pcr[:2], pcr, 0.1)
    assert n == b0.shape[0] and n > 100
    np.testing.assert_array_equal(labels.cpu().numpy(), l0)
    np.testing.assert_allclose(scores.cpu().numpy(), s0, atol=1e-6)
    np.testing.assert_allclose(boxes.cpu().numpy(), b0, atol=1e-4, rtol=1e-5)
    assert (np.diff(scores.cpu().numpy()) <= 0).all()          # sortedness property


def finite_close(got, want, atol):
    m = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), m)
    np.testing.assert_allclose(got[m], want[m], atol=atol, rtol=0)


def test_iou_matrices_match_reference_golden(oracle, hip, golden):
    g = golden("iou_bev")
    a, b = dev(g["a"]), dev(g["b"])
    np.testing.assert_allclose(ops.boxes_iou_bev(a, b).cpu().numpy(), g["iou_ab"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(ops.boxes_overlap_bev(a, b).cpu().numpy(), oracle.boxes_overlap_bev(g["a"], g["b"]), atol=1e-4)
    np.testing.assert_allclose(ops.boxes_iou3d(a, b).cpu().numpy(), oracle.boxes_iou3d(g["a"], g["b"]), atol=1e-4)
    np.testing.assert_allclose(iou3d_nms_utils.boxes_iou3d_gpu(a, b).cpu().numpy(), oracle.boxes_iou3d(g["a"], g["b"]), atol=1e-4)
    adv = dev(g["adv"])
    finite_close(ops.boxes_iou_bev(adv, adv).cpu().numpy(), g["iou_adv"], 1e-4)
    # the CPU entry point of the extension (boxes_bev_iou_cpu)
    np.testing.assert_allclose(iou3d_nms_utils.boxes_bev_iou_cpu(g["a"], g["b"]), g["iou_ab"], atol=1e-6)


@pytest.mark.parametrize("tag", ["n64", "n500", "n500_t3", "n1000_t1"])
def test_nms_matches_reference_golden(hip, golden, tag):
    """class_agnostic_nms semantics (model_nms_utils.py:115-134) through the iou3d_nms_utils mirror."""
    g = golden("nms")
    boxes, scores, thr = dev(g[tag + ".boxes"]), dev(g[tag + ".scores"]), float(g[tag + ".thr"])
    s_top, idx = torch.topk(scores, k=min(4096, scores.shape[0]))
    keep, _ = iou3d_nms_utils.nms_gpu(boxes[idx][:, 0:7], s_top, thr)
    sel = idx[keep[:500]]
    np.testing.assert_array_equal(sel.cpu().numpy(), g[tag + ".selected"])


def test_nms_4096_matches_reference_golden(hip, golden):
    """NMS_PRE_MAXSIZE boxes: 64 mask words per row, the multi-block path of the device scan, against the
    reference's class_agnostic_nms (tests/golden/make_golden.py nms_large)."""
    g = golden("nms_n4096")
    boxes, scores, thr = dev(g["boxes"]), dev(g["scores"]), float(g["thr"])
    s_top, idx = torch.topk(scores, k=4096)
    keep, _ = iou3d_nms_utils.nms_gpu(boxes[idx][:, 0:7], s_top, thr)
    np.testing.assert_array_equal(idx[keep].cpu().numpy(), g["selected"])


def _clean_boxes(oracle, n, thr):
    """Seeded random set with no pairwise IoU within 2e-4 of the threshold (fixtures stay away from
    borderline IoUs); the seed is advanced until the set is clean, so no case is ever skipped."""
    for seed in range(n, n + 64 * 7919, 7919):
        b, s = random_boxes(seed, n, span=max(8.0, n ** 0.5 * 1.5))
        bs = b[np.argsort(-s, kind="stable")]
        iou = oracle.boxes_iou_bev(bs, bs)
        if not np.any(np.abs(iou[np.triu_indices(n, 1)] - thr) < 2e-4):
            return bs
    raise AssertionError("no borderline-free set for n=%d" % n)


@pytest.mark.parametrize("n,thr", [(1, 0.5), (63, 0.3), (64, 0.3), (65, 0.3), (500, 0.8), (4096, 0.7)])
def test_nms_rotated_and_normal_vs_oracle(oracle, hip, n, thr):
    bs = _clean_boxes(oracle, n, thr)
    np.testing.assert_array_equal(ops.nms(dev(bs), thr).cpu().numpy(), oracle.nms(bs, thr))
    np.testing.assert_array_equal(ops.nms(dev(bs), thr, normal=True).cpu().numpy(), oracle.nms_normal(bs, thr))
    # idempotence: NMS of the survivors keeps everything
    kept = bs[oracle.nms(bs, thr)]
    assert ops.nms(dev(kept), thr).shape[0] == kept.shape[0]
    # extension-shaped entry point with the CPU `keep` contract (iou3d_nms.cpp:90-137)
    from cpd_amd import iou3d_nms_cuda
    keep = torch.LongTensor(n)
    num = iou3d_nms_cuda.nms_gpu(dev(bs), keep, thr)
    np.testing.assert_array_equal(keep[:num].numpy(), oracle.nms(bs, thr))


def test_empty_inputs(hip):
    e = torch.zeros((0, 7), device="cuda")
    assert ops.nms(e, 0.5).shape[0] == 0
    assert ops.boxes_iou_bev(e, torch.zeros((3, 7), device="cuda")).shape == (0, 3)


def test_nms_first_survivors_equal_the_full_scan(oracle, hip, golden):
    """cpd_nms_batch_first: the first `max_keep` survivors from a mask over each sample's first `row_limit` boxes, scan stopped at the
    max_keep-th survivor (class_agnostic_nms keeps selected[:NMS_POST_MAXSIZE], model_nms_utils.py:115-134). (a) the reference's own
    4096-box fixture: the first 200 / 500 selected boxes; (b) batches of random sets of different sizes and crowding against
    cpd_nms_batch, for several (max_keep, row_limit), rotated and axis-aligned; (c) `incomplete`: a crowded sample (30 distinct boxes,
    each repeated 40 times: 30 survivors among 1200) is flagged when the mask does not cover all of it and fewer than max_keep
    survived -- and not when the mask covers it, or when max_keep were found."""
    g = golden("nms_n4096")
    boxes, scores, thr = dev(g["boxes"]), dev(g["scores"]), float(g["thr"])
    s_top, idx = torch.topk(scores, k=4096)
    sorted_boxes = boxes[idx][:, 0:7].contiguous()[None]
    cnt = torch.tensor([4096], dtype=torch.int32, device="cuda")
    want = g["selected"]
    for max_keep, rows in ((200, 512), (500, 1024), (100, 4096), (64, 64)):
        keep, num, inc = ops.nms_batch_first(sorted_boxes, cnt, thr, max_keep, rows)
        k = min(int(num[0]), max_keep)
        if int(inc[0]):
            assert k < max_keep                                       # flagged only when the prefix came up short
            continue
        assert k == min(max_keep, len(want))
        np.testing.assert_array_equal(idx[keep[0, :k]].cpu().numpy(), want[:k])

    rng = np.random.default_rng(5)
    sizes = [1, 63, 64, 65, 700, 2048]
    cap = max(sizes)
    bb = np.zeros((len(sizes), cap, 7), np.float32)
    for i, n in enumerate(sizes):
        b, s = random_boxes(100 + i, n, span=max(6.0, n ** 0.5 * (0.6 if i % 2 else 1.5)))      # every other sample crowded
        bb[i, :n] = b[np.argsort(-s, kind="stable")]
    dbb = dev(bb)
    counts = torch.tensor(sizes, dtype=torch.int32, device="cuda")
    for normal in (False, True):
        full_keep, full_num = ops.nms_batch(dbb, counts, 0.3, normal=normal)
        for max_keep, rows in ((10, 64), (50, 256), (200, 512), (300, 2048), (5, 4096)):
            keep, num, inc = ops.nms_batch_first(dbb, counts, 0.3, max_keep, rows, normal=normal)
            for i, n in enumerate(sizes):
                fk = full_keep[i, :int(full_num[i])].cpu().numpy()
                k = min(int(num[i]), max_keep)
                np.testing.assert_array_equal(keep[i, :k].cpu().numpy(), fk[:k])            # always a prefix of the full answer
                if int(inc[i]):
                    assert n > rows and k < max_keep
                else:
                    assert k == min(max_keep, len(fk)), (normal, max_keep, rows, n, k, len(fk))

    base, _ = random_boxes(9, 30, span=60.0)
    crowd = np.repeat(base, 40, axis=0)[None].astype(np.float32)                            # 1200 boxes, 30 survivors
    cc = torch.tensor([1200], dtype=torch.int32, device="cuda")
    n_full = int(ops.nms_batch(dev(crowd), cc, 0.5)[1][0])
    assert n_full <= 30
    _, num, inc = ops.nms_batch_first(dev(crowd), cc, 0.5, 100, 512)
    assert int(inc[0]) == 1 and int(num[0]) < 100
    _, num, inc = ops.nms_batch_first(dev(crowd), cc, 0.5, 100, 1200)
    assert int(inc[0]) == 0 and int(num[0]) == n_full
    _, num, inc = ops.nms_batch_first(dev(crowd), cc, 0.5, 10, 512)
    assert int(inc[0]) == 0 and int(num[0]) >= 10


def test_nms_device_fallback_completes_the_flagged_samples(hip):
    """cpd_nms_batch_where: the full scan for exactly the samples cpd_nms_batch_first flagged, queued behind it without a read-back -- the
    pair returns cpd_nms_batch's first max_keep survivors for every sample; unflagged samples keep their early-exit answer."""
    rng = np.random.default_rng(11)
    cap, max_keep, rows = 1500, 100, 256
    bb = np.zeros((3, cap, 7), np.float32)
    b0, s0 = random_boxes(1, cap, span=60.0)                     # sparse: 100 survivors within the first 256
    bb[0] = b0[np.argsort(-s0, kind="stable")]
    base, _ = random_boxes(2, 30, span=60.0)
    bb[1, :1200] = np.repeat(base, 40, axis=0)                    # 30 survivors among 1200: flagged
    b2, s2 = random_boxes(3, 200, span=40.0)
    bb[2, :200] = b2[np.argsort(-s2, kind="stable")]              # fewer boxes than the mask covers: complete by construction
    counts = torch.tensor([cap, 1200, 200], dtype=torch.int32, device="cuda")
    dbb = dev(bb)
    full_keep, full_num = ops.nms_batch(dbb, counts, 0.5)
    keep, num, inc = ops.nms_batch_first(dbb, counts, 0.5, max_keep, rows)
    assert inc.tolist() == [0, 1, 0]
    early0 = (keep[0].clone(), int(num[0]))
    ops.nms_batch_where(dbb, counts, inc, 0.5, keep, num)
    assert torch.equal(keep[0], early0[0]) and int(num[0]) == early0[1]           # untouched
    for i in range(3):
        k = min(int(full_num[i]), max_keep)
        assert min(int(num[i]), max_keep) == k
        np.testing.assert_array_equal(keep[i, :k].cpu().numpy(), full_keep[i, :k].cpu().numpy())
    assert int(num[1]) == int(full_num[1])                                        # the flagged sample got the full answer


def test_select_boxes_packed_block_round_trips(hip):
    """ops.select_boxes(packed=True): counts (+ extra header words), boxes, scores and labels as views of ONE allocation -- the same values
    as the four separate tensors, and a host copy of the block cut up by ops.unpack_boxes gives them back (one D2H copy per step)."""
    g = torch.Generator().manual_seed(3)
    B, cap, post = 3, 300, 128
    boxes = torch.randn(B, cap, 7, generator=g).cuda()
    scores = torch.rand(B, cap, generator=g).cuda()
    labels = torch.randint(0, 3, (B, cap), generator=g).int().cuda()
    keep = torch.stack([torch.randperm(cap, generator=g) for _ in range(B)]).cuda()
    num_keep = torch.tensor([5, 200, 0], dtype=torch.int32, device="cuda")
    ob, os_, ol, on = ops.select_boxes(boxes, scores, labels, keep, num_keep, post, label_offset=1)
    pb, ps, pl, pn, blk = ops.select_boxes(boxes, scores, labels, keep, num_keep, post, label_offset=1, packed=True, extra_ints=2)
    assert pn.tolist() == on.tolist() == [5, 128, 0]
    lay = blk._cpd_layout
    hdr, hb, hs, hl = ops.unpack_boxes(blk.cpu(), lay)
    assert hdr.tolist() == [5, 128, 0, 0, 0]                      # the extra words start at zero
    for b, n in enumerate(on.tolist()):
        for dev_t, packed_t, host_t in ((ob, pb, hb), (os_, ps, hs), (ol, pl, hl)):
            assert torch.equal(dev_t[b, :n], packed_t[b, :n]) and torch.equal(dev_t[b, :n].cpu(), host_t[b, :n])
    assert hl.dtype == torch.int64 and hb.shape == (B, post, 7)


def test_multi_classes_nms_and_the_post_processing_branches_vs_oracle(oracle, hip):
    """model_nms_utils.multi_classes_nms (model_nms_utils.py:137-170) and Detector3DTemplate.post_processing's MULTI_CLASSES_NMS / WBF /
    OUTPUT_RAW_SCORE branches (detector3d_template.py:246-331) through cpd_amd.models, against a plain restatement on oracle.nms: per
    class column -- threshold, descending order, rotated NMS, first NMS_POST_MAXSIZE -- concatenated."""
    from cpd_amd import models
    n, nc, thr = 600, 3, 0.3
    bs = _clean_boxes(oracle, n, thr)
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 2, (n, nc)).astype(np.float32)
    prob = (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32)
    nms_cfg = dict(MULTI_CLASSES_NMS=True, NMS_TYPE="nms_gpu", NMS_THRESH=thr, NMS_PRE_MAXSIZE=256, NMS_POST_MAXSIZE=40)
    score_thresh = 0.2

    def want_multi(p):
        out_s, out_l, out_b = [], [], []
        for k in range(nc):
            ok = np.nonzero(p[:, k] >= score_thresh)[0]
            order = ok[np.argsort(-p[ok, k], kind="stable")][:nms_cfg["NMS_PRE_MAXSIZE"]]
            keep = oracle.nms(np.ascontiguousarray(bs[order]), thr)
            sel = order[keep][:nms_cfg["NMS_POST_MAXSIZE"]]
            out_s.append(p[sel, k]); out_l.append(np.full(len(sel), k)); out_b.append(bs[sel])
        return np.concatenate(out_s), np.concatenate(out_l), np.concatenate(out_b)

    s_, l_, b_ = models.multi_classes_nms(dev(prob), dev(bs), nms_cfg, score_thresh)
    ws, wl, wb = want_multi(prob)
    assert len(ws) > 30
    np.testing.assert_array_equal(l_.cpu().numpy(), wl)
    np.testing.assert_array_equal(b_.cpu().numpy(), wb)
    np.testing.assert_array_equal(s_.cpu().numpy(), ws)
    # the detector's post_processing on logits: sigmoid on the device (<= 1 ulp from numpy's: compare the selection through the boxes)
    pp = models.AttrDict(SCORE_THRESH=score_thresh, OUTPUT_RAW_SCORE=False, NMS_CONFIG=nms_cfg)
    fb, fs, fl = models.post_process_frame(pp, nc, dev(bs), dev(logits), False)
    np.testing.assert_array_equal(fb.cpu().numpy(), wb)
    np.testing.assert_array_equal(fl.cpu().numpy(), wl + 1)                  # classes 1 .. num_class
    np.testing.assert_allclose(fs.cpu().numpy(), ws, atol=1e-6)
    # multi-head list form with its label mapping (two heads over disjoint box ranges)
    heads = [dev(logits[:350, :2]), dev(logits[350:, 2:])]
    mapping = [torch.tensor([1, 2]).cuda(), torch.tensor([3]).cuda()]
    hb, hs, hl = models.post_process_frame(pp, nc, dev(bs), heads, False, None, mapping)
    assert set(hl.cpu().tolist()) <= {1, 2, 3} and len(hb) > 10
    first = (hl != 3).cpu().numpy()
    d = np.abs(hb.cpu().numpy()[:, None] - bs[None]).max(-1).argmin(1)       # which input box each output is
    assert (d[first] < 350).all() and (d[~first] >= 350).all()
    # WBF: the score mask only; class-agnostic with OUTPUT_RAW_SCORE: raw logits of the selected rows
    pw = models.AttrDict(SCORE_THRESH=0.6, OUTPUT_RAW_SCORE=False, WBF=True, NMS_CONFIG=dict(nms_cfg, MULTI_CLASSES_NMS=False))
    wb_, ws_, wl_ = models.post_process_frame(pw, nc, dev(bs), dev(prob), True)
    m = prob.max(-1) > 0.6
    np.testing.assert_array_equal(wb_.cpu().numpy(), bs[m])
    np.testing.assert_array_equal(wl_.cpu().numpy(), prob.argmax(-1)[m] + 1)
    pr = models.AttrDict(SCORE_THRESH=score_thresh, OUTPUT_RAW_SCORE=True, NMS_CONFIG=dict(nms_cfg, MULTI_CLASSES_NMS=False))
    rb, rs, rl = models.post_process_frame(pr, nc, dev(bs), dev(logits), False)
    sel = np.abs(rb.cpu().numpy()[:, None] - bs[None]).max(-1).argmin(1)
    np.testing.assert_array_equal(rs.cpu().numpy(), logits.max(-1)[sel])
