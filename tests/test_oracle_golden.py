"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py) and, when built, against the compiled reference iou3d_cpu.cpp."""
import numpy as np
import pytest


def test_mean_vfe_matches_reference(oracle, golden):
    g = golden("mean_vfe")
    out = oracle.mean_vfe(g["voxels"], g["num_points"])
    np.testing.assert_allclose(out, g["features"], rtol=1e-6, atol=1e-6)


def _bn(o, x, sd, prefix, eps, relu=True):
    return o.bn_relu(x, sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                     sd[prefix + ".running_var"], eps, relu)


def oracle_bev_backbone(o, sd, x, layer_nums=(2, 2), strides=(1, 2), ups=(1, 2)):
    """BaseBEVBackbone.forward (base_bev_backbone.py:85-122) composed from oracle primitives."""
    outs = []
    for lvl in range(len(layer_nums)):
        p = "sd.blocks.%d." % lvl
        x = o.conv2d(x, sd[p + "1.weight"], None, stride=strides[lvl], pad=1)  # ZeroPad2d(1)+conv pad 0
        x = _bn(o, x, sd, p + "2", 1e-3)
        for k in range(layer_nums[lvl]):
            x = o.conv2d(x, sd[p + "%d.weight" % (4 + 3 * k)], None, stride=1, pad=1)
            x = _bn(o, x, sd, p + "%d" % (5 + 3 * k), 1e-3)
        q = "sd.deblocks.%d." % lvl
        u = o.deconv2d(x, sd[q + "0.weight"], ups[lvl])
        outs.append(_bn(o, u, sd, q + "1", 1e-3))
    return np.concatenate(outs, axis=1)


def test_bev_backbone_matches_reference(oracle, golden):
    g = golden("bev_backbone")
    y = oracle_bev_backbone(oracle, g, g["bev_in"])
    assert y.shape == g["bev_out"].shape
    np.testing.assert_allclose(y, g["bev_out"], atol=1e-4, rtol=0)


def oracle_center_head(o, g, x):
    """shared_conv + SeparateHead (center_head.py:11-45,73-80,323-330); default BN eps 1e-5."""
    x = o.conv2d(x, g["shared.0.weight"], g["shared.0.bias"], 1, 1)
    x = o.bn_relu(x, g["shared.1.weight"], g["shared.1.bias"], g["shared.1.running_mean"],
                  g["shared.1.running_var"], 1e-5, True)
    outs = {}
    for name in ["center", "center_z", "dim", "rot", "hm"]:
        p = "sep.%s." % name
        h = o.conv2d(x, g[p + "0.0.weight"], g[p + "0.0.bias"], 1, 1)
        h = o.bn_relu(h, g[p + "0.1.weight"], g[p + "0.1.bias"], g[p + "0.1.running_mean"],
                      g[p + "0.1.running_var"], 1e-5, True)
        outs[name] = o.conv2d(h, g[p + "1.weight"], g[p + "1.bias"], 1, 1)
    return x, outs


def test_center_head_matches_reference(oracle, golden):
    g = golden("center_head")
    mid, outs = oracle_center_head(oracle, g, g["head_in"])
    np.testing.assert_allclose(mid, g["shared_out"], atol=1e-4, rtol=0)
    for k, v in outs.items():
        np.testing.assert_allclose(v, g["out." + k], atol=1e-4, rtol=0)


def test_topk_and_decode_match_reference(oracle, golden):
    g = golden("decode")
    K = int(g["K"])
    hm = g["hm"][0]
    sig = (1.0 / (1.0 + np.exp(-hm.astype(np.float64)))).astype(np.float32)
    H, W = hm.shape[1:]
    # stage 1 of _topk, per class (centernet_utils.py:139)
    s_all, i_all = [], []
    for c in range(hm.shape[0]):
        s, i = oracle.topk(sig[c], K)
        s_all.append(s); i_all.append(i)
    s2, i2 = oracle.topk(np.concatenate(s_all), K)
    np.testing.assert_allclose(s2, g["topk_scores"][0], atol=1e-6)
    np.testing.assert_array_equal(i2 // K, g["topk_classes"][0])
    np.testing.assert_array_equal(np.concatenate(i_all)[i2], g["topk_inds"][0])
    boxes, scores, labels = oracle.center_decode(
        hm, g["center"][0], g["center_z"][0], g["dim"][0], g["rot"][0], K, float(g["stride"]),
        g["vs"][:2], g["pcr"][:2], g["limit"], float(g["score_thresh"]))
    assert boxes.shape == g["boxes"].shape
    np.testing.assert_array_equal(labels, g["labels"])
    np.testing.assert_allclose(scores, g["scores"], atol=1e-6)
    np.testing.assert_allclose(boxes, g["boxes"], atol=1e-5, rtol=1e-5)


def test_iou_bev_matches_reference_golden(oracle, golden):
    g = golden("iou_bev")
    np.testing.assert_array_equal(oracle.boxes_iou_bev(g["a"], g["b"]), g["iou_ab"])
    got = oracle.boxes_iou_bev(g["adv"], g["adv"])
    # degenerate boxes (tiny / nan-producing) must agree including nan placement
    np.testing.assert_array_equal(np.isnan(got), np.isnan(g["iou_adv"]))
    np.testing.assert_array_equal(np.nan_to_num(got, nan=-1), np.nan_to_num(g["iou_adv"], nan=-1))


def test_iou_bev_matches_compiled_reference_live(oracle):
    from oracle import load_reference_iou
    ref = load_reference_iou()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(7)
    n = 256
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-8, 8, (n, 2)); b[:, 3:6] = rng.uniform(0.5, 5, (n, 3)); b[:, 6] = rng.uniform(-4, 4, n)
    np.testing.assert_array_equal(oracle.boxes_iou_bev(b, b), ref(b, b))


@pytest.mark.parametrize("tag", ["n64", "n500", "n500_t3", "n1000_t1"])
def test_nms_matches_reference(oracle, golden, tag):
    """class_agnostic_nms (model_nms_utils.py:115-134): topk -> nms_gpu -> [:POST]."""
    g = golden("nms")
    boxes, scores, thr = g[tag + ".boxes"], g[tag + ".scores"], float(g[tag + ".thr"])
    order = np.argsort(-scores, kind="stable")[:4096]
    keep = oracle.nms(boxes[order], thr)
    sel = order[keep][:500]
    np.testing.assert_array_equal(sel, g[tag + ".selected"])
    np.testing.assert_array_equal(scores[sel], g[tag + ".selected_scores"])


def test_nms_4096_matches_reference(oracle, golden):
    """NMS_PRE_MAXSIZE-sized set (64 mask words per row) through the reference's class_agnostic_nms."""
    g = golden("nms_n4096")
    boxes, scores, thr = g["boxes"], g["scores"], float(g["thr"])
    order = np.argsort(-scores, kind="stable")[:4096]
    sel = order[oracle.nms(boxes[order], thr)]
    np.testing.assert_array_equal(sel, g["selected"])
    np.testing.assert_array_equal(scores[sel], g["selected_scores"])


def test_iou3d_composition(oracle, golden):
    """boxes_iou3d_gpu (iou3d_nms_utils.py:67-100) recomposed in numpy from oracle overlaps."""
    g = golden("iou_bev")
    a, b = g["a"], g["b"]
    ov = oracle.boxes_overlap_bev(a, b)
    amax = (a[:, 2] + a[:, 5] / 2)[:, None]; amin = (a[:, 2] - a[:, 5] / 2)[:, None]
    bmax = (b[:, 2] + b[:, 5] / 2)[None]; bmin = (b[:, 2] - b[:, 5] / 2)[None]
    oh = np.clip(np.minimum(amax, bmax) - np.maximum(amin, bmin), 0, None)
    o3 = ov * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5])[:, None]; vb = (b[:, 3] * b[:, 4] * b[:, 5])[None]
    want = o3 / np.clip(va + vb - o3, 1e-6, None)
    np.testing.assert_allclose(oracle.boxes_iou3d(a, b), want, atol=1e-6)
