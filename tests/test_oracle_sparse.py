"""[SPCONV] parity-by-definition: the oracle's SubM / regular sparse conv and dense() must equal
torch's dense conv3d on the zero-filled grid evaluated at the active set.
spconv itself is not available, so this boundary is 'parity unpinned' by reference execution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def random_sites(rng, batch, shape, n):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=min(n, cells), replace=False)
    rng.shuffle(lin)
    b, r = np.divmod(lin, shape[0] * shape[1] * shape[2])
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def dense_from(feat, idx, batch, shape):
    c = feat.shape[1]
    d = np.zeros((batch, c) + tuple(shape), np.float32)
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
    return d


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (5, 8)])
def test_subm_equals_dense_conv3d(oracle, cin, cout):
    rng = np.random.default_rng(1)
    batch, shape = 2, [7, 12, 11]
    idx = random_sites(rng, batch, shape, 300)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=(cout, 3, 3, 3, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    nbr = oracle.subm_rulebook(idx, batch, shape, [3, 3, 3])
    out = oracle.sparse_conv(feat, w, bias, nbr)
    dense = torch.from_numpy(dense_from(feat, idx, batch, shape))
    wt = torch.from_numpy(w).permute(0, 4, 1, 2, 3).contiguous()
    ref = F.conv3d(dense, wt, torch.from_numpy(bias), padding=1).numpy()
    want = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    np.testing.assert_allclose(out, want, atol=1e-4, rtol=0)
    # centre tap is the identity pairing
    np.testing.assert_array_equal(nbr[13], np.arange(idx.shape[0]))


@pytest.mark.parametrize("ksize,stride,pad", [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
                                              ([3, 1, 1], [2, 1, 1], [0, 0, 0])])
def test_regular_conv_equals_dense_conv3d(oracle, ksize, stride, pad):
    rng = np.random.default_rng(2)
    batch, shape, cin, cout = 2, [9, 14, 13], 6, 10
    idx = random_sites(rng, batch, shape, 250)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    w = (rng.normal(size=[cout] + ksize + [cin]) / np.sqrt(np.prod(ksize) * cin)).astype(np.float32)
    out_idx = oracle.conv_outset(idx, batch, shape, ksize, stride, pad)
    oshape = oracle.conv_out_shape(shape, ksize, stride, pad)
    nbr = oracle.conv_rulebook(idx, out_idx, batch, shape, ksize, stride, pad)
    out = oracle.sparse_conv(feat, w, None, nbr)
    dense = torch.from_numpy(dense_from(feat, idx, batch, shape))
    wt = torch.from_numpy(w).permute(0, 4, 1, 2, 3).contiguous()
    ref = F.conv3d(dense, wt, None, stride=stride, padding=pad).numpy()
    assert list(ref.shape[2:]) == oshape
    # active outputs = sites whose receptive field holds an active input (NOT 'nonzero outputs')
    occ = torch.from_numpy(dense_from(np.ones((idx.shape[0], 1), np.float32), idx, batch, shape))
    cnt = F.conv3d(occ, torch.ones(1, 1, *ksize), None, stride=stride, padding=pad).numpy()[:, 0]
    want_idx = np.argwhere(cnt > 0).astype(np.int32)          # ascending (b,z,y,x): canonical order
    np.testing.assert_array_equal(out_idx, want_idx)
    want = ref[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]]
    np.testing.assert_allclose(out, want, atol=1e-4, rtol=0)


def test_densify_matches_definition(oracle):
    rng = np.random.default_rng(3)
    batch, shape, c = 2, [2, 9, 8], 6
    idx = random_sites(rng, batch, shape, 60)
    feat = rng.normal(size=(idx.shape[0], c)).astype(np.float32)
    out = oracle.densify(feat, idx, batch, shape)
    want = dense_from(feat, idx, batch, shape).reshape(batch, c * shape[0], shape[1], shape[2])
    np.testing.assert_array_equal(out, want)     # channel = c*D + d (height_compression.py:136-138)


def test_affine_rows(oracle):
    rng = np.random.default_rng(4)
    x = rng.normal(size=(50, 8)).astype(np.float32)
    s = rng.normal(size=8).astype(np.float32); t = rng.normal(size=8).astype(np.float32)
    r = rng.normal(size=(50, 8)).astype(np.float32)
    np.testing.assert_allclose(oracle.affine_rows(x, s, t, r, True), np.maximum(x * s + t + r, 0), atol=1e-6)
