#!/usr/bin/env python3
"""CPU study (oracle rulebooks, numpy): how many DISTINCT input rows does a tile of consecutive output rows of a sub-manifold level
touch, against its (row, tap) pairs -- the reuse an LDS-staged gather could exploit -- in canonical row order and
with the rows of every chunk sorted by tap pattern (the engine's row order).   python tools/unique_probe.py [seed]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import Oracle
from cpd_amd.synthetic import waymo_cloud, WAYMO

o = Oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pts = waymo_cloud(seed)
vox = o.voxelize(pts, WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 1000000)
coords = vox[1]
g = o.grid_size(WAYMO["voxel_size"], WAYMO["point_cloud_range"])
shape = [g[0] + 1, g[1], g[2]]
idx = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords[:, :3] if coords.shape[1] == 3 else coords[:, -3:]], 1).astype(np.int32)
def canon(i, shape):
    key = ((i[:, 0].astype(np.int64) * shape[0] + i[:, 1]) * shape[1] + i[:, 2]) * shape[2] + i[:, 3]
    return i[np.argsort(key, kind="stable")]
idx = canon(idx, shape)
DOWN = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1])]
levels = [("L1", idx, shape)]
for k, s, p in DOWN:
    oi = o.conv_outset(idx, 1, shape, k, s, p)
    shape = o.conv_out_shape(shape, k, s, p)
    idx = canon(oi, shape)
    levels.append(("L%d" % (len(levels) + 1), idx, shape))

def stats(name, nbr, tile, order=None):
    kv, n = nbr.shape
    if order is not None:
        nbr = nbr[:, order]
    nt = (n + tile - 1) // tile
    uniq, pairs, ex16 = [], [], 0
    for t in range(nt):
        blk = nbr[:, t * tile:(t + 1) * tile]
        v = blk[blk >= 0]
        uniq.append(np.unique(v).size); pairs.append(v.size)
    # executed (16-row group, tap) slots
    pad = (-n) % 16
    m = np.pad(nbr >= 0, ((0, 0), (0, pad))).reshape(kv, -1, 16).any(2)
    ex = m.sum() * 16
    uniq, pairs = np.array(uniq), np.array(pairs)
    full = uniq[:-1] if nt > 1 else uniq
    print("%-22s tile %4d: pairs/row %.2f  unique/row mean %.2f p95 %.2f max %.2f (rows: mean %d p95 %d max %d)  executed/useful %.3f" % (
        name, tile, pairs.sum() / n, uniq.sum() / n, np.percentile(full, 95) / tile, full.max() / tile, full.mean(), np.percentile(full, 95), full.max(), ex / pairs.sum()))

def pattern_order(nbr, chunk):
    kv, n = nbr.shape
    pat = np.zeros(n, np.int64)
    for t in range(kv):
        pat |= (nbr[t] >= 0).astype(np.int64) << t
    order = np.arange(n)
    for c0 in range(0, n, chunk):
        sl = slice(c0, min(n, c0 + chunk))
        order[sl] = c0 + np.argsort(pat[sl], kind="stable")
    return order

for name, idx, shape in levels:
    nbr = o.subm_rulebook(idx, 1, shape, [3, 3, 3])
    nbr = np.asarray(nbr)
    print("== %s: %d rows, shape %s" % (name, idx.shape[0], shape))
    for tile in (128, 256):
        stats(name + " canonical", nbr, tile)
    for chunk in (128, 256, 512, 4096):
        stats(name + " pattern/%d" % chunk, nbr, 128, pattern_order(nbr, chunk))

print("\n-- contiguous-range staging: per (tile, dz) the row range [min, max] of the 9 taps' neighbours")
for name, idx, shape in levels:
    nbr = np.asarray(o.subm_rulebook(idx, 1, shape, [3, 3, 3]))
    kv, n = nbr.shape
    for tile in (64, 128):
        nt = n // tile
        tot, spans = 0, []
        for dz in range(3):
            blk = nbr[dz * 9:(dz + 1) * 9, :nt * tile].reshape(9, nt, tile).transpose(1, 0, 2).reshape(nt, -1)
            big = np.where(blk >= 0, blk, 1 << 30).min(1)
            sm = np.where(blk >= 0, blk, -1).max(1)
            sp = np.where(sm >= 0, sm - big + 1, 0)
            spans.append(sp)
        spans = np.stack(spans, 1)            # [tiles, 3]
        per_tile = spans.sum(1)
        print("%s tile %3d: staged rows / tile row: mean %.2f  (per dz group: mean %d  p50 %d  p95 %d  p99 %d  max %d; groups > 512 rows: %.2f %%, > 1024: %.2f %%)" % (
            name, tile, np.median(per_tile) / tile, spans.mean(), np.percentile(spans, 50), np.percentile(spans, 95), np.percentile(spans, 99), spans.max(),
            100.0 * (spans > 512).mean(), 100.0 * (spans > 1024).mean()))

print("\n-- unique rows per (tile of 128, dz group of 9 taps), by row order")
for name, idx, shape in levels[1:]:
    nbr = np.asarray(o.subm_rulebook(idx, 1, shape, [3, 3, 3]))
    kv, n = nbr.shape
    for chunk in (0, 128, 512, 4096):
        nb = nbr
        if chunk:
            order = pattern_order(nbr, chunk)           # rows re-ordered: columns permuted AND ids renamed
            inv = np.empty(n, np.int64); inv[order] = np.arange(n)
            nb = nbr[:, order]
            nb = np.where(nb >= 0, inv[np.maximum(nb, 0)], -1)
        nt = n // 128
        cnt = np.zeros((nt, 3), np.int64)
        for t in range(nt):
            for dz in range(3):
                blk = nb[dz * 9:(dz + 1) * 9, t * 128:(t + 1) * 128]
                cnt[t, dz] = np.unique(blk[blk >= 0]).size
        print("%s %-13s unique/group: mean %5.1f p50 %3d p95 %3d p99 %3d max %3d  | > 192: %.2f %%  > 256: %.2f %%  > 320: %.2f %%   unique/tile-row %.2f" % (
            name, "pattern/%d" % chunk if chunk else "canonical", cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 95), np.percentile(cnt, 99), cnt.max(),
            100.0 * (cnt > 192).mean(), 100.0 * (cnt > 256).mean(), 100.0 * (cnt > 320).mean(), cnt.sum() / (nt * 128)))
