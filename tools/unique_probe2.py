#!/usr/bin/env python3
"""unique input rows per 128-row tile for brick row orders (rows sorted by (b, z / bz, y / by, x / bx, z, y, x))"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "unique_probe.py")).read().split("def stats")[0])

def reorder(nbr, order):
    n = nbr.shape[1]
    inv = np.empty(n, np.int64); inv[order] = np.arange(n)
    nb = nbr[:, order]
    return np.where(nb >= 0, inv[np.maximum(nb, 0)], -1)

def tile_stats(nb, tile=128):
    kv, n = nb.shape
    nt = n // tile
    u = np.zeros(nt, np.int64); g = np.zeros((nt, 3), np.int64)
    for t in range(nt):
        blk = nb[:, t * tile:(t + 1) * tile]
        u[t] = np.unique(blk[blk >= 0]).size
        for dz in range(3):
            b = blk[dz * 9:(dz + 1) * 9]
            g[t, dz] = np.unique(b[b >= 0]).size
    pad = (-n) % 16
    m = np.pad(nb >= 0, ((0, 0), (0, pad))).reshape(kv, -1, 16).any(2)
    return u, g, m.sum() * 16 / (nb >= 0).sum()

for name, idx, shape in levels[1:]:
    nbr = np.asarray(o.subm_rulebook(idx, 1, shape, [3, 3, 3]))
    n = nbr.shape[1]
    for (bz, by, bx) in [(1, 1, 1 << 20), (1, 4, 32), (1, 8, 16), (1, 8, 8), (2, 8, 8), (3, 8, 8), (2, 4, 16), (4, 4, 8)]:
        key = np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 3] // bx, idx[:, 2] // by, idx[:, 1] // bz, idx[:, 0]))
        nb = reorder(nbr, key)
        u, g, ex = tile_stats(nb)
        # then pattern-sort inside each 128-row tile
        pat = np.zeros(n, np.int64)
        for t in range(27):
            pat |= (nb[t] >= 0).astype(np.int64) << t
        o2 = np.arange(n)
        for c0 in range(0, n, 128):
            sl = slice(c0, min(n, c0 + 128)); o2[sl] = c0 + np.argsort(pat[sl], kind="stable")
        _, _, ex2 = tile_stats(reorder(nb, o2))
        print("%s brick z%d y%d x%-7d unique/row %.2f (p95 tile %d, max %d)  per dz group: mean %d p95 %d max %d   exec/useful %.3f -> %.3f with pattern sort in tile" % (
            name, bz, by, bx, u.sum() / (len(u) * 128), np.percentile(u, 95), u.max(), g.mean(), np.percentile(g, 95), g.max(), ex, ex2))
