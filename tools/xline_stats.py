"""Round 6 (VERDICT r5 #1b): how much of the sub-manifold levels' dx = +-1 gathers an x-line form could take from lane shifts of the
dx = 0 gather -- on the oracle's levels of the bench's synthetic cloud, in canonical and in tap-pattern row order. CPU only (oracle + numpy);
results in profiles/r06_gather_probes.txt."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle.binding import Oracle
import ref_pipeline as rp
from cpd_amd.engine import ModelConfig
from cpd_amd.synthetic import waymo_cloud
o = Oracle(); cfg = ModelConfig()
pts = waymo_cloud(0)
feats, coords = rp.voxelize_batch(o, cfg, [pts])
shape = cfg.sparse_shape
lev = {1: (coords, shape)}
c, s = coords, shape
for i, st in enumerate(["conv2", "conv3", "conv4"], start=2):
    k, sd, pd = rp.DOWN[st]
    c = o.conv_outset(c, 1, s, k, sd, pd); s = o.conv_out_shape(s, k, sd, pd)
    lev[i] = (c, s)
for L in (2, 3, 4):
    c, s = lev[L]
    n = len(c)
    D, H, W = s
    key = (c[:, 1].astype(np.int64) * H + c[:, 2]) * W + c[:, 3]
    assert (np.diff(key) > 0).all()
    lut = {int(k_): i for i, k_ in enumerate(key)}
    def nb(dz, dy, dx):
        z, y, x = c[:, 1] + dz, c[:, 2] + dy, c[:, 3] + dx
        ok = (z >= 0) & (z < D) & (y >= 0) & (y < H) & (x >= 0) & (x < W)
        kk = (z.astype(np.int64) * H + y) * W + x
        idx = np.searchsorted(key, kk); idx[idx >= n] = n - 1
        hit = ok & (key[idx] == kk)
        return np.where(hit, idx, -1)
    nbr = {(dz, dy, dx): nb(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)}
    pat = np.zeros(n, np.int64)
    for t, kx in enumerate(sorted(nbr)):
        pat |= (nbr[kx] >= 0).astype(np.int64) << t
    print("level %d: %d rows, avg nbrs %.2f" % (L, n, np.mean([ (nbr[k] >= 0).mean() for k in nbr]) * 27))
    for order_name in ("canonical", "taps4096"):
        if order_name == "canonical":
            new_to_old = np.arange(n)
        else:
            new_to_old = np.concatenate([st0 + np.argsort(pat[st0:st0 + 4096], kind="stable") for st0 in range(0, n, 4096)])
        # per 16-row tile stats
        tot = hit = zero_ok = need_zero = 0
        tile_all = tile_cnt = 0
        gathers_now = gathers_new = 0
        for (dz, dy) in [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]:
            c0 = nbr[(dz, dy, 0)][new_to_old]
            for dx in (-1, 1):
                cx = nbr[(dz, dy, dx)][new_to_old]
                # row r's dx-neighbour vs row r+dx's centre neighbour (same 16-tile)
                r = np.arange(n)
                rs = r + dx
                inside = (rs >= 0) & (rs < n) & ((rs // 16) == (r // 16))
                shifted = np.where(inside, c0[np.clip(rs, 0, n - 1)], -2)
                want = cx >= 0
                ok = want & (shifted == cx)
                tot += want.sum(); hit += ok.sum()
                # tiles: active (any want) and fully satisfied by shift (all lanes: shifted == cx or (cx == -1 and we can zero))
                nt = (n + 15) // 16
                padn = nt * 16 - n
                w16 = np.pad(want, (0, padn)).reshape(nt, 16)
                ok16 = np.pad(ok, (0, padn)).reshape(nt, 16)
                act = w16.any(1)
                tile_cnt += act.sum(); tile_all += (act & (w16 == ok16).all(1)).sum()
                miss = (w16 & ~ok16).sum(1)
                gathers_now += act.sum() * 16
                gathers_new += miss[act].sum()
        print("  %-10s dx=+-1 pairs %8d  shift-reusable %.3f   active (tile, dx tap) %7d  fully reusable %.3f   patch rows / tile-tap %.2f of 16" %
              (order_name, tot, hit / tot, tile_cnt, tile_all / tile_cnt, gathers_new / tile_cnt))
