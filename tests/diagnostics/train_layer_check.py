"""Debug: inside a real train step, check every layer's BatchNorm backward and weight gradient
against torch formulas evaluated on the same saved tensors (GPU, fp64)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from test_gpu_train import small_cfg, scene
from cpd_amd.engine import init_state_dict
from cpd_amd import train_engine, train_ops
from cpd_amd.train_engine import CenterPointTrainer, _Conv

orig = _Conv.backward
SNAP = {}


def checked(self, dy, nbr_adj, n_in, need_dx=True, add=None, dx_out=None):
    x, nbr, n_out, z, y, mean, invstd, has_res, dense, up_map = self.saved
    st = self.store
    dyc = dy.clone()
    res = orig(self, dy, nbr_adj, n_in, need_dx, add, dx_out)
    for nm in [self.wn, self.bn_] + ([self.gn, self.be] if self.has_bn else []):
        if nm:
            SNAP[nm] = st.g(nm).clone()
    if self.has_bn and not (self.mode == "up" and self.up > 1):
        g = dyc.double()
        if self.relu:
            g = g * (y > 0)
        xh = (z.double() - mean.double()) * invstd.double()
        dbeta, dgamma = g.sum(0), (g * xh).sum(0)
        e1 = (st.g(self.be).double() - dbeta).abs().max() / dbeta.abs().max().clamp_min(1e-6)
        e2 = (st.g(self.gn).double() - dgamma).abs().max() / dgamma.abs().max().clamp_min(1e-6)
        n = z.shape[0]
        dz = st.p(self.gn).double() * invstd.double() * (g - dbeta / n - xh * dgamma / n)
        # weight gradient from the reference dz
        if nbr is not None:
            idx = torch.where(nbr < 0, x.shape[0], nbr).long()
            xp = torch.cat([x[:, :self.c_in].double(), x.new_zeros(1, self.c_in).double()])
            dw = torch.stack([xp[idx[t]].T @ dz for t in range(self.kv)])
        else:
            dw = (x[:, :self.c_in].double().T @ dz)[None]
        e3 = (st.g(self.wn).double() - dw).abs().max() / dw.abs().max()
        e4 = -1.0
        if need_dx and res[0] is not None and self.mode in ("same", "strided"):
            w = st.p(self.wn).double()                                   # [kv, ci, co]
            tbl = nbr_adj if nbr_adj is not None else torch.arange(n_in, device=z.device, dtype=torch.int32)[None]
            idx = torch.where(tbl < 0, n, tbl).long()
            dzp = torch.cat([dz, dz.new_zeros(1, self.c_out)])
            dx = dz.new_zeros(n_in, self.c_in)
            for t in range(self.kv):
                wt = w[self.kv - 1 - t] if self.mode == "same" else w[t]
                dx += dzp[idx[t]] @ wt.T
            if add is not None:
                dx += add.double()
            e4 = float((res[0].double() - dx).abs().max() / dx.abs().max())
        flag = "  <<<<" if max(e1, e2, e3, e4) > 2e-3 else ""
        print("%-40s n %6d c %3d->%3d  dbeta %.1e dgamma %.1e dw %.1e dx %.1e%s" % (self.name, n, self.c_in, self.c_out, e1, e2, e3, e4, flag))
    return res


ACT = {}
origf = _Conv.forward
def fwd(self, x, nbr, n_out, **kw):
    y = origf(self, x, nbr, n_out, **kw)
    if self.saved[3] is not None:
        ACT[self.name] = (self.saved[3], y)
    return y
_Conv.forward = fwd
_Conv.backward = checked
cfg = small_cfg(); sd = init_state_dict(cfg, seed=3); pts, gt = scene()
tr = CenterPointTrainer(cfg, sd, num_max_objs=50)
tr.forward_backward([torch.from_numpy(p).cuda() for p in pts], torch.from_numpy(gt).cuda())

for nm, v in SNAP.items():
    d = (tr.store.g(nm) - v).abs().max().item()
    if d > 0:
        print("OVERWRITTEN after its backward:", nm, d, v.abs().max().item())
print("snap check done", len(SNAP))

import ref_train_torch as R
from oracle.binding import Oracle
P = R.make_leaves(sd); taps = {}
R.forward_loss(Oracle(), cfg, P, pts, gt, 50, taps=taps)
for k, (z, y) in taps.items():
    hz, hy = ACT[k]
    print("%-36s z err %.2e (max %.1f)  y err %.2e   mask mismatches %d" % (k, (hz.cpu().double() - z).abs().max(), z.abs().max(),
          (hy.cpu().double() - y).abs().max(), int(((hy.cpu() > 0) != (y > 0)).sum())))
