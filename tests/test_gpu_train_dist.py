"""N > 1 train step on real kernels: two ranks share the one GPU of the test box (RCCL refuses two ranks on one
device, so the process group is gloo, which all-reduces CUDA tensors through the host). Each rank trains on its
own frame; after every step both ranks must hold the SAME parameters (gradients were summed and averaged), equal
to a single process that averaged the two ranks' gradients itself."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cfg():
    from cpd_amd.engine import ModelConfig
    return ModelConfig(point_cloud_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], post_center_limit_range=[-20, -20, -2, 20, 20, 4],
                       bev_num_filters=[64, 128], bev_num_upsample_filters=[128, 128], bev_layer_nums=[1, 1], max_obj_per_sample=100)


def _frame(rank):
    from cpd_amd.synthetic import waymo_cloud
    pts = waymo_cloud(10 + rank, n_points=20000)
    pts[:, :2] *= 0.3
    rng = np.random.default_rng(rank)
    gt = np.zeros((1, 6, 8), np.float32)
    for i in range(5):
        gt[0, i] = [rng.uniform(-18, 18), rng.uniform(-18, 18), 0.8, 4.5, 2.0, 1.6, rng.uniform(-3, 3), rng.integers(1, 4)]
    return torch.from_numpy(pts).cuda(), torch.from_numpy(gt).cuda()


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from cpd_amd.engine import init_state_dict
    from cpd_amd.train_engine import CenterPointTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _cfg()
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=4), lr=1e-3, world_size=world, num_max_objs=20, grad_clip=0.0)
    pts, gt = _frame(rank)
    sums = []
    for _ in range(2):
        tr.step([pts], gt)
        sums.append(tr.store.flat.double().sum().item())
    q.put((rank, sums, tr.store.flat[:4096].cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_averages_gradients(hip):
    from cpd_amd.engine import init_state_dict
    from cpd_amd.train_engine import CenterPointTrainer
    from cpd_amd import train_ops
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, s0, w0), (_, s1, w1) = out
    assert s0 == s1                                              # identical parameters on both ranks after every step
    np.testing.assert_array_equal(w0, w1)
    # single process doing the two ranks' work and the averaging by hand (first step)
    cfg = _cfg()
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=4), lr=1e-3, num_max_objs=20, grad_clip=0.0)
    g = []
    for r in range(2):
        pts, gt = _frame(r)
        tr.forward_backward([pts], gt)
        g.append(tr.store.grad.clone())
    # BatchNorm running stats differ (they saw two frames) but parameters only depend on the averaged gradient
    tr.store.grad.copy_((g[0] + g[1]) * 0.5)
    tr.steps_done = 0
    tr.optimizer_step()
    assert abs(tr.store.flat.double().sum().item() - s0[0]) <= 1e-6 * abs(s0[0]) + 1e-4


# ---------------------------------------------------------------------------------------------------------------------------------
# SyncBatchNorm (tools/train.py:32,117 `--sync_bn`; round 6): statistics and the backward pass's two sums over all ranks' rows.

def _sync_bn_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from cpd_amd import train_ops
    from cpd_amd.engine import init_state_dict
    from cpd_amd.train_engine import CenterPointTrainer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (i) the layer arithmetic on ragged row counts (sparse levels hold different numbers of rows on every rank)
    x, dy, gamma, beta = _sync_bn_data()
    lo, hi = (0, 700) if rank == 0 else (700, 2000)
    xr, dyr = x[lo:hi].cuda().contiguous(), dy[lo:hi].cuda().contiguous()
    rm, rv = torch.zeros(48, device="cuda"), torch.ones(48, device="cuda")
    mean, invstd, scale, shift, n_total = train_ops.bn_stats_finalize_sync(xr, 1e-3, 0.01, gamma.cuda(), beta.cuda(), rm, rv)
    y = train_ops.affine_rows(xr, scale, shift, None, True)
    dx, dgamma, dbeta, _ = train_ops.bn_backward(dyr, y, xr, mean, invstd, gamma.cuda(), sync=(n_total, None))
    # (numpy through the queue: a torch tensor travels as a shared-memory handle that dies with this process)
    layer = dict(mean=mean.cpu().numpy(), invstd=invstd.cpu().numpy(), n_total=float(n_total.item()), y=y.cpu().numpy(), dx=dx.cpu().numpy(),
                 dgamma=dgamma.cpu().numpy(), dbeta=dbeta.cpu().numpy(), rm=rm.cpu().numpy(), rv=rv.cpu().numpy())
    # (ii) a whole train step with sync_bn: N ranks x 1 frame normalise like one process with the N frames in its batch
    cfg = _cfg()
    tr = CenterPointTrainer(cfg, init_state_dict(cfg, seed=4), lr=1e-3, world_size=world, num_max_objs=20, grad_clip=0.0, sync_bn=True)
    pts, gt = _frame(rank)
    tr.step([pts], gt)
    sd = tr.state_dict()
    q.put((rank, layer, {k: v.cpu().numpy() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")},
           tr.store.flat.double().sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def _sync_bn_data():
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2000, 48, generator=g) * 2.0 + 0.5
    dy = torch.randn(2000, 48, generator=g)
    return x, dy, torch.rand(48, generator=g) + 0.5, torch.randn(48, generator=g) * 0.1


def test_sync_batchnorm_matches_one_process_over_all_rows(hip):
    """Two ranks (700 / 1300 rows) through bn_stats_finalize_sync / bn_backward(sync=...) == one process over the 2000 rows: statistics,
    running statistics, outputs and the input gradient row for row; parameter gradients stay per-rank sums that add up to the whole.
    Then CenterPointTrainer(sync_bn=True) on two ranks x one frame: identical parameters on both ranks and, for EVERY BatchNorm of the
    model, the running statistics one process leaves after a forward pass over the two frames as one batch."""
    from cpd_amd import train_ops
    from cpd_amd.engine import init_state_dict
    from cpd_amd.train_engine import CenterPointTrainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sync_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, l0, rs0, sum0), (_, l1, rs1, sum1) = out
    l0, l1 = ({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in l.items()} for l in (l0, l1))
    rs0, rs1 = ({k: torch.from_numpy(v) for k, v in r.items()} for r in (rs0, rs1))
    x, dy, gamma, beta = _sync_bn_data()
    xc, dyc = x.cuda(), dy.cuda()
    rm, rv = torch.zeros(48, device="cuda"), torch.ones(48, device="cuda")
    mean, invstd, scale, shift = train_ops.bn_stats_finalize(xc, 1e-3, 0.01, gamma.cuda(), beta.cuda(), rm, rv)
    y = train_ops.affine_rows(xc, scale, shift, None, True)
    dx, dgamma, dbeta, _ = train_ops.bn_backward(dyc, y, xc, mean, invstd, gamma.cuda())
    close = lambda a, b, tol=2e-6: float((a - b.cpu()).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    for l in (l0, l1):
        assert l["n_total"] == 2000.0
        assert close(l["mean"], mean) and close(l["invstd"], invstd) and close(l["rm"], rm) and close(l["rv"], rv)
    assert close(torch.cat([l0["y"], l1["y"]]), y, 1e-5) and close(torch.cat([l0["dx"], l1["dx"]]), dx, 1e-5)
    assert close(l0["dgamma"] + l1["dgamma"], dgamma, 1e-5) and close(l0["dbeta"] + l1["dbeta"], dbeta, 1e-5)
    assert not close(l0["dgamma"], dgamma, 1e-3)                   # ... per-rank sums, not the totals
    # the whole model: both ranks hold the same parameters and the same running statistics ...
    assert sum0 == sum1
    assert rs0.keys() == rs1.keys() and len(rs0) >= 40
    for k in rs0:
        assert torch.equal(rs0[k], rs1[k]), k
    # ... which are those of ONE process whose batch is the two frames
    cfg = _cfg()
    one = CenterPointTrainer(cfg, init_state_dict(cfg, seed=4), lr=1e-3, num_max_objs=20, grad_clip=0.0)
    (p0, g0), (p1, g1) = _frame(0), _frame(1)
    one.forward([p0, p1])
    ref = {k: v.cpu() for k, v in one.state_dict().items() if k in rs0}
    worst = max(float((rs0[k] - ref[k]).abs().max()) / max(1e-3, float(ref[k].abs().max())) for k in rs0)
    assert worst <= 2e-4, worst
    local = CenterPointTrainer(cfg, init_state_dict(cfg, seed=4), lr=1e-3, num_max_objs=20, grad_clip=0.0)
    local.forward([p0])
    ref_local = {k: v.cpu() for k, v in local.state_dict().items() if k in rs0}
    assert max(float((rs0[k] - ref_local[k]).abs().max()) / max(1e-3, float(ref_local[k].abs().max())) for k in rs0) > 1e-3   # (not the local ones)
