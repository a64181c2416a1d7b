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
