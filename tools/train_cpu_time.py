"""How much of a train step is host time? (issue time of all launches vs. time to completion)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, cProfile, pstats
from cpd_amd.engine import ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud, gt_boxes
from cpd_amd.train_engine import CenterPointTrainer
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
tr = CenterPointTrainer(cfg, sd, total_steps=100)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pts = [torch.from_numpy(waymo_cloud(i)).cuda() for i in range(B)]
gt = torch.stack([torch.from_numpy(gt_boxes(i)) for i in range(B)]).cuda()
for _ in range(3):
    tr.step(pts, gt)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); tr.step(pts, gt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("issue %.1f ms   complete %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); tr.step(pts, gt); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
