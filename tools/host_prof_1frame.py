import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, os.getcwd())
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
eng = CenterPointEngine(cfg, sd, host_results=True)
c = [torch.from_numpy(waymo_cloud(0)).cuda()]
for _ in range(20): eng.forward(c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): eng.forward(c)
torch.cuda.synchronize()
print("1 frame: %.3f ms/step" % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): eng.forward(c)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
