"""Where does the per-step D2H of the results go? (bench: 27.0 ms/step device-resident vs 33.4 with the host copy)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(8)]
for host, mode in ((False, ""), (True, "pinned_stream_sync"), (True, "cpu"), (True, "tolist_first"), (True, "pinned_event"), (True, "pinned_item"), (False, "")):
    os.environ["CPD_D2H_MODE"] = mode
    eng = CenterPointEngine(cfg, sd, host_results=host)
    for _ in range(3): eng.forward(clouds * 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.forward(clouds * 2)
    torch.cuda.synchronize(); print("host_results", host, mode, "ms/step %.2f" % ((time.perf_counter() - t0) * 100))
# isolated copies
ob = torch.randn(16, 500, 7, device="cuda"); pin = torch.empty(ob.shape).pin_memory()
for name, fn in [("pinned copy_ nonblocking + sync", lambda: (pin.copy_(ob, non_blocking=True), torch.cuda.current_stream().synchronize())),
                 (".cpu()", lambda: ob.cpu()), ("clone pinned", lambda: pin.clone()), ("tolist small", lambda: ob[0, 0].tolist())]:
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    print(name, "%.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
