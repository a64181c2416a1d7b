#!/usr/bin/env python3
"""Soak of the side-stream engine: batch sizes alternate (1, 4, 48, 3, 16, ...) through ONE engine for many steps; every repeat of a
batch must reproduce its first digest, and the allocator's reserved memory must settle (side-stream pools are per stream: a leak or an
unbounded pool would show as growth)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd.digest import step_digest
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud
from concurrent.futures import ThreadPoolExecutor
N = int(os.environ.get("ROUNDS", "30"))
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
with ThreadPoolExecutor(8) as ex:
    clouds = [torch.from_numpy(c).cuda() for c in ex.map(waymo_cloud, range(48))]
eng = CenterPointEngine(cfg, sd, host_results=True)
sizes = [1, 4, 48, 3, 16, 2, 8]
first = {}
res0 = None
t0 = time.perf_counter()
for r in range(N):
    for b in sizes:
        d = step_digest(eng.forward(clouds[:b]))
        if b not in first:
            first[b] = d
        assert d == first[b], "round %d, batch %d: digest changed" % (r, b)
    torch.cuda.synchronize()
    res = torch.cuda.memory_reserved() / 2**30
    if r == 4:
        res0 = res
    if r % 5 == 0 or r == N - 1:
        print("round %3d  reserved %.2f GiB  allocated %.2f GiB  (%.1f s)" % (r, res, torch.cuda.memory_allocated() / 2**30, time.perf_counter() - t0), flush=True)
assert res <= res0 * 1.10 + 0.5, "reserved memory keeps growing: %.2f -> %.2f GiB" % (res0, res)
print("soak ok: %d rounds x %s frames, digests stable, reserved %.2f -> %.2f GiB" % (N, sizes, res0, res))
