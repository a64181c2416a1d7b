"""N > 1 path on CPU: world_size-2 gloo run of the frame sharding + barrier + max-over-ranks timing
that bench.py uses with RCCL on the GPUs (the forward path has no data-path collective), and of the
train step's one collective: the flat-gradient all-reduce + averaging."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from cpd_amd import dist_utils as du
    assert du.init("gloo")
    seeds = du.frame_seeds(rank, 4)
    du.barrier()
    elapsed = 1.0 + rank            # rank 1 is the slow one
    mx = du.max_over_ranks(elapsed)
    thr = du.aggregate_throughput(units_per_rank=10, elapsed_local=elapsed)
    import torch
    grad = torch.full((1000,), float(rank + 1))          # rank r holds gradient r+1 everywhere
    scale = du.reduce_gradients(grad, world)
    assert scale == 0.5 and torch.all(grad == 3.0)       # sum over ranks, mean = sum * scale
    # the bucketed, overlapped form (train_engine: the dense half's bucket starts while the sparse half back-propagates) == one all-reduce
    g1 = torch.arange(1000, dtype=torch.float32) * (rank + 1) + rank
    g2 = g1.clone()
    du.reduce_gradients(g1, world)
    br = du.BucketedReduce(g2, world)
    br.start(600, 1000)
    br.start(100, 200)
    assert br.finish() == 0.5 and torch.equal(g1, g2)
    du.barrier()
    q.put((rank, seeds, mx, thr))
    du.shutdown()


def test_two_rank_frame_sharding_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, m0, t0), (r1, s1, m1, t1) = out
    assert set(s0).isdisjoint(s1) and len(s0) == len(s1) == 4      # disjoint frame shards
    assert m0 == m1 == 2.0                                          # max over ranks
    assert abs(t0 - 10.0) < 1e-9 and t0 == t1                       # 2 ranks * 10 units / 2.0 s


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (the flag used to be
    parsed and ignored). --launch-check runs the launcher, the rendezvous and the timing collectives only (gloo here,
    RCCL on the GPUs), so this runs without a GPU."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CPD_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                     # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["gpus_arg"] == 2 and rec["max_over_ranks"] == 2.0
    # and under an external launcher (the driver's way) the same command does not spawn again
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--launch-check"], env=env2,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_eight_rank_launch_check_pins_cpus_and_reduces_the_gradient_buffer():
    """`python bench.py --gpus 8 --launch-check` (the first 8-GPU run should be boring): the command starts its eight
    ranks itself, every rank pins itself to its own slice of the node's CPUs (dist_utils.rank_cpus: NUMA order, disjoint slices),
    the per-rank clocks of the bench line come back in rank order, whole-job throughput is all ranks' units over the SLOWEST rank's time,
    and the train step's bucketed all-reduce of a gradient-sized buffer gives the mean on every element. gloo here, RCCL on the GPUs."""
    import json
    import subprocess
    import sys
    from cpd_amd import dist_utils
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(CPD_DIST_BACKEND="gloo", CPD_LAUNCH_CHECK_GRAD_FLOATS="1000000", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["gpus_arg"] == 8 and rec["max_over_ranks"] == 8.0 and rec["backend"] == "gloo"
    pr = rec["per_rank"]
    assert pr["elapsed_s"] == [1.0 + 0.25 * r for r in range(8)]                    # every rank's own clock, in rank order
    assert abs(rec["whole_job_units_per_s"] - 8 * 48.0 / 2.75) < 1e-9                # ... and the slowest one prices the job
    assert len(pr["allreduce_in_step_ms"]) == 8 and all(v > 0 for v in pr["allreduce_in_step_ms"])
    assert rec["allreduce_mean"] == [rec["expected_mean"]] * 2 == [4.5, 4.5]        # both buckets reduced and averaged
    n_cpu = len(os.sched_getaffinity(0))
    aff = pr["cpu_affinity"]
    if n_cpu >= 8:
        assert all(a["cpus"] == n_cpu // 8 for a in aff), aff
        spans = sorted((a["first"], a["last"]) for a in aff)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans          # disjoint slices
    # the slicing itself: equal, disjoint, NUMA-ordered; fewer CPUs than ranks = no pinning
    slices = [dist_utils.rank_cpus(r, 4) for r in range(4)]
    assert len({len(s) for s in slices}) == 1 and len({c for s in slices for c in s}) == sum(len(s) for s in slices)
    assert dist_utils.rank_cpus(0, 10 * n_cpu) == [c for node in dist_utils._numa_cpu_lists() for c in node]
