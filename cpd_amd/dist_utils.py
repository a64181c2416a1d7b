"""One-process-per-GPU helpers for the frame-sharded multi-GPU path (bench.py, N > 1).

The forward path shards by FRAME (every stage is per batch element: SURVEY 8e), so there is no
data-path collective: ranks only meet at the timing barrier and at the max-over-ranks of the
elapsed time. Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU for the tests."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, script, argv, env=None):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start N ranks of the same command, one per
    GPU, under torch.distributed.run on this node -- the role of tools/dist_train.sh:3 (`torch.distributed.launch
    --nproc_per_node=N`) and tools/train.py:63-65 (`init_dist_pytorch`) in the reference. Rendezvous on 127.0.0.1 (the
    container hostname may not resolve). Returns the launcher's exit code."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=e)


def _numa_cpu_lists():
    """[[cpu ids of NUMA node 0], [node 1], ...] from sysfs, restricted to the CPUs this process may run on; one list when sysfs has
    no node information"""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    nodes = []
    try:
        base = "/sys/devices/system/node"
        names = sorted((d for d in os.listdir(base) if d.startswith("node") and d[4:].isdigit()), key=lambda d: int(d[4:]))
        for d in names:
            cpus = []
            with open(os.path.join(base, d, "cpulist")) as f:
                for part in f.read().strip().split(","):
                    if not part:
                        continue
                    lo, _, hi = part.partition("-")
                    cpus.extend(range(int(lo), int(hi or lo) + 1))
            cpus = [c for c in cpus if c in set(allowed)]
            if cpus:
                nodes.append(cpus)
    except OSError:
        nodes = []
    covered = {c for n in nodes for c in n}
    if not nodes or covered != set(allowed):
        return [allowed]
    return nodes


def rank_cpus(local_rank, local_world):
    """The CPUs rank `local_rank` of `local_world` ranks on this node should run on: the node's CPUs in NUMA order cut into
    `local_world` equal contiguous slices -- ranks sharing a NUMA node split it, a rank never straddles two nodes when the counts
    divide (8 ranks on a 2-socket box: four per socket, the GPUs' usual split). Each rank runs two Python worker threads + the HIP
    runtime's helper threads; without a slice of its own, 8 ranks' threads migrate across both sockets of the host."""
    cpus = [c for node in _numa_cpu_lists() for c in node]
    n = len(cpus)
    local_world = max(1, int(local_world))
    if n < local_world:
        return cpus                                     # fewer CPUs than ranks: no pinning
    per = n // local_world
    lo = (int(local_rank) % local_world) * per
    return cpus[lo:lo + per]


def pin_rank(local_rank=None, local_world=None):
    """sched_setaffinity of this rank to rank_cpus(...) (CPD_NO_PIN=1 switches it off; a launcher that already pinned the rank -- an
    affinity mask smaller than the node -- is left alone). -> the CPU list now in force."""
    if not hasattr(os, "sched_setaffinity"):
        return []
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world > 1 and not os.environ.get("CPD_NO_PIN"):
        cpus = rank_cpus(local_rank, local_world)
        if cpus:
            try:
                os.sched_setaffinity(0, cpus)
                torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
            except OSError:
                pass
    return sorted(os.sched_getaffinity(0))


def gather_ints(values, device="cpu"):
    """every rank's equally long list of ints, in rank order ([values] without a process group)"""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(v) for v in o.tolist()] for o in out]


def init(backend="nccl", device=None):
    rank, world, _ = env_rank()
    if world == 1 and not os.environ.get("CPD_FORCE_DIST"):     # CPD_FORCE_DIST=1: exercise the collectives with one rank
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # lazy communicator creation (no device_id): the first collective binds each rank to the GPU
    # selected with torch.cuda.set_device(LOCAL_RANK)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return True


def frame_seeds(rank, pool):
    """Disjoint synthetic-frame seeds per rank: rank r owns frames [r*pool, (r+1)*pool)."""
    return [rank * pool + i for i in range(pool)]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device="cpu"):
    """every rank's `value`, in rank order, on every rank ([value] without a process group)"""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def aggregate_throughput(units_per_rank, elapsed_local, device="cpu"):
    """Whole-job units/s: all ranks' units over the slowest rank's time."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return world * units_per_rank / max_over_ranks(elapsed_local, device)


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def reduce_gradients(flat_grad, world=None, group=None):
    """The train step's one collective: sum the flat gradient buffer over the data-parallel ranks in
    place (RCCL ring/direct all-reduce on the GPUs, gloo in the CPU tests) and return the factor that
    turns the sum into the mean (DistributedDataParallel's averaging, tools/train.py:143)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = world or dist.get_world_size(group)
    if world <= 1 and not os.environ.get("CPD_FORCE_DIST"):
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


class BucketedReduce:
    """The train step's gradient exchange in BUCKETS that overlap the backward pass (round 5; DDP's bucketing, tools/train.py:143): the
    backward produces gradients head -> BEV -> sparse, so the bucket of the dense half (the tail of the flat buffer) can be on the wire
    while the sparse half is still being back-propagated. `start(lo, hi)` launches the all-reduce of flat[lo:hi] asynchronously -- it
    is ordered after everything queued so far on the CURRENT stream --, `finish()` reduces whatever was not started (synchronously) and
    waits for the started ones. The result is the one all-reduce's, bit for bit (an all-reduce is element-wise). Without a process
    group (or with one rank) everything is a no-op and the factor is 1."""

    def __init__(self, flat_grad, world=None, group=None):
        self.flat, self.group = flat_grad, group
        self.active = dist.is_available() and dist.is_initialized()
        if self.active:
            world = world or dist.get_world_size(group)
            self.active = world > 1 or bool(os.environ.get("CPD_FORCE_DIST"))
        self.world = world if self.active else 1
        self.started = []                 # (lo, hi, work)

    def start(self, lo, hi):
        if not self.active or hi <= lo:
            return
        for a, b, _ in self.started:
            assert hi <= a or lo >= b, "gradient buckets must not overlap"
        work = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.started.append((lo, hi, work))

    def finish(self):
        """-> the factor that turns the sums into means"""
        if not self.active:
            return 1.0
        n = self.flat.numel()
        pos = 0
        for lo, hi, _ in sorted(self.started, key=lambda t: t[0]) + [(n, n, None)]:
            if lo > pos:
                dist.all_reduce(self.flat[pos:lo], op=dist.ReduceOp.SUM, group=self.group)
            pos = max(pos, hi)
        for _, _, work in self.started:
            work.wait()
        self.started = []
        return 1.0 / self.world
