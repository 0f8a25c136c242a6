"""Scene-per-GPU data parallelism for the hot path: every op is segmented by cloud, so ranks take
disjoint scenes and the path itself needs NO collective; torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU) is used only to agree on timing and to aggregate counts — the same role it has around the reference's
DistributedSampler (/root/reference/pytorch/tool/train.py:238) and metric all-reduces (:333-338)."""
import os

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """initialise the default process group from the torchrun environment; returns (world, rank, local_rank)"""
    import torch.distributed as dist
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_scenes(num_scenes, rank, world):
    """scene ids of this rank: round-robin like DistributedSampler without shuffling/padding; disjoint and covering"""
    return list(range(rank, num_scenes, world))


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_scalar(value, op="max", device=None):
    """max / sum of a python float over all ranks (identity when not distributed)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(units_this_rank, elapsed_this_rank):
    """whole-job throughput = sum of units over ranks / max elapsed over ranks (bench.py contract)"""
    total = reduce_scalar(units_this_rank, "sum")
    worst = reduce_scalar(elapsed_this_rank, "max")
    return total / worst, total, worst
