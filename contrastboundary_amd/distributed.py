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
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.SUM)
    return float(t.item())


def group_ranks():
    """number of ranks of the initialised default process group (1 when there is none) and its backend name"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_backend()
    return 1, None


def aggregate_throughput(units_this_rank, elapsed_this_rank):
    """whole-job throughput = sum of units over ranks / max elapsed over ranks (bench.py contract)"""
    total = reduce_scalar(units_this_rank, "sum")
    worst = reduce_scalar(elapsed_this_rank, "max")
    return total / worst, total, worst


class GradientReducer:
    """Data-parallel gradient averaging for one model replica per GPU — what DistributedDataParallel does for the reference
    (/root/reference/pytorch/tool/train.py:181-185), written for this node: every parameter's `.grad` is a VIEW into one flat fp32 buffer (no
    bucket copies), the buffer is cut into buckets in reverse parameter order (the order backward produces gradients in), and a bucket's
    all-reduce (RCCL over xGMI when the backend is "nccl", gloo on CPU) is started from autograd's post-accumulate hooks as soon as its last
    gradient has been written — so the collective of the deep layers runs beside the backward kernels of the shallow ones.  Buckets are launched
    strictly in bucket order on every rank (a bucket that is ready early waits for the ones before it), which keeps the collective sequence
    identical across ranks whatever order the hooks fire in; parameters that received no gradient in a step contribute zeros
    (`finish()` launches what is still pending).

        red = GradientReducer(model.parameters(), bucket_bytes=...)      # after the model is on its device, before the optimizer's first step
        loss.backward(); red.finish(); optimizer.step(); red.zero_grad()

    bucket_bytes: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound and wants FEW, LARGE messages:
    the default 8 MiB gives 4 buckets for the reference network's 31.2 MB of gradients — enough to overlap, large enough to stay bandwidth-bound.
    graph mode (`hooks=False`): when the backward is replayed from a hipGraph no hook fires; `reduce_all()` then issues every bucket in order
    after the replay (the all-reduce runs beside the next batch's geometry on its side streams, not beside its own backward)."""

    def __init__(self, params, bucket_bytes=8 << 20, process_group=None, hooks=True):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), "one device and one dtype per reducer"
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # flat layout in REVERSE parameter order: bucket 0 holds the last layers, whose gradients arrive first
        order = list(reversed(range(len(self.params))))
        sizes = [self.params[i].numel() for i in order]
        self.flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
        per = max(1, int(bucket_bytes) // self.flat.element_size())
        self.buckets, self.bucket_of = [], {}
        start = pos = 0
        members = []
        for i, nel in zip(order, sizes):
            p = self.params[i]
            p.grad = self.flat[pos:pos + nel].view_as(p)
            members.append(i)
            self.bucket_of[i] = len(self.buckets)
            pos += nel
            if pos - start >= per:
                self.buckets.append((start, pos, tuple(members)))
                start, members = pos, []
        if members:
            self.buckets.append((start, pos, tuple(members)))
        self.pending = [len(b[2]) for b in self.buckets]
        self.next_bucket = 0
        self.works = []
        self.handles = []
        if hooks:
            self.rehook()

    def rehook(self):
        """(re-)register the post-accumulate hooks that start a bucket's all-reduce during an eager backward — a GraphedTrainStep removes them
        for its capture (no collective inside a hipGraph); a reducer that goes back to eager steps afterwards calls this to get the overlap back"""
        self.remove()
        for i, p in enumerate(self.params):
            self.handles.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._arrived(i)))

    # ---- collective issue
    def _launch(self, k):
        s, e, _ = self.buckets[k]
        if self.world > 1:
            self.works.append(self.dist.all_reduce(self.flat[s:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _arrived(self, i):
        k = self.bucket_of[i]
        self.pending[k] -= 1
        while self.next_bucket < len(self.buckets) and self.pending[self.next_bucket] <= 0:
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def reduce_all(self):
        """graph mode: every bucket, in order, now (gradients already complete)"""
        self.next_bucket = 0
        self.pending = [0] * len(self.buckets)
        self.finish()

    def finish(self):
        """launch what no hook has launched (unused parameters), wait for every bucket on the current stream, average.  No host block on a GPU:
        Work.wait() of the NCCL/RCCL backend makes the current stream wait for the collective's stream."""
        while self.next_bucket < len(self.buckets):
            self._launch(self.next_bucket)
            self.next_bucket += 1
        for w in self.works:
            w.wait()
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        self.works = []
        self.pending = [len(b[2]) for b in self.buckets]
        self.next_bucket = 0

    def zero_grad(self):
        """gradients back to zero in ONE fill (and still views of the flat buffer: `optimizer.zero_grad(set_to_none=True)` would drop them)"""
        self.flat.zero_()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                self._rebind()
                break

    def _rebind(self):
        for k, (s, e, members) in enumerate(self.buckets):
            pos = s
            for i in members:
                p = self.params[i]
                p.grad = self.flat[pos:pos + p.numel()].view_as(p)
                pos += p.numel()

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []


def broadcast_parameters(module, src=0):
    """every rank starts from rank `src`'s weights and buffers (DDP's constructor does the same, train.py:181)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def broadcast_buffers(modules, src=0):
    """rank `src`'s BUFFERS (BatchNorm running statistics, batch counters) on every rank: DistributedDataParallel re-broadcasts them at the start of
    every forward (broadcast_buffers=True, its default, which the reference keeps: train.py:181-189), so every rank validates and checkpoints the same
    running statistics.  One flat buffer per dtype: a handful of small collectives per call."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    bufs = [b for m in modules for b in m.buffers()]
    n = 0
    with torch.no_grad():
        for dt in sorted({b.dtype for b in bufs}, key=str):
            group = [b for b in bufs if b.dtype == dt]
            flat = torch.cat([b.reshape(-1) for b in group])
            dist.broadcast(flat, src)
            pos = 0
            for b in group:
                b.copy_(flat[pos:pos + b.numel()].view_as(b)); pos += b.numel()
            n += len(group)
    return n
