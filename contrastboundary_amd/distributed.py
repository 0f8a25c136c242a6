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
        self.grouped = bool(dist.is_available() and dist.is_initialized())          # a process group exists: the collective is ISSUED, whatever its size
        self.world = dist.get_world_size(process_group) if self.grouped else 1
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
        if self.grouped:                                             # also on a ONE-rank group: hooks + flat buffer + RCCL run together (no special case to hide behind)
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


class FlatState:
    """The trainable parameters of `modules` as views of ONE flat fp32 buffer per optimizer param group, their gradients as views of a second one, the
    floating-point buffers (BatchNorm running statistics) as views of a third — what makes a data-parallel step cost what a single-GPU step costs:

      * gradients: the captured backward leaves every `.grad` where autograd put it (no per-parameter accumulate kernel), `pack()` gathers them into the flat
        gradient buffer with a handful of multi-tensor copies (inside the graph), and the bucketed all-reduce runs over slices of that buffer;
      * optimizer: `flat_optimizer(opt)` is the caller's optimizer re-created over ONE Parameter per param group (the flat buffer, `.grad` = the flat gradient):
        the update of 300 tensors is one fused kernel.  Valid for optimizers whose update is elementwise (SGD / momentum / weight decay, Adam, AdamW: every
        optimizer the reference configures, train.py:154); hyper-parameters are re-read from the caller's optimizer before every step (LR schedulers keep working);
      * buffers: DDP re-broadcasts rank 0's buffers at the start of every forward (broadcast_buffers=True, train.py:181-189): here ONE broadcast of the flat buffer
        (int64 batch counters ride in a second, tiny one).

    Build it AFTER the modules are on their device (a later .cuda() / .to() re-allocates every tensor and drops the views) and BEFORE any graph capture."""

    ALIGN = 64                                                       # elements

    @classmethod
    def _triples_adjacent(cls, group, modules):
        """`group` reordered so that the q / k / v weights (and biases) of every attention layer follow one another: with sizes that are multiples of ALIGN they
        then lie back to back in the flat buffer, and pt_layer._stacked takes the three as ONE (3, ...) view for its batched product"""
        from .blocks import PointTransformerLayer
        pos = {id(p): i for i, p in enumerate(group)}
        follow, taken = {}, set()
        for m in modules:
            for layer in m.modules():
                if not isinstance(layer, PointTransformerLayer):
                    continue
                for name in ("weight", "bias"):
                    t = [getattr(l, name) for l in (layer.linear_q, layer.linear_k, layer.linear_v)]
                    if (all(x is not None and id(x) in pos and id(x) not in taken for x in t) and len({id(x) for x in t}) == 3
                            and t[0].shape == t[1].shape == t[2].shape and t[0].numel() % cls.ALIGN == 0):
                        follow[id(t[0])] = t[1:]
                        taken.update(id(x) for x in t)
        out = []
        for p in group:
            if id(p) in taken and id(p) not in follow:
                continue                                             # placed behind its q
            out.append(p)
            out.extend(follow.get(id(p), ()))
        return out

    def __init__(self, modules, optimizer=None):
        seen, groups = set(), []
        if optimizer is not None:
            for g in optimizer.param_groups:
                ps = [p for p in g["params"] if p.requires_grad and id(p) not in seen]
                seen.update(id(p) for p in ps)
                groups.append(ps)
        rest = [p for m in modules for p in m.parameters() if p.requires_grad and id(p) not in seen]
        uniq = []
        for p in rest:
            if id(p) not in seen:
                seen.add(id(p)); uniq.append(p)
        if uniq:
            assert optimizer is None, "trainable parameters the optimizer does not hold"
            groups.append(uniq)
        self.groups = [self._triples_adjacent(g, modules) for g in groups if g]
        assert self.groups, "no trainable parameters"
        dev, dt = self.groups[0][0].device, self.groups[0][0].dtype
        assert all(p.device == dev and p.dtype == dt for g in self.groups for p in g), "one device and one dtype"
        self.params = [p for g in self.groups for p in g]
        # every tensor starts on a 256-byte boundary of the flat buffer: the kernels take parameters by raw pointer and load them 16 bytes at a time
        # (cbl_host_aligned16 in the C entries); the padding holds zeros for ever (zero gradient, weight decay of 0 is 0)
        al = self.ALIGN
        total = sum((p.numel() + al - 1) // al * al for p in self.params)
        self.flat_param = torch.zeros(total, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total, dtype=dt, device=dev)
        self.group_range, self.grad_views = [], []
        pos = 0
        with torch.no_grad():
            for g in self.groups:
                start = pos
                for p in g:
                    n = p.numel()
                    view = self.flat_param[pos:pos + n].view(p.shape)
                    view.copy_(p.data)
                    p.data = view                                    # the Parameter object (and everything that references it) stays; its storage is the flat buffer now
                    self.grad_views.append(self.flat_grad[pos:pos + n].view(p.shape))
                    pos += (n + al - 1) // al * al
                self.group_range.append((start, pos))
        # buffers: floating point ones in one flat tensor, integer ones (num_batches_tracked) in another
        self.flat_buffers = {}
        bufs, seen_b = [], set()
        for m in modules:
            for mod in m.modules():
                for name, b in mod._buffers.items():
                    if b is not None and id(b) not in seen_b:
                        seen_b.add(id(b)); bufs.append((mod, name, b))
        for key, pick in (("float", lambda b: b.is_floating_point()), ("int", lambda b: not b.is_floating_point())):
            assert all(b.device == dev for _, _, b in bufs), "FlatState: buffers on another device than the parameters"
            mine = [(mod, name, b) for mod, name, b in bufs if pick(b)]
            if not mine:
                continue
            dtb = mine[0][2].dtype
            odd = [name for _, name, b in mine if b.dtype != dtb]
            assert not odd, "FlatState: %s buffers of more than one dtype (%s): they would drop out of broadcast_buffers" % (key, odd[:4])
            flat = torch.zeros(sum((b.numel() + 15) // 16 * 16 for _, _, b in mine), dtype=dtb, device=dev)
            pos = 0
            with torch.no_grad():
                for mod, name, b in mine:
                    n = b.numel()
                    view = flat[pos:pos + n].view(b.shape)
                    view.copy_(b)
                    mod._buffers[name] = view
                    pos += (n + 15) // 16 * 16
            self.flat_buffers[key] = flat

    # ---- gradients
    def drop_grads(self):
        """before a backward pass whose gradients `pack()` will collect: autograd then WRITES each gradient (a fresh tensor) instead of accumulating into one"""
        for p in self.params:
            p.grad = None

    def pack(self):
        """every parameter's gradient into its slice of the flat gradient buffer: multi-tensor copies (a few launches for hundreds of tensors); parameters that
        received none contribute zeros"""
        have = [(v, p.grad) for v, p in zip(self.grad_views, self.params) if p.grad is not None]
        self._idle = [p for p in self.params if p.grad is None]      # left alone by step()
        if len(have) != len(self.params):
            self.flat_grad.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    # ---- optimizer
    def flat_optimizer(self, optimizer):
        """the same optimizer class and hyper-parameters over one Parameter per param group (storage: the flat buffers)"""
        assert len(optimizer.param_groups) == len(self.groups), "FlatState was built for another optimizer"
        self._src_optimizer = optimizer
        self.flat_tensors = []
        groups = []
        for g, (s0, s1) in zip(optimizer.param_groups, self.group_range):
            P = torch.nn.Parameter(self.flat_param[s0:s1], requires_grad=True)
            P.grad = self.flat_grad[s0:s1]
            self.flat_tensors.append(P)
            groups.append(dict({k: v for k, v in g.items() if k != "params"}, params=[P]))
        flat = type(optimizer)(groups)
        self.optimizer = flat
        self._import_state()                                         # state the caller's optimizer already holds (warm-up steps, a loaded checkpoint) moves over
        return flat

    def _slices(self, gi):
        """(parameter, start, stop) of group gi's parameters inside the group's flat range"""
        out, pos = [], 0
        for p in self.groups[gi]:
            out.append((p, pos, pos + p.numel()))
            pos += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        return out

    def _import_state(self):
        """per-parameter optimizer state of the caller's optimizer -> the flat optimizer's one-tensor-per-group state.  Elementwise state tensors (SGD's
        momentum_buffer, Adam's exp_avg / exp_avg_sq: same shape as the parameter) are laid into a flat tensor at the parameter's offset (zeros where a parameter
        has none yet); per-parameter scalars (Adam's `step`) must agree inside a group and are taken once.  Anything else cannot be carried and raises."""
        src = self._src_optimizer
        for gi, (P, (s0, s1)) in enumerate(zip(self.flat_tensors, self.group_range)):
            per = [(p, a, b, src.state.get(p, {})) for p, a, b in self._slices(gi)]
            keys = []
            for _, _, _, st in per:
                keys.extend(k for k in st if k not in keys)
            fstate = {}
            for k in keys:
                vals = [st[k] for _, _, _, st in per if k in st and st[k] is not None]
                if not vals:
                    continue
                if all(torch.is_tensor(v) and v.dim() > 0 for v in vals) and all(v.numel() == p.numel() for p, _, _, st in per if st.get(k) is not None for v in [st[k]]):
                    flat_v = torch.zeros(s1 - s0, dtype=vals[0].dtype, device=P.device)
                    for p, a, b, st in per:
                        if st.get(k) is not None:
                            flat_v[a:b].copy_(st[k].reshape(-1))
                    fstate[k] = flat_v
                elif all((torch.is_tensor(v) and v.dim() == 0) or isinstance(v, (int, float)) for v in vals):
                    nums = {float(v) for v in vals}
                    if len(nums) != 1 or len(vals) != len(per):
                        raise ValueError("FlatState: optimizer state %r differs between the parameters of one group (%s): it cannot be carried by one flat tensor" % (k, sorted(nums)))
                    fstate[k] = vals[0].clone() if torch.is_tensor(vals[0]) else vals[0]
                else:
                    raise ValueError("FlatState: optimizer state %r is neither elementwise nor a per-parameter scalar; this optimizer cannot run on the flat buffers" % k)
            if fstate:
                self.optimizer.state[P] = fstate
            elif P in self.optimizer.state:
                del self.optimizer.state[P]

    def _export_state(self):
        """the flat optimizer's state written through to the caller's optimizer in its own per-parameter format (copies: a checkpoint taken from
        `optimizer.state_dict()` afterwards holds the live momentum)"""
        src = self._src_optimizer
        for gi, P in enumerate(self.flat_tensors):
            fstate = self.optimizer.state.get(P, {})
            for p, a, b in self._slices(gi):
                st = src.state[p]                                    # defaultdict: creates the entry
                for k, v in fstate.items():
                    if torch.is_tensor(v) and v.dim() > 0:
                        st[k] = v[a:b].view(p.shape).clone()
                    else:
                        st[k] = v.clone() if torch.is_tensor(v) else v

    def state_dict(self):
        """the optimizer state in the reference's checkpoint format ('optimizer': optimizer.state_dict(), /root/reference/pytorch/tool/train.py:216): the caller's
        optimizer with the live flat state written through to its per-parameter entries"""
        self._export_state()
        return self._src_optimizer.state_dict()

    def load_state_dict(self, state_dict):
        """resume (train.py:292 optimizer.load_state_dict): into the caller's optimizer, then into the flat one"""
        self._src_optimizer.load_state_dict(state_dict)
        self._import_state()

    def step(self):
        """one optimizer step on the flat buffers; the caller's optimizer is the source of truth for the hyper-parameters (an LR scheduler steps THAT one).
        Parameters that received no gradient in the packed backward pass keep their value and state: torch's optimizers skip a parameter whose .grad is None
        (DDP with find_unused_parameters, train.py:184), while the fused flat step would decay and move them."""
        for gs, gf in zip(self._src_optimizer.param_groups, self.optimizer.param_groups):
            for k, v in gs.items():
                if k != "params":
                    gf[k] = v
        idle = getattr(self, "_idle", ())
        keep = []
        if idle:
            offs = {}
            for gi in range(len(self.groups)):
                for p, a, b in self._slices(gi):
                    offs[id(p)] = (gi, a, b)
            for p in idle:
                gi, a, b = offs[id(p)]
                P = self.flat_tensors[gi]
                saved = [(P.data[a:b], P.data[a:b].clone())]
                for v in self.optimizer.state.get(P, {}).values():
                    if torch.is_tensor(v) and v.dim() > 0:
                        saved.append((v[a:b], v[a:b].clone()))
                keep.append(saved)
        self.optimizer.step()
        for saved in keep:
            for dst, val in saved:
                dst.copy_(val)

    # ---- buffers
    def broadcast_buffers(self, src=0, group=None):
        """rank `src`'s buffers on every rank: one collective per flat buffer (two in all)"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        for flat in self.flat_buffers.values():
            dist.broadcast(flat, src, group=group)
        return len(self.flat_buffers)


class PackedGradientReducer:
    """GradientReducer's role for a step whose gradients arrive PACKED (FlatState.pack inside a replayed hipGraph): the flat gradient buffer is cut into
    buckets of `bucket_bytes` in parameter order and `reduce_all()` issues one all-reduce per bucket, in order, behind the replay, then averages.  The
    collective is issued whenever a process group exists — also a one-rank group, so that a 1-GPU box runs RCCL, the packing and the graph together."""

    def __init__(self, state, bucket_bytes=8 << 20, process_group=None):
        import torch.distributed as dist
        self.dist, self.group, self.state = dist, process_group, state
        self.flat = state.flat_grad
        self.grouped = bool(dist.is_available() and dist.is_initialized())
        self.world = dist.get_world_size(process_group) if self.grouped else 1
        per = max(1, int(bucket_bytes) // self.flat.element_size())
        n = self.flat.numel()
        self.buckets = [(s, min(n, s + per), ()) for s in range(0, n, per)]
        self.params = state.params
        self.handles = []

    def reduce_all(self):
        if not self.grouped:
            return
        works = [self.dist.all_reduce(self.flat[s:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for s, e, _ in self.buckets]
        for w in works:
            w.wait()                                                 # the current stream waits for the collective's stream; the host does not block (NCCL / RCCL backend)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)

    def zero_grad(self):
        self.state.drop_grads()

    def finish(self):
        """eager use (warm-up steps): pack what the backward produced, reduce"""
        self.state.pack()
        self.reduce_all()

    def rehook(self):
        pass

    def remove(self):
        pass
