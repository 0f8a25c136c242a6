"""Host mirror of the reference's segmentation network and criterion, /root/reference/pytorch/model/pointtransformer_seg.py:
    Loss :15-25, PointTransformerSeg :27-143, pointtransformer_seg_repro :146-150, with MultiHead (model/heads.py:13-60) and
    MLP (model/blocks.py:157-189).
Same class names, constructor arguments, sub-module names and construction ORDER (state_dicts are interchangeable, and a model
built under the same torch.manual_seed has the same initial parameters as the reference's), same `forward(inputs) -> (logits,
stage_list)` contract, so the reference's tool/train.py can use it as `model` / `criterion`.

What changes is the neighbourhood work: every block is the fused mirror of blocks.py, and one forward + criterion runs inside
`pointops.neighbor_cache()` when `forward_and_loss` is used — the 5 self-KNNs (one per stage, shared by all encoder and decoder
blocks of the stage), 4 down-sampling KNNs, 4+4 interpolation KNNs and the CBL head's 5+4 searches are each computed once
(57+ launches in the reference).
"""
import torch
import torch.nn as nn

from . import dense, pointops
from .blocks import PointTransformerBlock, TransitionDown, TransitionUp
from .heads import ContrastHead, parse_stage


class Config(dict):
    """attribute-style dict like the reference's util.config.CfgNode (config.py:9-34): `cfg.key`, `'key' in cfg`, nested dicts"""

    def __init__(self, init=None, **kw):
        super().__init__()
        for k, v in dict(init or {}, **kw).items():
            self[k] = Config(v) if type(v) is dict else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def get_ftype(ftype):
    """model/utils.py:59-69"""
    if ftype in ["out", "fout", "f_out", "latent", "logits", "probs"]:
        return ("f_out" if ftype in ["out", "fout"] else ftype), "p_out"
    if ftype in ["sample", "fsample", "f_sample"]:
        return ("f_sample" if ftype in ["sample", "fsample"] else ftype), "p_sample"
    raise KeyError(f"not supported ftype = {ftype}")


def fetch_pxo(stage_n, stage_i, stage_list, ftype):
    stage = stage_list[stage_n][stage_i]
    return stage["p_out"], stage[ftype], stage["offset"]


class MLP(nn.Module):
    """f_out -> latent / logits (blocks.py:157-189; the `*_ops` string-configured variants are not part of the shipped configs)"""
    fkey_to_dims = None

    def __init__(self, fdim, head_cfg, config, fkey, drop=None):
        super().__init__()
        fkey = get_ftype(fkey)[0]
        valid_fkey = {"latent": config.base_fdim, "logits": config.num_classes}
        assert fkey in valid_fkey
        if MLP.fkey_to_dims is None:
            MLP.fkey_to_dims = valid_fkey
        for key in ("latent_ops", "logits_ops"):
            if key in head_cfg and head_cfg[key]:
                raise NotImplementedError(f"{key}: string-configured MLPs (blocks.py:191-240) are not mirrored")
        d_out = valid_fkey["latent"]
        infer_list = [nn.Linear(fdim, d_out), nn.BatchNorm1d(d_out), nn.ReLU(inplace=True)]
        if fkey == "logits":
            infer_list += [nn.Linear(d_out, valid_fkey["logits"])]
        self.infer = nn.Sequential(*infer_list)

    def forward(self, stage, k):
        return dense.sequential(self.infer, stage[k])


class MultiHead(nn.Module):
    """heads.py:13-60: per-stage MLP to the latent, nearest-neighbour upsampling to stage 0, concat, classifier"""

    def __init__(self, fdims, head_cfg, config):
        super().__init__()
        self.head_cfg = head_cfg
        self.ftype = get_ftype(head_cfg.ftype)[0]
        infer_list, ni_list = nn.ModuleList(), []
        for n, i in parse_stage(head_cfg.stage, config.num_layers):
            infer_list.append(MLP(fdims[i], head_cfg, config, self.ftype))
            ni_list.append((n, i))
        self.infer_list, self.ni_list = infer_list, ni_list
        if not head_cfg.combine.startswith("concat"):
            raise ValueError(f"not supported {head_cfg.combine}")
        fdim = MLP.fkey_to_dims[head_cfg.ftype] * len(ni_list)
        k = config.num_classes
        if head_cfg.combine.endswith("mlp"):
            d = config.base_fdim
            self.cls = nn.Sequential(nn.Linear(fdim, d), nn.BatchNorm1d(d), nn.ReLU(inplace=True), nn.Linear(d, k))
        else:
            self.cls = nn.Linear(fdim, k)

    def upsample(self, stage_n, stage_i, stage_list):
        p, x, o = fetch_pxo(stage_n, stage_i, stage_list, self.ftype)
        if stage_i == 0:
            return x
        p0, _, o0 = fetch_pxo("up", 0, stage_list, self.ftype)
        return pointops.interpolation(p, p0, x.contiguous(), o, o0, k=1)           # :50

    def forward(self, stage_list):
        collect_list = []
        for (n, i), func in zip(self.ni_list, self.infer_list):
            stage_list[n][i][self.ftype] = func(stage_list[n][i], "f_out")         # :56-57
            collect_list.append(self.upsample(n, i, stage_list))
        x = torch.cat(collect_list, 1)
        # the classifier runs over EVERY point: through dense (streaming / split weight-gradient kernels), not the library's single-tile GEMM over 10^5 rows
        return (dense.sequential(self.cls, x) if isinstance(self.cls, nn.Sequential) else dense.apply(self.cls, x)), stage_list


class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index) over (n, k) logits as two launches forward and one backward (cbl_cross_entropy_*); the library's nll_loss reduction
    is one workgroup over every point, each way"""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        import ctypes
        from . import _lib
        from .neighbor_state import scratch
        n, k = logits.shape
        L = _lib.lib()
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        stats = torch.empty(2, dtype=torch.float32, device=logits.device)
        ws = scratch(_xe_ws, "xe", L.cbl_cross_entropy_workspace_bytes(ctypes.c_longlong(n)), logits.device)
        _lib.check(L.cbl_cross_entropy_forward(ctypes.c_longlong(n), ctypes.c_int(k), _lib.ptr(logits), _lib.ptr(target), ctypes.c_longlong(int(ignore_index)),
                                               _lib.ptr(loss), _lib.ptr(stats), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(logits)), "cbl_cross_entropy_forward")
        ctx.save_for_backward(logits, target, stats)
        ctx.ignore_index = int(ignore_index)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import _lib
        logits, target, stats = ctx.saved_tensors
        n, k = logits.shape
        grad = torch.empty_like(logits)
        g = g.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().cbl_cross_entropy_backward(ctypes.c_longlong(n), ctypes.c_int(k), _lib.ptr(logits), _lib.ptr(target), ctypes.c_longlong(ctx.ignore_index),
                                                         _lib.ptr(stats), _lib.ptr(g), _lib.ptr(grad), _lib.stream_of(logits)), "cbl_cross_entropy_backward")
        return grad, None, None


_xe_ws = {}


def cross_entropy(logits, target, ignore_index=-100):
    """F.cross_entropy(logits, target, ignore_index=ignore_index) (mean over the points that count); the fused kernels for (n, k <= 64) float32 logits and int64
    targets on the GPU, the library otherwise"""
    if (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and 1 <= logits.shape[1] <= 64 and target.dtype == torch.int64
            and target.shape == logits.shape[:1] and target.device == logits.device):
        return _CrossEntropy.apply(logits.contiguous(), target.contiguous(), ignore_index)
    return nn.functional.cross_entropy(logits, target, ignore_index=ignore_index)


class Loss(nn.Module):
    """pointtransformer_seg.py:15-25: cross entropy + the CBL losses, stacked (1 + stages,)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.contrast_head = ContrastHead(config.contrast, config) if "contrast" in config else None
        self.xen = nn.CrossEntropyLoss(ignore_index=config.ignore_label)

    def forward(self, output, target, stage_list):
        loss_list = [cross_entropy(output, target, self.xen.ignore_index)]          # self.xen: the reference's module, kept for its attributes / state_dict shape
        if self.contrast_head is not None:
            loss_list += self.contrast_head(output, target, stage_list)
        return torch.stack(loss_list)


class PointTransformerSeg(nn.Module):
    """U-shaped network: 5 encoder stages (stage s = TransitionDown with stride[s] followed by blocks[s]-1 attention blocks) and 5 decoder
    stages (TransitionUp followed by one attention block).  Sub-module names (`enc1`..`enc5`, `dec5`..`dec1`, `head` | `cls`) and their
    creation order follow pointtransformer_seg.py:27-70 so that parameters — by name and by seeded initialisation — are the reference's."""
    NUM_STAGES = 5
    STRIDE = [1, 4, 4, 4, 4]
    NSAMPLE = [8, 16, 16, 16, 16]                                   # neighbours of the attention / down-sampling blocks (:44)

    def __init__(self, block, blocks, c=6, k=13, config=None):
        super().__init__()
        config = config if config is not None else Config()
        for key, default in (("planes", [32, 64, 128, 256, 512]), ("share_planes", 8), ("stride", self.STRIDE), ("nsample", self.NSAMPLE)):
            if key not in config:
                config[key] = default                               # note: a CBL config carries its own (larger) `nsample` for the head only
        self.c, self.config = c, config
        planes, share = config.planes, config.share_planes
        width = c                                                   # running feature width while stacking
        for s in range(self.NUM_STAGES):                            # encoders, shallow to deep
            out_w = planes[s] * block.expansion
            layers = [TransitionDown(width, out_w, self.STRIDE[s], self.NSAMPLE[s])]
            layers += [block(out_w, out_w, share, nsample=self.NSAMPLE[s]) for _ in range(blocks[s] - 1)]
            setattr(self, f"enc{s + 1}", nn.Sequential(*layers))
            width = out_w
        for s in reversed(range(self.NUM_STAGES)):                  # decoders, deep to shallow; the deepest one has no coarser input
            out_w = planes[s] * block.expansion
            up = TransitionUp(width, None if s == self.NUM_STAGES - 1 else out_w)
            setattr(self, f"dec{s + 1}", nn.Sequential(up, block(out_w, out_w, share, nsample=self.NSAMPLE[s])))
            width = out_w
        self.in_planes = width
        config.num_layers, config.num_classes = self.NUM_STAGES, k
        self.head = self.cls = None
        if "multi" in config:
            self.head = MultiHead(planes, config.multi, config)
        else:
            self.cls = nn.Sequential(nn.Linear(planes[0], planes[0]), nn.BatchNorm1d(planes[0]), nn.ReLU(inplace=True), nn.Linear(planes[0], k))

    def forward(self, inputs):
        p, x, o = inputs["points"], inputs["features"], inputs["offset"]
        if self.c == 3:
            x = p
        elif self.c == 6:
            x = torch.cat((p, x), 1)
        elif self.c == 7:
            x = torch.cat((torch.ones_like(p[..., :1]), p, x), 1)
        else:
            raise ValueError(f"in_feature_dims c={self.c}")
        pxo = [p, x, o]
        enc = []
        for s in range(self.NUM_STAGES):                            # :97-101
            pxo = getattr(self, f"enc{s + 1}")(pxo)
            enc.append(pxo)
        stage_list = {"inputs": inputs, "down": [{"p_out": q[0], "f_out": q[1], "offset": q[2]} for q in enc]}
        feats = [None] * self.NUM_STAGES
        for s in reversed(range(self.NUM_STAGES)):                  # :113-117
            dec = getattr(self, f"dec{s + 1}")
            ps, xs, os_ = enc[s]
            if s == self.NUM_STAGES - 1:
                fused = dec[0]([ps, xs, os_])                       # per-cloud mean context instead of an upsampled coarser stage
            else:
                fused = dec[0]([ps, xs, os_], [enc[s + 1][0], feats[s + 1], enc[s + 1][2]])
            feats[s] = dec[1:]([ps, fused, os_])[1]
        stage_list["up"] = [{"p_out": enc[s][0], "f_out": feats[s], "offset": enc[s][2]} for s in range(self.NUM_STAGES)]
        if self.head is not None:
            logits, stage_list = self.head(stage_list)
        else:
            logits = dense.sequential(self.cls, feats[0])
        return logits, stage_list


def pointtransformer_seg_repro(**kwargs):
    """pointtransformer_seg.py:146-150"""
    return PointTransformerSeg(PointTransformerBlock, [2, 3, 4, 6, 3], **kwargs)


def forward_and_loss(model, criterion, inputs, target, geometry=None):
    """one training-step forward: network + criterion under ONE neighbour cache -> (logits, stage_list, loss vector, cache).
    `geometry`: a cache being filled by `geometry.prefetch(inputs['points'], inputs['offset'], ...)` on a side stream."""
    with (geometry if geometry is not None else pointops.neighbor_cache()) as nc:
        output, stage_list = model(inputs)
        loss = criterion(output, target, stage_list)
    return output, stage_list, loss, nc


def prefetch_geometry(model, inputs, criterion=None):
    """`geometry.prefetch` with this model's / criterion's settings"""
    from . import geometry
    cfg = model.config
    cbl = criterion is not None and getattr(criterion, "contrast_head", None) is not None
    return geometry.prefetch(inputs["points"], inputs["offset"], stride=model.STRIDE, nsample=model.NSAMPLE,
                             cbl_nsample=cfg.nsample if cbl else None, nstride=cfg.nstride if cbl else None, multi_head=model.head is not None)


class GraphedTrainStep:
    """forward + criterion + backward + optimizer step captured once in a hipGraph and replayed per batch — the network is ~1500 kernel
    launches per step (round 5; ~2700 in round 2), which bounds an eagerly issued step by the host.  depth + 1 buffer sets (inputs, geometry, graph)
    rotate: while one set replays, the geometry of the next `depth` batches (furthest point sampling: one workgroup on one CU, ~0.9 us per sample
    whatever the cloud) is refreshed into the other sets on `depth` side streams.  depth = 2 because one batch's geometry (a serial chain: 11 ms per
    40960-point scene in round 5, 17 ms when this was written) takes as long as the step itself (10.7 ms): with a single batch in flight the step
    waits for it (round 5, `--depth 1`: 13.3 ms per step, 3.7 ms of it idle); two chains side by side deliver a batch every 5.5 ms.  All batches must
    have the first batch's shapes (fixed points per scene, as the reference's voxel_max crop gives).

        step = GraphedTrainStep(model, criterion, optimizer, first_inputs, first_target)
        step.stage(batch0); step.stage(batch1)                  # `depth` batches ahead
        for t in ...: loss, logits = step.run(); step.stage(batch[t + 2])"""

    def __init__(self, model, criterion, optimizer, inputs, target, warmup=3, depth=2, reducer=None):
        """reducer: a distributed.GradientReducer over the model's parameters (data-parallel runs).  The captured graph then ends behind the
        backward pass (gradients accumulate into the reducer's flat buffer, zeroed at the top of the graph); run() issues the bucketed all-reduce
        behind the replay and the optimizer step behind that, eagerly."""
        from . import geometry
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        # a distributed.PackedGradientReducer: gradients are packed into the flat buffer INSIDE the graph, the optimizer is its FlatState's one-tensor twin
        self.packed = getattr(reducer, "state", None)
        if self.packed is None:
            from . import pt_layer
            pt_layer.adjoin_qkv(model)                               # q / k / v weights back to back: one batched product per wide layer without a copy (a flat state does this itself)
        self.sync_buffers = True                                      # rank 0's buffers before every replay (DDP's broadcast_buffers)
        self.events = None                                            # profile(): per-step (start, replayed, reduced, stepped) events
        plan = dict(stride=model.STRIDE, nsample=model.NSAMPLE, multi_head=model.head is not None)
        if getattr(criterion, "contrast_head", None) is not None:
            plan.update(cbl_nsample=model.config.nsample, nstride=model.config.nstride)
        dev = inputs["points"].device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                               # eager warm-up on a side stream (workspaces, momentum buffers, autotune)
            for _ in range(warmup):
                if reducer is None:
                    optimizer.zero_grad(set_to_none=True)
                else:
                    reducer.zero_grad()
                _, _, loss, _ = forward_and_loss(model, criterion, inputs, target)
                loss.sum().backward()
                if reducer is not None:
                    reducer.finish()
                if self.packed is not None:
                    self.packed.step()
                else:
                    optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.depth = min(3, max(1, int(depth)))
        from . import hotpath
        # streams with hardware queues of their own, also beside the stream the step replays on (two fresh streams can share a queue, or the step's)
        from . import geometry
        # EVERY geometry stream is probed against the step's stream: the sampler is one workgroup for ~10 ms, and a geometry stream that shares a hardware
        # queue with the replaying stream puts the whole step behind it; which streams share a queue depends on what the process created before
        import os
        if os.environ.get("CBL_GEO_STREAMS") == "unprobed_first":     # round 4's choice, kept for the A/B of tools/gpu_r05_call2.sh
            first = geometry.side_stream(dev)
            self.geo_streams = [first] + (hotpath.concurrent_streams(self.depth - 1, beside=[torch.cuda.current_stream(dev), first]) if self.depth > 1 else [])
        else:
            self.geo_streams = hotpath.concurrent_streams(self.depth, beside=[torch.cuda.current_stream(dev)])
        self.sets = []
        for _ in range(self.depth + 1):
            st_in = {k: v.clone() for k, v in inputs.items()}
            st_tg = target.clone()
            geom = geometry.StaticGeometry(st_in["points"], st_in["offset"], **plan)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog THREAD polls the events of finished collectives; under the default (global) capture mode any such
            # call from another thread while this one captures is an error that takes the process down ("operation not permitted when stream is
            # capturing", seen on the first run that happened to capture while the watchdog still held work): thread-local mode confines the check to
            # the capturing thread
            import torch.distributed as _dist
            mode = dict(capture_error_mode="thread_local") if (_dist.is_available() and _dist.is_initialized()) else {}
            if reducer is None:
                optimizer.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph, **mode):
                    out, _, loss, _ = forward_and_loss(model, criterion, st_in, st_tg, geometry=geom)
                    loss.sum().backward()
                    optimizer.step()
            else:
                for h in reducer.handles:                           # no collective inside a capture: the buckets go out behind the replay (reducer.rehook() for eager use afterwards)
                    h.remove()
                reducer.handles = []
                with torch.cuda.graph(graph, **mode):
                    reducer.zero_grad()
                    out, _, loss, _ = forward_and_loss(model, criterion, st_in, st_tg, geometry=geom)
                    loss.sum().backward()
                    if self.packed is not None:
                        self.packed.pack()                           # a handful of multi-tensor copies: every gradient into its slice of the flat buffer
            self.sets.append(dict(inputs=st_in, target=st_tg, geom=geom, graph=graph, loss=loss, logits=out))
        self.run_turn = self.stage_turn = self.staged = 0

    def stage(self, inputs, target, ready=None):
        """copy the NEXT batch into the idle buffer set and start its geometry — all on the side stream, behind the last replay that read
        this buffer set and behind `ready`: an event recorded by the caller behind whatever produces `inputs` / `target` (a non-blocking H2D
        copy, GPU augmentation).  Without `ready` the batch must already be complete (the side stream deliberately does NOT wait for the
        caller's current stream: the previous step's replay is queued there, and waiting for it would put the geometry behind the step it is
        meant to run beside)."""
        from . import geometry
        assert self.staged < len(self.sets), "every buffer set holds a staged batch: run() first"
        s = self.sets[self.stage_turn]
        dev = s["target"].device
        side = self.geo_streams[self.stage_turn % len(self.geo_streams)]
        if s.get("done") is not None:
            side.wait_event(s["done"])
        if ready is not None:
            side.wait_event(ready)
        s["geom"].check_offset(inputs.get("offset"))                 # a host-side batch with other cloud boundaries than the captured one: refuse, not a silently wrong geometry
        with torch.cuda.stream(side):
            for k, v in inputs.items():
                s["inputs"][k].copy_(v, non_blocking=True)
                if v.is_cuda:                                        # a pinned host batch has no stream to be kept alive for (the caller keeps it until `ready`)
                    v.record_stream(side)
            s["target"].copy_(target, non_blocking=True)
            if target.is_cuda:
                target.record_stream(side)
        s["geom"].refresh(side)
        self.stage_turn = (self.stage_turn + 1) % len(self.sets)
        self.staged += 1

    def run(self):
        """replay the staged batch -> (loss vector, logits) living in static buffers (valid until this set is replayed again)"""
        assert self.staged > 0, "stage(inputs, target) first"
        s = self.sets[self.run_turn]
        cur = torch.cuda.current_stream(s["target"].device)
        cur.wait_event(s["geom"].ready)
        ev = None
        if self.events is not None:
            import time
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            host = [time.perf_counter()]                              # host clock at the same points: what the issuing thread itself spends in each segment
            ev[0].record(cur)
        if self.reducer is not None and self.sync_buffers:
            # DDP's broadcast_buffers (train.py:181-189): rank 0's running statistics on every rank before the forward that updates them — the fused
            # attention layers update them INSIDE the replayed graph, so without this the ranks' statistics drift apart (one flat collective when packed)
            if self.packed is not None:
                self.packed.broadcast_buffers()
            elif self.reducer.world > 1:
                from . import distributed as D
                D.broadcast_buffers([m for m in (self.model, self.criterion) if isinstance(m, nn.Module)])
        if ev:
            host.append(time.perf_counter())
        s["graph"].replay()
        if ev:
            ev[1].record(cur); host.append(time.perf_counter())
        if self.reducer is not None:
            self.reducer.reduce_all()                               # every bucket, in order, behind the replay; averaged on this stream
            if ev:
                ev[2].record(cur); host.append(time.perf_counter())
            if self.packed is not None:
                self.packed.step()                                  # one fused kernel over the flat parameter buffer
            else:
                self.optimizer.step()
        elif ev:
            ev[2].record(cur); host.append(time.perf_counter())
        if ev:
            ev[3].record(cur); host.append(time.perf_counter())
            self.events.append(ev + [host])
        s["done"] = torch.cuda.Event()
        s["done"].record(cur)
        self.run_turn = (self.run_turn + 1) % len(self.sets)
        self.staged -= 1
        return s["loss"], s["logits"]

    def profile(self, on=True):
        """record four events per run() from now on: before the replay, behind it, behind the gradient all-reduce, behind the optimizer step"""
        self.events = [] if on else None

    def profile_summary(self):
        """-> mean milliseconds per step of the three segments (synchronises); {} when nothing was recorded"""
        if not self.events:
            return {}
        torch.cuda.synchronize()
        n = len(self.events)
        seg = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in self.events) / n
        hseg = lambda a, b: sum(e[4][b] - e[4][a] for e in self.events) / n * 1e3
        gap = sum(a[3].elapsed_time(b[0]) for a, b in zip(self.events[:-1], self.events[1:])) / max(n - 1, 1)
        return {"steps": n, "replay_ms": seg(0, 1), "allreduce_ms": seg(1, 2), "optimizer_ms": seg(2, 3), "stream_idle_between_steps_ms": gap,
                "host_ms": {"buffers": hseg(0, 1), "replay_launch": hseg(1, 2), "allreduce_issue": hseg(2, 3), "optimizer_issue": hseg(3, 4),
                            "between_runs": sum(b[4][0] - a[4][4] for a, b in zip(self.events[:-1], self.events[1:])) / max(n - 1, 1) * 1e3}}
