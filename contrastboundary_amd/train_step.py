"""One data-parallel training step of the reference's network (BASELINE configs C3 / C4 "8 GPUs"): one process per GPU, every
rank its own scenes (DistributedSampler's role, /root/reference/pytorch/tool/train.py:238), the model replicated, gradients averaged with a
bucketed all-reduce that runs beside the backward pass (what DistributedDataParallel does there, train.py:181-185), then the optimizer step.

    trainer = DataParallelTrainer(model, criterion, optimizer)           # after dist.init_process_group; world 1 works too (no collective)
    loss = trainer.step(inputs, target)

`forward_loss(model, criterion, inputs, target) -> loss tensor` is the network-specific part; the default is the Point Transformer + CBL
step (pointtransformer_seg.forward_and_loss).  With graph=True the forward + backward is a replayed hipGraph (pointtransformer_seg.GraphedTrainStep
with a reducer): no autograd hook fires during a replay, so the buckets are issued right behind it, in order.
"""
import torch

from . import distributed as D


def _pt_forward_loss(model, criterion, inputs, target):
    from . import pointtransformer_seg as M
    return M.forward_and_loss(model, criterion, inputs, target)[2]


class DataParallelTrainer:
    def __init__(self, model, criterion, optimizer, bucket_bytes=8 << 20, forward_loss=None, broadcast=True, sync_buffers=True, flat=False):
        """The criterion is a module too: the reference wraps it in DistributedDataParallel whenever it has trainable parameters (the CBL head's
        `project` MLP, train.py:189), so its parameters are broadcast and averaged with the model's, and — sync_buffers, DDP's broadcast_buffers
        default — rank 0's buffers (BatchNorm running statistics) are re-broadcast at the start of every step, model's and criterion's alike."""
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.forward_loss = forward_loss or _pt_forward_loss
        self.modules = [model] + ([criterion] if isinstance(criterion, torch.nn.Module) else [])
        if broadcast:
            for m in self.modules:
                D.broadcast_parameters(m)                           # rank 0's initial weights (and buffers) everywhere
        params = [p for m in self.modules for p in m.parameters()]
        self.state = None
        if flat:
            # parameters, gradients and buffers as views of flat buffers (distributed.FlatState): the step then costs what a single-GPU step costs —
            # gradients packed by a few multi-tensor copies, bucketed all-reduce over slices, ONE fused optimizer kernel, ONE buffer broadcast
            self.state = D.FlatState(self.modules, optimizer)
            self.state.flat_optimizer(optimizer)
            self.reducer = D.PackedGradientReducer(self.state, bucket_bytes=bucket_bytes)
        else:
            self.reducer = D.GradientReducer(params, bucket_bytes=bucket_bytes)
        self.world = self.reducer.world
        self.sync_buffers = sync_buffers

    def step(self, inputs, target):
        """buffers from rank 0 -> zero -> forward -> backward (buckets all-reduced as they complete) -> wait + average -> optimizer step; returns the loss vector"""
        if self.sync_buffers:
            if self.state is not None:
                self.state.broadcast_buffers()
            else:
                D.broadcast_buffers(self.modules)
        self.reducer.zero_grad()
        loss = self.forward_loss(self.model, self.criterion, inputs, target)
        loss.sum().backward()
        from . import neighbor_state
        neighbor_state.release_unowned_transposes()                 # tables the backward built for itself (no neighbour cache on autograd's thread)
        self.reducer.finish()
        if self.state is not None:
            self.state.step()
        else:
            self.optimizer.step()
        return loss.detach()

    def describe(self):
        r = self.reducer
        return {"gradient_bytes": r.flat.numel() * r.flat.element_size(), "buckets": len(r.buckets),
                "bucket_bytes": [(e - s) * r.flat.element_size() for s, e, _ in r.buckets], "ranks": r.world, "collective_issued": bool(r.grouped),
                "layout": "flat parameter / gradient / buffer tensors, gradients packed behind the backward, one fused optimizer kernel" if self.state is not None
                          else "every .grad a view of one flat buffer, accumulated in place by autograd",
                "overlap": "bucket k's all-reduce is started by autograd's post-accumulate hooks when its last gradient is written (reverse parameter order), "
                           "and joined before the optimizer step"}
