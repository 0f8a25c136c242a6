#!/usr/bin/env python3
"""C4 / C3-shaped training step: forward + CE and CBL losses + backward + SGD of Point Transformer + CBL (7.8 M parameters, shipped config) on
synthetic S-room scenes, on 1..N GPUs of one node.

    python tools/bench_model.py [--gpus N] [--n 40960] [--scenes 1] [--steps 10] [--graph]      -> one JSON line from rank 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_model.py --gpus N ...

N > 1: one process per GPU (re-executed under torch.distributed.run when no torchrun environment is present), the scenes of the job dealt
round-robin over the ranks (DistributedSampler's role, /root/reference/pytorch/tool/train.py:238; `--scenes` = scenes per rank per step, so the
job's batch grows with N: weak scaling), rank 0's weights broadcast, and the REAL gradients averaged by a bucketed RCCL all-reduce started from
autograd hooks while the backward pass is still running (contrastboundary_amd/distributed.GradientReducer: what DDP does at train.py:181-185).
The timed region is bracketed by barrier + synchronize; rank 0 reports the max over ranks and the whole-job points/s.  `--single-rank-group`
runs the N = 1 step through a one-rank RCCL group and the reducer (exercises the collective path on a 1-GPU box; not a scaling number).
`--host-dry-run`: the launcher / sharding / reducer / timing logic on CPU over gloo with a small dense model (tests; not a measurement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--n", type=int, default=40960); ap.add_argument("--scenes", type=int, default=1, help="scenes per rank per step")
    ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--blas", default="cublas", help="torch.backends.cuda.preferred_blas_library: 'cublas' = rocBLAS (default here: 2-20x faster than hipBLASLt on this network's small weight-gradient GEMMs), 'cublaslt' = torch's default")
    ap.add_argument("--graph", action="store_true", help="the whole training step as a replayed hipGraph, geometry double-buffered (implies prefetch)")
    ap.add_argument("--depth", type=int, default=2, help="--graph: batches whose geometry is in flight ahead of the running step, 1..3 (buffer sets = depth + 1)")
    ap.add_argument("--foreach-sgd", action="store_true", help="torch.optim.SGD's default foreach implementation instead of fused=True (the same update in ~3 kernels instead of ~32)")
    ap.add_argument("--prefetch", action="store_true", help="geometry (FPS + every neighbour search) of the NEXT step on a side stream, one step ahead")
    ap.add_argument("--bucket-mb", type=float, default=8.0, help="gradient bucket size of the all-reduce (xGMI rings want few large messages)")
    ap.add_argument("--single-rank-group", action="store_true", help="N = 1 through a one-rank RCCL group and the gradient reducer (the all-reduce IS issued)")
    ap.add_argument("--hook-reducer", action="store_true", help="data-parallel runs: round 4's layout (every .grad a view of the flat buffer, accumulated in place, per-tensor optimizer) "
                                                               "instead of the flat state (packed gradients, one fused optimizer kernel, one buffer broadcast)")
    ap.add_argument("--no-buffer-broadcast", action="store_true", help="data-parallel graph runs: skip the per-step broadcast of rank 0's buffers (bisecting the wrapper's cost)")
    ap.add_argument("--host-dry-run", action="store_true")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def spawn(a, argv):
    if not a.host_dry_run and torch.cuda.device_count() < a.gpus:
        sys.stderr.write("bench_model.py: --gpus %d but only %d device(s) visible\n" % (a.gpus, torch.cuda.device_count()))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def timed(step, a, sync, D):
    for _ in range(a.warmup):
        step()
    sync(); D.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    sync()
    mine = time.perf_counter() - t0                                  # this rank's own steps, before the closing barrier
    D.barrier()
    total = time.perf_counter() - t0
    ranks, backend = D.group_ranks()
    timed.ranks = {"rccl_ranks": ranks, "backend": backend, "ms_per_step_min_rank": D.reduce_scalar(mine, "min") / a.steps * 1e3,
                   "ms_per_step_max_rank": D.reduce_scalar(mine, "max") / a.steps * 1e3}
    return D.reduce_scalar(total, "max"), out


def host_dry_run(a, D, world, rank):
    """CPU / gloo: scenes dealt per rank, a small dense model through DataParallelTrainer, max-over-ranks timing, one JSON line"""
    import torch.distributed as dist
    from contrastboundary_amd import train_step
    D.init("gloo")
    assert (dist.get_world_size() if dist.is_initialized() else 1) == a.gpus
    torch.manual_seed(1234 + rank)                                   # different initial weights per rank: the broadcast must equalise them
    model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(), torch.nn.Linear(32, 13))
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss()
    tr = train_step.DataParallelTrainer(model, crit, opt, bucket_bytes=2048, forward_loss=lambda m, c, x, t: c(m(x), t).reshape(1))
    mine = D.shard_scenes(a.scenes * world, rank, world)
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(a.scenes * world, 64, 6, generator=g); ts = torch.randint(0, 13, (a.scenes * world, 64), generator=g)
    x, t = torch.cat([xs[i] for i in mine]), torch.cat([ts[i] for i in mine])
    elapsed, loss = timed(lambda: tr.step(x, t), a, lambda: None, D)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    lo, hi = flat.clone(), flat.clone()
    if dist.is_initialized():
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"workload": "host dry run (NOT a measurement)", "n_gpus": world, "ranks": timed.ranks, "scenes_of_rank0": mine, "ms_per_step": elapsed / a.steps * 1e3,
                          "replicas_identical": bool(torch.equal(lo, hi)), "grad_allreduce": tr.describe(), "loss": float(loss.sum())}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def run(a, D, world, rank, local):
    from contrastboundary_amd import neighbor_state, pointtransformer_seg as M, synthetic as S, train_step
    torch.cuda.set_device(local)
    dist_on = world > 1 or a.single_rank_group
    if world > 1:
        D.init("nccl")
    elif a.single_rank_group:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == a.gpus
    torch.backends.cuda.preferred_blas_library(a.blas)
    cfg = M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "voxel_size": 0.04,
                    "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2", "temperature": 1, "weight": "w.1"},
                    "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})
    torch.manual_seed(0)
    model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg).cuda().train()
    crit = M.Loss(cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=not a.foreach_sgd)
    mine = D.shard_scenes(a.scenes * world, rank, world)             # this rank's scenes of the job's batch
    xs, ls = zip(*[S.s_room(a.n, seed=i) for i in mine])
    inputs = {"points": torch.from_numpy(np.concatenate(xs)).cuda(), "features": torch.rand(a.n * len(mine), 3, device="cuda"),
              "offset": torch.tensor(np.cumsum([a.n] * len(mine)), dtype=torch.int32, device="cuda")}
    target = torch.from_numpy(np.concatenate(ls)).cuda()
    trainer = train_step.DataParallelTrainer(model, crit, opt, bucket_bytes=int(a.bucket_mb * (1 << 20)), flat=not a.hook_reducer) if dist_on else None
    nc_last = [None]
    geom_next = [M.prefetch_geometry(model, inputs, crit) if a.prefetch else None]

    if a.graph:
        gstep = M.GraphedTrainStep(model, crit, opt, inputs, target, depth=a.depth, reducer=trainer.reducer if trainer else None)
        for _ in range(gstep.depth):
            gstep.stage(inputs, target)
        gstep.profile(True)                                          # four event records per step: replay / all-reduce / optimizer segments on the step's stream
        gstep.sync_buffers = not a.no_buffer_broadcast
        stage_host = []

        def step():                                                  # replay batch t while batches t+1 .. t+depth are staged (here: the same scenes again)
            loss, _ = gstep.run()
            t0 = time.perf_counter()
            gstep.stage(inputs, target)
            stage_host.append(time.perf_counter() - t0)
            return loss

    elif trainer is not None:
        def fwd(model_, crit_, inputs_, target_):
            geom = geom_next[0]
            if a.prefetch:
                geom_next[0] = M.prefetch_geometry(model_, inputs_, crit_)
            _, _, loss, nc_last[0] = M.forward_and_loss(model_, crit_, inputs_, target_, geometry=geom)
            return loss
        trainer.forward_loss = fwd

        def step():
            return trainer.step(inputs, target)
    else:
        def step():
            opt.zero_grad(set_to_none=True)
            if a.no_cache:
                out, sl = model(inputs); loss = crit(out, target, sl)
            else:
                geom = geom_next[0]
                if a.prefetch:
                    geom_next[0] = M.prefetch_geometry(model, inputs, crit)      # the data loader's next batch (here: the same scene again)
                out, sl, loss, nc_last[0] = M.forward_and_loss(model, crit, inputs, target, geometry=geom)
            loss.sum().backward()
            neighbor_state.release_unowned_transposes()              # tables the backward built outside the forward's neighbour cache
            opt.step()
            return loss

    elapsed, loss = timed(step, a, torch.cuda.synchronize, D)
    dt = elapsed / a.steps
    segments = None
    if a.graph:
        gstep.events = gstep.events[-a.steps:]                       # the timed steps only
        segments = gstep.profile_summary()
        segments["host_ms"]["stage_next_batch"] = sum(stage_host[-a.steps:]) / a.steps * 1e3
    nc = nc_last[0]
    out = {"workload": f"PointTransformerSeg+CBL train step, {a.scenes} x S-room({a.n}) per rank", "n_gpus": world, "ranks": timed.ranks, "ms_per_step": dt * 1e3,
           "points_per_s": a.n * a.scenes * world / dt, "scaling": "weak", "scenes_of_rank0": mine,
           "knn_requests": None if nc is None else nc.hits + nc.misses, "knn_searches": None if nc is None else nc.misses,
           "geometry_prefetch": bool(a.prefetch or a.graph), "hipgraph": bool(a.graph), "blas": a.blas, "segments_ms": segments,
           "grad_allreduce": None if trainer is None else dict(trainer.describe(), mode="behind the graph replay, in bucket order" if a.graph else "from autograd hooks, beside the backward pass"),
           "loss": [round(float(v), 5) for v in loss.detach().cpu()]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(a, argv)
    from contrastboundary_amd import distributed as D
    world, rank, local = D.env_world()
    if world != a.gpus:
        sys.stderr.write("bench_model.py: WORLD_SIZE=%d but --gpus %d\n" % (world, a.gpus))
        return 2
    if a.host_dry_run:
        return host_dry_run(a, D, world, rank)
    if not torch.cuda.is_available():
        sys.stderr.write("bench_model.py: no GPU visible (the network's kernels have no CPU fallback)\n")
        return 2
    return run(a, D, world, rank, local)


if __name__ == "__main__":
    sys.exit(main())
