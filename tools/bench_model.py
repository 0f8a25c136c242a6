#!/usr/bin/env python3
"""C4: one training-step forward+backward of Point Transformer + CBL (7.8 M parameters, shipped config) on synthetic S-room scenes.
python tools/bench_model.py [--n 40960] [--scenes 1] [--steps 10]   -> one JSON line (ms per step, points/s, KNN cache statistics)"""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import pointtransformer_seg as M, synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40960); ap.add_argument("--scenes", type=int, default=1)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--no-cache", action="store_true")
ap.add_argument("--blas", default="cublas", help="torch.backends.cuda.preferred_blas_library: 'cublas' = rocBLAS (default here: 2-20x faster than hipBLASLt on this network's small weight-gradient GEMMs), 'cublaslt' = torch's default")
ap.add_argument("--graph", action="store_true", help="the whole training step as a replayed hipGraph, geometry double-buffered (implies prefetch)")
ap.add_argument("--depth", type=int, default=2, help="--graph: batches whose geometry is in flight ahead of the running step, 1..3 (buffer sets = depth + 1)")
ap.add_argument("--foreach-sgd", action="store_true", help="torch.optim.SGD's default foreach implementation instead of fused=True (the same update in ~3 kernels instead of ~32)")
ap.add_argument("--prefetch", action="store_true", help="geometry (FPS + every neighbour search) of the NEXT step on a side stream, one step ahead")
a = ap.parse_args()
torch.backends.cuda.preferred_blas_library(a.blas)
cfg = M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "voxel_size": 0.04,
                "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2", "temperature": 1, "weight": "w.1"},
                "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})
torch.manual_seed(0)
model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg).cuda().train()
crit = M.Loss(cfg)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=not a.foreach_sgd)
xs, ls = zip(*[S.s_room(a.n, seed=i) for i in range(a.scenes)])
inputs = {"points": torch.from_numpy(np.concatenate(xs)).cuda(), "features": torch.rand(a.n * a.scenes, 3, device="cuda"),
          "offset": torch.tensor(np.cumsum([a.n] * a.scenes), dtype=torch.int32, device="cuda")}
target = torch.from_numpy(np.concatenate(ls)).cuda()


geom_next = [M.prefetch_geometry(model, inputs, crit) if a.prefetch else None]


def step():
    opt.zero_grad(set_to_none=True)
    if a.no_cache:
        out, sl = model(inputs); loss = crit(out, target, sl); nc = None
    else:
        geom = geom_next[0]
        if a.prefetch:
            geom_next[0] = M.prefetch_geometry(model, inputs, crit)      # the data loader's next batch (here: the same scene again)
        out, sl, loss, nc = M.forward_and_loss(model, crit, inputs, target, geometry=geom)
    loss.sum().backward()
    opt.step()
    return loss, nc


if a.graph:
    gstep = M.GraphedTrainStep(model, crit, opt, inputs, target, depth=a.depth)
    for _ in range(gstep.depth):
        gstep.stage(inputs, target)

    def step():                                                      # replay batch t while batches t+1 .. t+depth are staged (here: the same scene again)
        loss, _ = gstep.run()
        gstep.stage(inputs, target)
        return loss, None

for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, nc = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"workload": f"PointTransformerSeg+CBL train step, {a.scenes} x S-room({a.n})", "ms_per_step": dt * 1e3,
                  "points_per_s": a.n * a.scenes / dt, "knn_requests": None if nc is None else nc.hits + nc.misses,
                  "knn_searches": None if nc is None else nc.misses, "geometry_prefetch": bool(a.prefetch or a.graph), "hipgraph": bool(a.graph), "blas": a.blas, "loss": [round(float(v), 5) for v in loss.detach().cpu()]}))
