#!/usr/bin/env python3
"""tests/golden/pt_layer_bench_pytorch.npz: the reference's own PointTransformerLayer (pytorch/model/blocks.py:8-44) at the two FULL-RESOLUTION
shapes of the network on the scene bench.py times — (n, K, C) = (40960, 16, 64), BASELINE's synthetic shape, and (40960, 8, 32), the real
first stage — as a FLOAT64 pass on CPU in the build container (same substitutions as gen_blocks_goldens.py: empty CUDA module, knnquery
through the CPU oracle).  This is what holds csrc/pt_layer.hip to the REFERENCE at the bench shape (round 4 compared it there with this
repository's own unfused layer only).

Nothing of size (n, C) is stored whole (2 x 10.5 MB per shape would triple the fixture directory).  Per case:
  * seeds; the state_dict's per-tensor checksums (the mirror built under the same seed must have the same initial parameters — same construction
    order; the BatchNorm affine parameters are then redrawn from a seeded generator on both sides, so that gamma / beta gradients mean something);
    checksums of the seeded inputs x / g (the CPU generator reproduces them on the GPU box);
  * of the output and of d(sum(out * g))/d(x): every `STEP`-th row (rows 0, STEP, 2 STEP, …) as float32, the float64 column sums over ALL rows,
    the float64 sum of squares and the largest magnitude — sampled rows are compared entry by entry, the column sums hold every other row;
  * every parameter gradient whole (they are small), and the three BatchNorms' running statistics after the pass."""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")
sys.path.insert(0, "/root/reference/pytorch")
torch.cuda.FloatTensor = torch.FloatTensor
torch.cuda.IntTensor = torch.IntTensor
from lib.pointops.functions import pointops as rp      # noqa: E402
from model import blocks as rb                           # noqa: E402
from tests import oracle_lib as O                        # noqa: E402
from contrastboundary_amd import synthetic as S          # noqa: E402

STEP = 16


def knnquery_cpu(nsample, xyz, new_xyz, offset, new_offset):
    if new_xyz is None:
        new_xyz = xyz
    idx, d2 = O.knnquery(int(nsample), xyz.detach().numpy(), new_xyz.detach().numpy(), offset.numpy(), new_offset.numpy())
    return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2))


def redraw_bn_affine(layer, seed):
    """BatchNorm gamma in [0.5, 1.5), beta in [-0.3, 0.3) from a CPU generator: the SAME function is in tests/test_gpu_pt_layer.py"""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.rand(m.bias.shape, generator=gen) * 0.6 - 0.3)


def summary(t):
    t = t.detach().double()
    return t[::STEP].numpy().astype(np.float32), t.sum(0).numpy(), np.float64([float((t * t).sum()), float(t.abs().max())])


rp.knnquery = knnquery_cpu
torch.set_num_threads(8)
out = {}
for n, K, C, seed in ((40960, 16, 64, 21), (40960, 8, 32, 22)):
    t0 = time.time()
    xyz, _ = S.s_room(n, seed=0)                                    # the scene bench.py times
    p = torch.from_numpy(xyz); o = torch.tensor([n], dtype=torch.int32)
    torch.manual_seed(seed)
    layer = rb.PointTransformerLayer(C, C, 8, K)
    layer.train()
    names = sorted(layer.state_dict().keys())
    sums = np.float64([float(layer.state_dict()[k].double().sum()) for k in names])      # before the redraw and before the forward pass
    redraw_bn_affine(layer, 2000 + seed)
    gen = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(n, C, generator=gen)
    g = torch.randn(n, C, generator=gen)
    xg_sums = np.float64([float(x.double().sum()), float(g.double().sum())])
    layer = layer.double()
    x = x.double().requires_grad_(True)
    y = layer([p.double(), x, o])
    (y * g.double()).sum().backward()
    pre = f"n{n}_k{K}_c{C}"
    out[f"{pre}/meta"] = np.int64([n, K, C, seed, STEP])
    out[f"{pre}/xg_sums"] = xg_sums
    out[f"{pre}/sd_names"] = np.array(names); out[f"{pre}/sd_sums"] = sums
    out[f"{pre}/out_rows"], out[f"{pre}/out_colsum"], out[f"{pre}/out_norm"] = summary(y)
    out[f"{pre}/gx_rows"], out[f"{pre}/gx_colsum"], out[f"{pre}/gx_norm"] = summary(x.grad)
    pnames = [k for k, _ in layer.named_parameters()]
    out[f"{pre}/param_names"] = np.array(pnames)
    for k, t in layer.named_parameters():
        out[f"{pre}/grad/{k}"] = t.grad.numpy().astype(np.float64)
    for k, t in layer.named_buffers():
        out[f"{pre}/buffer/{k}"] = t.detach().numpy().astype(np.float64)
    print(pre, "done in %.0f s" % (time.time() - t0), flush=True)
np.savez_compressed(os.path.join(HERE, "pt_layer_bench_pytorch.npz"), **out)
print("ok", len(out), "arrays")
