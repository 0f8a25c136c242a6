#!/usr/bin/env python3
"""tests/golden/blocks_wide_pytorch.npz: the reference's own PointTransformerLayer (pytorch/model/blocks.py:8-44) at the widths of the deeper
stages, C = 128 / 256 / 512 with share_planes 8 and nsample 16, run on CPU in the build container (same substitutions as
gen_blocks_goldens.py: empty CUDA module, knnquery through the CPU oracle).  The weights are NOT stored: the layer is built under
torch.manual_seed(seed) and our mirror, built under the same seed, has the same initial parameters (same construction order) — the fixture
keeps a per-tensor checksum of the reference's state_dict so the test can assert exactly that before it compares anything.
Stored per case: seed, n, checksums of the seeded inputs x / g, and output (train-mode BatchNorm) and d(sum(out * g))/d(x) of a FLOAT64 pass of the
reference layer (`*64`): the summation-order-free value the 1e-4 parity bound is tested against."""
import os
import sys
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


def knnquery_cpu(nsample, xyz, new_xyz, offset, new_offset):
    if new_xyz is None:
        new_xyz = xyz
    idx, d2 = O.knnquery(int(nsample), xyz.detach().numpy(), new_xyz.detach().numpy(), offset.numpy(), new_offset.numpy())
    return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2))


rp.knnquery = knnquery_cpu
out = {}
for c, n, seed in ((128, 1024, 11), (256, 320, 12), (512, 160, 13)):
    xyz, _ = S.s_room(n, seed=seed)
    p = torch.from_numpy(xyz); o = torch.tensor([n // 3, n], dtype=torch.int32)
    torch.manual_seed(seed)
    layer = rb.PointTransformerLayer(c, c, 8, 16)
    layer.train()
    names = sorted(layer.state_dict().keys())
    sums = np.float64([float(layer.state_dict()[k].double().sum()) for k in names])      # before the forward pass updates the running statistics
    gen = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(n, c, generator=gen).requires_grad_(True)
    g = torch.randn(n, c, generator=gen)
    xg_sums = np.float64([float(x.detach().double().sum()), float(g.double().sum())])    # x / g are not stored: the generator reproduces them
    # float64 pass of the same module and inputs: the value both fp32 implementations round around (the fp32 reference is within ~1e-6 of it)
    layer = layer.double()
    x = x.detach().double().requires_grad_(True)
    y = layer([p.double(), x, o])
    (y * g.double()).sum().backward()
    pre = f"c{c}"
    out[f"{pre}/meta"] = np.int64([c, n, seed])
    out[f"{pre}/p"] = xyz; out[f"{pre}/offset"] = o.numpy()
    out[f"{pre}/xg_sums"] = xg_sums
    out[f"{pre}/out64"] = y.detach().numpy().astype(np.float32); out[f"{pre}/grad_x64"] = x.grad.numpy().astype(np.float32)
    out[f"{pre}/sd_names"] = np.array(names)
    out[f"{pre}/sd_sums"] = sums
np.savez_compressed(os.path.join(HERE, "blocks_wide_pytorch.npz"), **out)
print("ok", len(out), "arrays")
