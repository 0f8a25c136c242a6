#!/usr/bin/env python3
"""tests/golden/blocks_pytorch.npz: the reference's own Point-Transformer blocks (pytorch/model/blocks.py) run on CPU.
Build container only.  As in gen_cbl_goldens.py the CUDA module is replaced by an empty module and the two native ops the blocks
reach (knnquery, furthestsampling) by the CPU oracle; `torch.cuda.FloatTensor/IntTensor` are aliased to their CPU types so that
the reference's pure-torch pointops composites (queryandgroup, interpolation: pointops.py:79-100,164-178) run unmodified.
Stored per case: state_dict, inputs, outputs (train-mode BatchNorm = batch statistics) and d(sum(out * g))/d(x) — from the fp32 modules as the
reference runs them, and (`*64` keys) from the same modules and inputs in float64: the summation-order-free value both fp32 implementations
round around, which is what the 1e-4 parity bound is tested against."""
import copy
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


def fps_cpu(xyz, offset, new_offset):
    idx, _ = O.furthestsampling(xyz.detach().numpy(), offset.numpy(), new_offset.numpy())
    return torch.from_numpy(idx)


rp.knnquery = knnquery_cpu
rp.furthestsampling = fps_cpu

out = {}


class fp64:
    """the reference's composites allocate torch.cuda.FloatTensor results (pointops.py:175): doubles while a float64 pass runs"""
    def __enter__(self):
        torch.cuda.FloatTensor = torch.DoubleTensor
    def __exit__(self, *exc):
        torch.cuda.FloatTensor = torch.FloatTensor


def save_sd(prefix, mod):
    for k, v in mod.state_dict().items():
        out[f"{prefix}/sd/{k}"] = v.numpy()


torch.manual_seed(0)
n, c = 1536, 32
xyz, _ = S.s_room(n, seed=2)
p = torch.from_numpy(xyz); o = torch.tensor([600, 1536], dtype=torch.int32)
x = torch.randn(n, c, requires_grad=True)
g = torch.randn(n, c)

# a4: PointTransformerLayer and the full block
for name, mod in {"layer": rb.PointTransformerLayer(c, c, 8, 16), "block": rb.PointTransformerBlock(c, c, 8, 16)}.items():
    mod.train()
    save_sd(name, mod)
    y = mod([p, x, o])
    y = y[1] if isinstance(y, list) else y
    x.grad = None
    (y * g).sum().backward()
    out[f"{name}/out"] = y.detach().numpy(); out[f"{name}/grad_x"] = x.grad.numpy().copy()
    with fp64():
        m64 = copy.deepcopy(mod).double(); m64.load_state_dict({k: v.double() for k, v in {kk[len(name) + 4:]: torch.from_numpy(out[kk]) for kk in out if kk.startswith(name + "/sd/")}.items()})
        m64.train()
        x64 = x.detach().double().requires_grad_(True)
        y64 = m64([p.double(), x64, o])
        y64 = y64[1] if isinstance(y64, list) else y64
        (y64 * g.double()).sum().backward()
        out[f"{name}/out64"] = y64.detach().numpy().astype(np.float32); out[f"{name}/grad_x64"] = x64.grad.numpy().astype(np.float32)

# a5: TransitionDown stride 4
td = rb.TransitionDown(c, 64, 4, 16); td.train(); save_sd("down", td)
x.grad = None
p2, y2, o2 = td([p, x, o])
g2 = torch.randn_like(y2)
(y2 * g2).sum().backward()
out["down/p"] = p2.numpy(); out["down/out"] = y2.detach().numpy(); out["down/offset"] = o2.numpy(); out["down/g"] = g2.numpy()
out["down/grad_x"] = x.grad.numpy().copy()
with fp64():
    td64 = copy.deepcopy(td).double(); td64.load_state_dict({k[8:]: torch.from_numpy(out[k]).double() for k in out if k.startswith("down/sd/")}); td64.train()
    x64 = x.detach().double().requires_grad_(True)
    _, y64, _ = td64([p.double(), x64, o])
    (y64 * g2.double()).sum().backward()
    out["down/out64"] = y64.detach().numpy().astype(np.float32); out["down/grad_x64"] = x64.grad.numpy().astype(np.float32)

# TransitionUp, both forms
tu = rb.TransitionUp(64, c); tu.train(); save_sd("up", tu)
x2 = y2.detach().clone().requires_grad_(True)
x.grad = None
y = tu([p, x, o], [p2, x2, o2])
(y * g).sum().backward()
out["up/out"] = y.detach().numpy(); out["up/grad_x1"] = x.grad.numpy().copy(); out["up/grad_x2"] = x2.grad.numpy().copy()
th = rb.TransitionUp(64); th.train(); save_sd("uphead", th)
x2.grad = None
y = th([p2, x2, o2])
(y * g2).sum().backward()
out["uphead/out"] = y.detach().numpy(); out["uphead/grad_x"] = x2.grad.numpy().copy()

out["p"] = xyz; out["offset"] = o.numpy(); out["x"] = x.detach().numpy(); out["g"] = g.numpy()
np.savez_compressed(os.path.join(HERE, "blocks_pytorch.npz"), **out)
print("ok", len(out), "arrays")
