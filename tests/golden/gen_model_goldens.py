#!/usr/bin/env python3
"""tests/golden/model_pytorch.npz: the reference's full network + criterion (pytorch/model/pointtransformer_seg.py:
PointTransformerSeg via pointtransformer_seg_repro, MultiHead, Loss with the CBL ContrastHead) run on CPU.  Build container only.

As in gen_blocks_goldens.py / gen_cbl_goldens.py the CUDA module is an empty stand-in, `pointops.knnquery` /
`pointops.furthestsampling` are the CPU oracle (pinned bit-exact to the reference kernels by pointops_*.npz) and the
`torch.cuda.*Tensor` constructors are aliased to CPU so that the reference's pure-torch composites run unmodified.  Config =
the shipped yaml (config/s3dis/origin_multi-Ua-concat-latent_contrast-Ua-softnn-latent-label-l2-w.1.yaml:54-72).

The 7.8 M parameters are NOT stored: the model is built under torch.manual_seed(seed) and the mirror, which constructs the same
modules in the same order, reproduces them from the seed (a parameter checksum is stored to prove it).  Stored per case: inputs,
target, logits, the (1+5,) loss vector, the gradients of loss.sum() w.r.t. two parameter tensors (first and last layer), and the
number of knnquery calls the reference made."""
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
from lib.pointops.functions import pointops as rp            # noqa: E402
from model import pointtransformer_seg as rm                  # noqa: E402
from util.config import CfgNode                               # noqa: E402
from tests import oracle_lib as O                             # noqa: E402
from contrastboundary_amd import synthetic as S               # noqa: E402

calls = {"knn": 0}


def knnquery_cpu(nsample, xyz, new_xyz, offset, new_offset):
    calls["knn"] += 1
    if new_xyz is None:
        new_xyz = xyz
    idx, d2 = O.knnquery(int(nsample), xyz.detach().numpy(), new_xyz.detach().numpy(), offset.numpy(), new_offset.numpy())
    return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2))


def fps_cpu(xyz, offset, new_offset):
    idx, _ = O.furthestsampling(xyz.detach().numpy(), offset.numpy(), new_offset.numpy())
    return torch.from_numpy(idx)


rp.knnquery = knnquery_cpu
rp.furthestsampling = fps_cpu


def shipped_config():
    return CfgNode({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "voxel_size": 0.04,
                    "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2",
                                 "temperature": 1, "weight": "w.1"},
                    "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})


out = {}
for case, (lens, seed) in {"one_cloud_8192": ([8192], 0), "two_clouds_12000": ([7000, 5000], 1)}.items():
    n = sum(lens)
    torch.manual_seed(seed)
    cfg = shipped_config()
    model = rm.pointtransformer_seg_repro(c=6, k=13, config=cfg)
    crit = rm.Loss(cfg)
    model.train()
    xyz, labels = S.s_room(n, seed=seed)
    rng = np.random.default_rng(seed + 100)
    feat = rng.uniform(0, 1, (n, 3)).astype(np.float32)          # colours
    inputs = {"points": torch.from_numpy(xyz), "features": torch.from_numpy(feat), "offset": torch.tensor(np.cumsum(lens), dtype=torch.int32)}
    target = torch.from_numpy(labels)
    calls["knn"] = 0
    logits, stage_list = model(inputs)
    loss = crit(logits, target, stage_list)
    loss.sum().backward()
    sd = model.state_dict()
    out[f"{case}/seed"] = np.int64(seed)
    out[f"{case}/xyz"] = xyz; out[f"{case}/feat"] = feat; out[f"{case}/offset"] = np.cumsum(lens).astype(np.int32); out[f"{case}/target"] = labels
    out[f"{case}/logits"] = logits.detach().numpy(); out[f"{case}/loss"] = loss.detach().numpy()
    out[f"{case}/param_abs_sum"] = np.float64(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k))
    out[f"{case}/grad_first"] = model.enc1[0].linear.weight.grad.numpy().copy()
    out[f"{case}/grad_last"] = model.head.cls.weight.grad.numpy().copy()
    out[f"{case}/state_dict_keys"] = np.array([f"{k}:{tuple(v.shape)}" for k, v in sd.items()])
    out[f"{case}/stage_sizes"] = np.array([st["p_out"].shape[0] for st in stage_list["up"]], np.int64)
    out[f"{case}/ref_knn_calls"] = np.int64(calls["knn"])
    # the same model, inputs and indices in float64 (`*64`): the summation-order-free value the 1e-4 parity bound is tested against
    torch.cuda.FloatTensor = torch.DoubleTensor
    m64 = copy.deepcopy(model).double(); m64.zero_grad(); m64.train()
    # running statistics were advanced by the fp32 pass above: train-mode BatchNorm normalises with batch statistics, so they do not enter the outputs
    in64 = {"points": inputs["points"].double(), "features": inputs["features"].double(), "offset": inputs["offset"]}
    lg64, sl64 = m64(in64)
    ls64 = crit(lg64, target, sl64)
    ls64.sum().backward()
    torch.cuda.FloatTensor = torch.FloatTensor
    out[f"{case}/logits64"] = lg64.detach().numpy().astype(np.float32); out[f"{case}/loss64"] = ls64.detach().numpy().astype(np.float64)
    out[f"{case}/grad_first64"] = m64.enc1[0].linear.weight.grad.numpy().astype(np.float32)
    out[f"{case}/grad_last64"] = m64.head.cls.weight.grad.numpy().astype(np.float32)
    print(case, "fp32 vs fp64 reference: logits", float(np.abs(out[f"{case}/logits"] - out[f"{case}/logits64"]).max()), "loss", np.abs(out[f"{case}/loss"] - out[f"{case}/loss64"]).max())
    print(case, "loss", loss.detach().numpy(), "knn calls", int(out[f"{case}/ref_knn_calls"]), "stages", out[f"{case}/stage_sizes"])

np.savez_compressed(os.path.join(HERE, "model_pytorch.npz"), **out)
print("wrote", os.path.join(HERE, "model_pytorch.npz"), os.path.getsize(os.path.join(HERE, "model_pytorch.npz")) // 1024, "KiB")
