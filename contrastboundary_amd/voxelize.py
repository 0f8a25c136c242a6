"""Host mirror of /root/reference/pytorch/util/voxelize.py (voxelize :38-56, fnv_hash_vec :4-16) and of the nearest-voxel_max
crop of data_prepare (/root/reference/pytorch/util/data_util.py:57-67) — the dataloader stage that runs as numpy on 16 CPU
workers in the reference, here on the GPU so that 8 GPUs are not starved by the host."""
import ctypes

import torch

from . import _lib

_ws = {}


def _workspace(n, device):
    need = _lib.lib().cbl_voxelize_workspace_bytes(ctypes.c_int(n))
    from .neighbor_state import scratch
    return scratch(_ws, "voxelize", need, device, grow=1.25)


def _coord(coord):
    if not (isinstance(coord, torch.Tensor) and coord.is_cuda and coord.dim() == 2 and coord.shape[1] == 3 and coord.is_contiguous()
            and coord.dtype in (torch.float32, torch.float64)):
        raise TypeError("coord: expected a contiguous (n,3) float32/float64 CUDA tensor")
    return coord


def voxelize(coord, voxel_size=0.05, hash_type="fnv", mode=0, rand=None):
    """mode 1 (val): -> (idx_sort (n,) int64, count (v,) int64)         voxelize.py:53-55
    mode 0 (train): -> idx_unique (v,) int64, one point per voxel: idx_sort[start + rand % count] with rand in [0, count.max())
                    (the reference draws rand from np.random, :47-51; pass `rand` (v,) for a reproducible choice)."""
    if hash_type != "fnv":
        raise NotImplementedError("hash_type='ravel' is not used by the reference's pipelines")
    _coord(coord)
    n, dev = coord.shape[0], coord.device
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    idx_sort = torch.empty(n, dtype=torch.int32, device=dev)
    start = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.empty(n, dtype=torch.int32, device=dev)
    nv = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _workspace(n, dev)
    _lib.check(_lib.lib().cbl_voxelize(ctypes.c_int(n), ctypes.c_int(1 if coord.dtype == torch.float64 else 0), _lib.ptr(coord),
                                       ctypes.c_double(float(voxel_size)), _lib.ptr(keys), _lib.ptr(idx_sort), _lib.ptr(start), _lib.ptr(count),
                                       _lib.ptr(nv), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(coord)), "cbl_voxelize")
    v = int(nv.item())
    idx_sort, start, count = idx_sort.long(), start[:v].long(), count[:v].long()
    if mode != 0:
        return idx_sort, count
    if rand is None:
        rand = torch.randint(0, int(count.max().item()), (v,), device=dev)
    return idx_sort[start + rand.to(dev).long() % count]


def crop_nearest(coord, center_idx, voxel_max):
    """indices of the voxel_max points nearest to coord[center_idx] (ascending distance)      data_util.py:62-64"""
    _coord(coord)
    n, dev = coord.shape[0], coord.device
    order = torch.empty(n, dtype=torch.int32, device=dev)
    ws = _workspace(n, dev)
    _lib.check(_lib.lib().cbl_crop_order(ctypes.c_int(n), ctypes.c_int(1 if coord.dtype == torch.float64 else 0), _lib.ptr(coord), ctypes.c_int(int(center_idx)),
                                         _lib.ptr(order), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(coord)), "cbl_crop_order")
    return order[:voxel_max].long()


def test_time_crops(coord, voxel_max, potentials=None):
    """The spatially regular crops of the reference's test loop (tool/test.py:197-215) for a cloud with more than voxel_max points:
    until every point is covered — centre = the point of minimum potential, crop = its voxel_max nearest points, potentials of the crop
    raised by (1 - d2 / max d2)^2.  coord (n,3) CUDA; potentials (n,) float64 (default: rand * 1e-3 like the reference, :198).
    -> list of (voxel_max,) int64 index tensors, ascending distance.  One host synchronisation per crop (the loop's exit test)."""
    _coord(coord)
    n, dev = coord.shape[0], coord.device
    pot = (torch.rand(n, dtype=torch.float64, device=dev) * 1e-3) if potentials is None else potentials.to(device=dev, dtype=torch.float64).clone()
    covered = torch.zeros(n, dtype=torch.bool, device=dev)
    crops, ncov = [], 0
    while ncov != n:
        init = int(torch.argmin(pot).item())                         # first minimum, like np.argmin
        idx = crop_nearest(coord, init, voxel_max)
        d = coord[idx] - coord[init]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        delta = torch.square(1 - dist / dist.max())
        pot[idx] += delta.to(torch.float64)
        covered[idx] = True
        ncov = int(covered.sum().item())
        crops.append(idx)
    return crops


def init_cumulate_dict(n, num_classes, device="cuda", probs_last=False):
    """tool/test.py:313-328: the per-cloud accumulators of the test loop ('probs', optionally 'probs_last')"""
    cum = {"probs": torch.zeros((n, num_classes), dtype=torch.float32, device=device)}
    if probs_last:
        cum["probs_last"] = torch.zeros((n, num_classes), dtype=torch.float32, device=device)
    return cum


def cumulate_probs(cum_dict, pred, inds, smooth=None, pred_type="logits"):
    """tool/test.py:330-352: add the predictions of one batch of crops (pred (m, ncls), inds (m) = the points the rows belong to, the
    concatenation of the crops) into cum_dict['probs'] (and overwrite 'probs_last').  Where crops of the batch overlap the indexed update
    keeps ONE row per point — the last, as the reference's expression does on the CPU (cbl_cumulate_probs)."""
    assert pred_type in ["logits"], f"not support pred_type = {pred_type}"
    if "probs_1st" in cum_dict:
        raise NotImplementedError("probs_1st: the reference raises on this path itself (tool/test.py:340)")
    probs = cum_dict["probs"]
    if not (probs.is_cuda and probs.dtype == torch.float32 and probs.is_contiguous()):
        raise TypeError("cum_dict['probs']: expected a contiguous float32 CUDA tensor")
    n, ncls = probs.shape
    pred = pred.detach().to(torch.float32).contiguous()
    inds = torch.as_tensor(inds, device=probs.device).to(torch.int64).contiguous()
    m = inds.shape[0]
    if pred.shape != (m, ncls):
        raise ValueError(f"pred has shape {tuple(pred.shape)}, expected {(m, ncls)}")
    scratch = torch.empty(max(n, 1), dtype=torch.int32, device=probs.device)
    L = _lib.lib()
    for key, mode in (("probs", 0 if smooth is None else 1), ("probs_last", 2)):
        if key in cum_dict:
            _lib.check(L.cbl_cumulate_probs(ctypes.c_int(n), ctypes.c_int(ncls), ctypes.c_int(m), _lib.ptr(inds), _lib.ptr(pred),
                                            ctypes.c_float(0.0 if smooth is None else float(smooth)), ctypes.c_int(mode), _lib.ptr(cum_dict[key]),
                                            _lib.ptr(scratch), _lib.stream_of(probs)), "cbl_cumulate_probs")
    return cum_dict
