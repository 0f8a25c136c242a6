"""Host mirror of /root/reference/pytorch/model/basic_operators.py (the parts on the hot path):
get_subscene_label :9-14, get_subscene_features :16-50, get_boundary_mask :69-97 — same names and arguments,
bodies are C-ABI launches (contrastboundary_amd/csrc/cbl.hip)."""
import ctypes

import torch

from . import _lib, pointops

_c_int = ctypes.c_int


def _as_int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def get_subscene_label(stage_n, stage_i, stage_list, target, nstride, num_classes, **kwargs):
    """(m, ncls) float32 label distribution of stage `stage_i` points = mean one-hot label of their kr nearest stage-0 points
    (kr = prod(nstride[:i])); one-hot for stage 0.            basic_operators.py:9-14"""
    num_classes = _as_int(num_classes)
    if stage_i == 0 and not kwargs.get("extend", False):
        return torch.nn.functional.one_hot(target, num_classes).float()          # :13, :17-18
    kr = kwargs.get("kr")
    if kr is None:
        i = 1 if stage_i == 0 and kwargs.get("extend", False) else stage_i
        kr = int(torch.prod(torch.as_tensor(nstride)[:i]).item())              # :22
    kr = _as_int(kr)
    stage_from = stage_list["up"][0]
    p_from, o_from = stage_from["p_out"], stage_from["offset"]
    stage_to = stage_list[stage_n][stage_i]
    p_to, o_to = stage_to["p_out"], stage_to["offset"]
    neighbor_idx, _ = pointops.knnquery_raw(kr, p_from, p_to, o_from, o_to, algo="set")     # :30; a mean over the set
    m = p_to.shape[0]
    if target.dtype != torch.int64 or not target.is_cuda or not target.is_contiguous():
        raise TypeError("target must be a contiguous int64 CUDA tensor")
    out = torch.empty((m, num_classes), dtype=torch.float32, device=p_to.device)
    _lib.check(_lib.lib().cbl_subscene_label(_c_int(m), _c_int(kr), _c_int(num_classes), _lib.ptr(target), _lib.ptr(neighbor_idx),
                                             _lib.ptr(out), _lib.stream_of(p_to)), "cbl_subscene_label")
    if kwargs.get("return_neighbor", False):
        return out, neighbor_idx.view(-1).long(), kr
    return out


def get_subscene_features(stage_n, stage_i, stage_list, x, nstride, kr=None, extend=False, return_neighbor=False):
    """(m, c) float32 = mean over the kr nearest stage-0 points of their rows of `x` (n, c), any per-point features; `x.float()` itself for
    stage 0.                                                            basic_operators.py:16-50 (get_subscene_label is its one-hot case)"""
    if stage_i == 0 and not extend:
        return x.float()                                                                 # :17-18
    if kr is None:
        i = 1 if stage_i == 0 and extend else stage_i
        kr = int(torch.prod(torch.as_tensor(nstride)[:i]).item())                      # :22
    kr = _as_int(kr)
    stage_from = stage_list["up"][0]
    p_from, o_from = stage_from["p_out"], stage_from["offset"]
    stage_to = stage_list[stage_n][stage_i]
    p_to, o_to = stage_to["p_out"], stage_to["offset"]
    neighbor_idx, _ = pointops.knnquery_raw(kr, p_from, p_to, o_from, o_to, algo="set")     # :30; a mean over the set
    weight = torch.full((p_to.shape[0], kr), 1.0 / kr, dtype=torch.float32, device=p_to.device)
    out = pointops.WeightedGather.apply(x.float().contiguous(), neighbor_idx, weight)      # :44-45 without the (m, kr, c) gather
    if return_neighbor:
        return out, neighbor_idx.view(-1).long(), kr
    return out


def get_boundary_mask(labels, neighbor_label=None, neighbor_idx=None, valid_mask=None, get_plain=False, get_cnt=False):
    """basic_operators.py:69-97.  `neighbor_idx` (n,k) int32 is the native path; a precomputed `neighbor_label` is accepted for
    signature parity and handled with the same comparisons in torch."""
    if neighbor_label is not None:
        valid_nb = neighbor_label >= 0
        lab = labels.unsqueeze(-1)
        neq = (lab != neighbor_label) & valid_nb
        bound = neq.sum(-1) if get_cnt else neq.any(-1)
        plain = ((lab == neighbor_label) | ~valid_nb).all(-1)
    else:
        n, k = neighbor_idx.shape
        if labels.dtype != torch.int64:
            raise TypeError("labels must be int64")
        b8 = torch.empty(n, dtype=torch.uint8, device=labels.device)
        p8 = torch.empty(n, dtype=torch.uint8, device=labels.device)
        cnt = torch.empty(n, dtype=torch.int32, device=labels.device)
        labels_c, nidx_c = labels.contiguous(), neighbor_idx.contiguous()      # named: both must outlive the call (see tf_ops.py)
        _lib.check(_lib.lib().cbl_boundary_mask(_c_int(n), _c_int(k), _lib.ptr(labels_c), _lib.ptr(nidx_c),
                                                _lib.ptr(b8), _lib.ptr(p8), _lib.ptr(cnt), _lib.stream_of(labels)), "cbl_boundary_mask")
        bound = cnt.long() if get_cnt else b8.bool()
        plain = p8.bool()
    if valid_mask is not None:
        bound = bound * valid_mask if get_cnt else torch.logical_and(bound, valid_mask)
        plain = torch.logical_and(plain, valid_mask)
    return (bound, plain) if get_plain else bound


def boundary_iou(pred, labels, neighbor_idx=None, xyz=None, offset=None, kr=None, num_classes=13, ignore_label=255):
    """Boundary / inner-area IoU statistics of one room (tool/test.py:250-257 + :392-417), on the GPU: the kr-neighbourhood search on
    the full-resolution cloud, the boundary / plain masks and the masked intersection-and-union histograms.
    pred, labels (n,) int64; either neighbor_idx (n,kr) int32 or (xyz, offset, kr).
    -> {'bound': (i, u, t), 'plain': (i, u, t)} int64 tensors of length num_classes, exactly intersectionAndUnion's triplets"""
    if neighbor_idx is None:
        neighbor_idx, _ = pointops.knnquery_raw(kr, xyz, xyz, offset, offset, algo="set")      # the masks only use the neighbour set
    n, k = neighbor_idx.shape
    if pred.dtype != torch.int64 or labels.dtype != torch.int64:
        raise TypeError("pred and labels must be int64")
    hist = torch.zeros((2, 3, int(num_classes)), dtype=torch.int64, device=labels.device)
    pred_c, labels_c, nidx_c = pred.contiguous(), labels.contiguous(), neighbor_idx.contiguous()
    _lib.check(_lib.lib().cbl_boundary_iou(_c_int(n), _c_int(k), _c_int(int(num_classes)), ctypes.c_longlong(int(ignore_label)), _lib.ptr(pred_c),
                                           _lib.ptr(labels_c), _lib.ptr(nidx_c), _lib.ptr(hist), _lib.stream_of(labels)), "cbl_boundary_iou")
    out = {}
    for mi, name in enumerate(("bound", "plain")):
        i, o, t = hist[mi, 0], hist[mi, 1], hist[mi, 2]
        out[name] = (i, o + t - i, t)
    return out
