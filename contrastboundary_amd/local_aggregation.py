"""Host mirror of the TF-side local aggregation operators (value semantics of the TF1 graph code):
    kpconv (PseudoGrid)   /root/reference/tensorflow/models/local_aggregation_operators.py:620-746
    adaptive_weight       ...:316-500 (shipped options, config/s3dis/adapt.yaml:19-26)
    ind_max_pool / ind_closest_pool   /root/reference/tensorflow/models/basic_operators.py:155-192
Arguments keep the reference's names and order (query_points, support_points, neighbors_indices, features, ...); the
trainable variables the TF code creates inside its variable scope (kernel weights, FC weight/bias) are explicit tensors.
The batch-norm / activation / 1x1 convs that follow in the reference are dense layers outside this path (torch)."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_i = ctypes.c_int
_f = ctypes.c_float


def _chk(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous CUDA tensor of {dtype}")
    return t


class _KPConv(Function):
    @staticmethod
    def forward(ctx, query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, influence, closest):
        n, K = neighbors_indices.shape
        n0, C = features.shape
        KP = kernel_points.shape[0]
        out = torch.empty((n, C), dtype=torch.float32, device=features.device)
        from . import pointops
        order = pointops.spatial_order(query_points)             # processing order only: same values (cbl_amd.h)
        _lib.check(_lib.lib().cbl_kpconv_forward_ordered(_i(n), _i(n0), _i(K), _i(C), _i(KP), _lib.ptr(query_points), _lib.ptr(support_points),
                                                         _lib.ptr(neighbors_indices), _lib.ptr(features), _lib.ptr(kernel_points), _lib.ptr(kernel_weights),
                                                         _f(extent), _i(influence), _i(closest), _lib.ptr(order), _lib.ptr(out),
                                                         _lib.stream_of(features)), "cbl_kpconv_forward")
        ctx.save_for_backward(query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights)
        ctx.cfg = (extent, influence, closest)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, s, idx, f, kp, kw = ctx.saved_tensors
        extent, influence, closest = ctx.cfg
        n, K = idx.shape
        n0, C = f.shape
        grad_out = grad_out.contiguous()
        L = _lib.lib()
        from . import pointops
        tr = None
        if C % 4 == 0:
            # gather over the transposed neighbour table (no atomics); built where that pays, taken where it already exists
            tr = pointops.neighbor_transpose(idx, n0, build=(n * K >= pointops.TRANSPOSE_MIN_PAIRS))
        if tr is not None:
            order, inv_start, inv_src = tr
            gf = torch.empty_like(f) if ctx.needs_input_grad[3] else None
            gkw = torch.empty_like(kw) if ctx.needs_input_grad[5] else None
            need = L.cbl_kpconv_backward_csr_workspace_bytes(_i(n0), _i(C), _i(kp.shape[0])) if gkw is not None else 0
            ws = torch.empty(max(need, 1), dtype=torch.uint8, device=f.device)
            rc = L.cbl_kpconv_backward_csr(_i(n), _i(n0), _i(K), _i(C), _i(kp.shape[0]), _lib.ptr(q), _lib.ptr(s), _lib.ptr(f), _lib.ptr(kp), _lib.ptr(kw),
                                           _f(extent), _i(influence), _i(closest), _lib.ptr(grad_out), _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src),
                                           _lib.ptr(gf), _lib.ptr(gkw), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(f))
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cbl_kpconv_backward_csr")
                return None, None, None, gf, None, gkw, None, None, None
        gf = torch.zeros_like(f) if ctx.needs_input_grad[3] else None
        gkw = torch.zeros_like(kw) if ctx.needs_input_grad[5] else None
        _lib.check(L.cbl_kpconv_backward(_i(n), _i(n0), _i(K), _i(C), _i(kp.shape[0]), _lib.ptr(q), _lib.ptr(s), _lib.ptr(idx), _lib.ptr(f),
                                         _lib.ptr(kp), _lib.ptr(kw), _f(extent), _i(influence), _i(closest), _lib.ptr(grad_out),
                                         _lib.ptr(gf), _lib.ptr(gkw), _lib.stream_of(f)), "cbl_kpconv_backward")
        return None, None, None, gf, None, gkw, None, None, None


def kpconv(query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent,
           KP_influence="linear", convolution_mode="sum"):
    """PseudoGrid's kernel-point convolution (depthwise): (n, C) before batch norm / activation.
    extent = KP_extent * radius / density_parameter (local_aggregation_operators.py:664)."""
    _chk(query_points, torch.float32, "query_points"); _chk(support_points, torch.float32, "support_points")
    _chk(neighbors_indices, torch.int32, "neighbors_indices"); _chk(features, torch.float32, "features")
    _chk(kernel_points, torch.float32, "kernel_points"); _chk(kernel_weights, torch.float32, "kernel_weights")
    if KP_influence not in ("linear", "constant"):
        raise NotImplementedError("KP_influence 'gaussian': radius_gaussian is not defined in the reference")
    if convolution_mode not in ("sum", "closest"):
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    return _KPConv.apply(query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, float(extent),
                         1 if KP_influence == "linear" else 0, 1 if convolution_mode == "closest" else 0)


class _AdaptiveWeight(Function):
    @staticmethod
    def forward(ctx, query_points, support_points, neighbors_indices, features, fc_weight, fc_bias, radius, reduction_mean):
        n, K = neighbors_indices.shape
        n0, C = features.shape
        L = _lib.lib()
        pad = torch.empty(1, dtype=torch.int32, device=features.device)
        if reduction_mean:
            _lib.check(L.cbl_index_max(ctypes.c_longlong(n * K), _lib.ptr(neighbors_indices), _lib.ptr(pad), _lib.stream_of(features)), "cbl_index_max")
        out = torch.empty((n, C), dtype=torch.float32, device=features.device)
        from . import pointops
        order = pointops.spatial_order(query_points)             # processing order only: same values (None: the rows as they are)
        _lib.check(L.cbl_adaptive_weight_forward_ordered(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(query_points), _lib.ptr(support_points), _lib.ptr(neighbors_indices),
                                                         _lib.ptr(features), _f(radius), _lib.ptr(fc_weight), _lib.ptr(fc_bias), _lib.ptr(pad), _i(reduction_mean),
                                                         _lib.ptr(order), _lib.ptr(out), _lib.stream_of(features)), "cbl_adaptive_weight_forward")
        ctx.save_for_backward(query_points, support_points, neighbors_indices, features, fc_weight, fc_bias, pad)
        ctx.cfg = (radius, reduction_mean)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, s, idx, f, w, b, pad = ctx.saved_tensors
        radius, reduction_mean = ctx.cfg
        n, K = idx.shape
        n0, C = f.shape
        grad_out = grad_out.contiguous()
        L = _lib.lib()
        need_f, need_w, need_b = ctx.needs_input_grad[3], ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        from . import pointops
        tr = None
        if C % 4 == 0:
            # gather over the transposed neighbour table (no atomics, deterministic); built where that pays, taken where it already exists
            tr = pointops.neighbor_transpose(idx, n0, build=(n * K >= pointops.TRANSPOSE_MIN_PAIRS))
        if tr is not None:
            order, inv_start, inv_src = tr
            gf = torch.empty_like(f) if need_f else None
            gw = torch.empty_like(w) if need_w else None
            gb = torch.empty_like(b) if need_b else None
            ws = torch.empty(L.cbl_adaptive_weight_backward_csr_workspace_bytes(_i(n), _i(n0), _i(C)), dtype=torch.uint8, device=f.device)
            rc = L.cbl_adaptive_weight_backward_csr(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(q), _lib.ptr(s), _lib.ptr(idx), _lib.ptr(f), _f(radius), _lib.ptr(w),
                                                    _lib.ptr(b), _lib.ptr(pad), _i(reduction_mean), _lib.ptr(grad_out), _lib.ptr(order), _lib.ptr(inv_start),
                                                    _lib.ptr(inv_src), _lib.ptr(gf), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                                    _lib.stream_of(f))
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cbl_adaptive_weight_backward_csr")
                return None, None, None, gf, gw, gb, None, None
        gf = torch.zeros_like(f) if need_f else None
        gw = torch.zeros_like(w) if need_w else None
        gb = torch.zeros_like(b) if need_b else None
        _lib.check(L.cbl_adaptive_weight_backward(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(q), _lib.ptr(s), _lib.ptr(idx), _lib.ptr(f), _f(radius),
                                                  _lib.ptr(w), _lib.ptr(b), _lib.ptr(pad), _i(reduction_mean), _lib.ptr(grad_out),
                                                  _lib.ptr(gf), _lib.ptr(gw), _lib.ptr(gb), _lib.stream_of(f)), "cbl_adaptive_weight_backward")
        return None, None, None, gf, gw, gb, None, None


def adaptive_weight(query_points, support_points, neighbors_indices, features, radius, fc_weight, fc_bias, reduction="mean"):
    """AdaptiveWeight aggregation_feature (n, C) before batch norm / activation, shipped options (see module docstring)."""
    _chk(query_points, torch.float32, "query_points"); _chk(support_points, torch.float32, "support_points")
    _chk(neighbors_indices, torch.int32, "neighbors_indices"); _chk(features, torch.float32, "features")
    _chk(fc_weight, torch.float32, "fc_weight"); _chk(fc_bias, torch.float32, "fc_bias")
    if reduction not in ("mean", "avg", "sum"):
        raise NotImplementedError(f"Reduction {reduction} not supported in the fused AdaptiveWeight")
    return _AdaptiveWeight.apply(query_points, support_points, neighbors_indices, features, fc_weight, fc_bias, float(radius),
                                 1 if reduction in ("mean", "avg") else 0)


POSPOOL_EMBEDDINGS = {"one": 0, "xyz": 1, "distance": 2, "exp_-d": 3, "direction_exp_-d": 4, "direction_d": 5, "sin_cos": 6,
                      "two_order": 7, "three_order": 8}
_POSPOOL_REDUCTIONS = {"sum": 0, "mean": 1, "avg": 1, "max": 2}


class _PosPool(Function):
    @staticmethod
    def forward(ctx, query_points, support_points, neighbors_indices, features, radius, pe, red):
        n, K = neighbors_indices.shape
        n0, C = features.shape
        L = _lib.lib()
        pad = torch.empty(1, dtype=torch.int32, device=features.device)
        if red == 1:
            _lib.check(L.cbl_index_max(ctypes.c_longlong(n * K), _lib.ptr(neighbors_indices), _lib.ptr(pad), _lib.stream_of(features)), "cbl_index_max")
        out = torch.empty((n, C), dtype=torch.float32, device=features.device)
        _lib.check(L.cbl_pospool_forward(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(query_points), _lib.ptr(support_points), _lib.ptr(neighbors_indices),
                                         _lib.ptr(features), _f(radius), _i(pe), _i(red), _lib.ptr(pad), _lib.ptr(out), _lib.stream_of(features)),
                   "cbl_pospool_forward")
        ctx.save_for_backward(query_points, support_points, neighbors_indices, features, pad)
        ctx.cfg = (radius, pe, red)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, s, idx, f, pad = ctx.saved_tensors
        radius, pe, red = ctx.cfg
        n, K = idx.shape
        n0, C = f.shape
        grad_out = grad_out.contiguous()
        L = _lib.lib()
        if red != 2 and C % 4 == 0:
            # 'sum' / 'mean': a gather over the transposed neighbour table (no atomics, deterministic), built where that pays, taken where it exists
            from . import pointops
            tr = pointops.neighbor_transpose(idx, n0, build=(n * K >= pointops.TRANSPOSE_MIN_PAIRS))
            if tr is not None:
                order, inv_start, inv_src = tr
                gf = torch.empty_like(f)
                ws = torch.empty(L.cbl_pospool_backward_csr_workspace_bytes(_i(n)), dtype=torch.uint8, device=f.device)
                rc = L.cbl_pospool_backward_csr(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(q), _lib.ptr(s), _lib.ptr(idx), _f(radius), _i(pe), _i(red), _lib.ptr(pad),
                                                _lib.ptr(grad_out), _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(gf), _lib.ptr(ws),
                                                ctypes.c_size_t(ws.numel()), _lib.stream_of(f))
                if rc != _lib.ERR_UNSUPPORTED:
                    _lib.check(rc, "cbl_pospool_backward_csr")
                    return None, None, None, gf, None, None, None
        gf = torch.zeros_like(f)
        _lib.check(_lib.lib().cbl_pospool_backward(_i(n), _i(n0), _i(K), _i(C), _lib.ptr(q), _lib.ptr(s), _lib.ptr(idx), _lib.ptr(f), _f(radius),
                                                   _i(pe), _i(red), _lib.ptr(pad), _lib.ptr(grad_out), _lib.ptr(gf), _lib.stream_of(f)),
                   "cbl_pospool_backward")
        return None, None, None, gf, None, None, None


def pospool(query_points, support_points, neighbors_indices, features, radius, position_embedding="sin_cos", reduction="mean"):
    """PosPool aggregation_feature (n, C) before pool_bn / activation / output_conv
    (tensorflow/models/local_aggregation_operators.py:15-250; options as config.pospool.position_embedding / .reduction)."""
    _chk(query_points, torch.float32, "query_points"); _chk(support_points, torch.float32, "support_points")
    _chk(neighbors_indices, torch.int32, "neighbors_indices"); _chk(features, torch.float32, "features")
    if position_embedding not in POSPOOL_EMBEDDINGS:
        raise NotImplementedError("position_embedding [{}] not supported in PosPool ".format(position_embedding))
    if reduction not in _POSPOOL_REDUCTIONS:
        raise NotImplementedError("Reduction {} not supported in PosPool".format(reduction))
    return _PosPool.apply(query_points, support_points, neighbors_indices, features, float(radius), POSPOOL_EMBEDDINGS[position_embedding],
                          _POSPOOL_REDUCTIONS[reduction])


def ind_max_pool(x, inds):
    """basic_operators.py:155-172"""
    _chk(x, torch.float32, "x"); _chk(inds, torch.int32, "inds")
    n1, d = x.shape
    n2, k = inds.shape
    scratch = torch.empty(d, dtype=torch.int32, device=x.device)
    out = torch.empty((n2, d), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().cbl_ind_max_pool(_i(n1), _i(n2), _i(k), _i(d), _lib.ptr(x), _lib.ptr(inds), _lib.ptr(scratch), _lib.ptr(out), _lib.stream_of(x)),
               "cbl_ind_max_pool")
    return out


def ind_closest_pool(x, inds):
    """basic_operators.py:175-192"""
    _chk(x, torch.float32, "x"); _chk(inds, torch.int32, "inds")
    n1, d = x.shape
    n2, k = inds.shape
    out = torch.empty((n2, d), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().cbl_ind_closest_pool(_i(n1), _i(n2), _i(k), _i(d), _lib.ptr(x), _lib.ptr(inds), _lib.ptr(out), _lib.stream_of(x)),
               "cbl_ind_closest_pool")
    return out
