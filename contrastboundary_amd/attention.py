"""Fused C-wide part of PointTransformerLayer's vector attention (csrc/attention.hip, /root/reference/pytorch/model/blocks.py:31-44):
`attn_w2` and `attn_agg` as autograd Functions over the layer's own parameter tensors.  Available for the two full-resolution
stages (C = 32 / 64 with share_planes = 8, K <= 64) in training mode (the BatchNorm inside is the train-mode one; evaluation takes the
separate kernels); `supported()` says when."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_ws = {}


def _workspace(nbytes, device):
    from .neighbor_state import scratch
    return scratch(_ws, "attn", nbytes, device)


def supported(layer, x):
    C = layer.out_planes
    return (layer.training and x.is_cuda and x.dtype == torch.float32 and layer.mid_planes == C and layer.share_planes == 8 and C in (32, 64, 128, 256, 512)
            and layer.nsample <= 64 and (C > 64 or x.shape[0] * layer.nsample >= 16384)
            and isinstance(layer.linear_w[0], torch.nn.BatchNorm1d) and layer.linear_w[0].track_running_stats and layer.linear_w[0].momentum is not None)


_i, _f = ctypes.c_int, ctypes.c_float


def _table(idx, n, C):
    """transposed neighbour table of a self-search's idx (n targets) for the gather-form backward passes of the two full-resolution widths; built
    where that pays (the layers of a stage share one idx: one table per stage and step), taken where it exists, None otherwise (atomics)"""
    if C > 64:
        return None
    from . import pointops
    return pointops.neighbor_transpose(idx, n, build=(idx.numel() >= pointops.TRANSPOSE_MIN_PAIRS))


class AttnW2(Function):
    @staticmethod
    def forward(ctx, x_q, x_k, p1, W3C, b3C, bn_w, bn_b, Wa, ba, idx, bn, training):
        n, C = x_q.shape
        K, G = idx.shape[1], Wa.shape[0]
        L = _lib.lib()
        dev = x_q.device
        ws = _workspace(L.cbl_attn_workspace_bytes(_i(C), _i(G)), dev)
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev); invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean = bn.running_mean.clone(); invstd = torch.rsqrt(bn.running_var + bn.eps)
        w2 = torch.empty((n, K, G), dtype=torch.float32, device=dev)
        _lib.check(L.cbl_attn_w2_forward(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_q), _lib.ptr(x_k), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C),
                                         _lib.ptr(bn_w), _lib.ptr(bn_b), _f(bn.eps), _f(bn.momentum if bn.momentum is not None else 0.1),
                                         _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(bn.num_batches_tracked), _i(1 if training else 0),
                                         _lib.ptr(Wa), _lib.ptr(ba), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(w2), _lib.ptr(ws),
                                         ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)), "cbl_attn_w2_forward")
        ctx.save_for_backward(x_q, x_k, p1, W3C, b3C, bn_w, bn_b, Wa, idx, mean, invstd)
        ctx.training = training
        return w2

    @staticmethod
    def backward(ctx, g_w2):
        x_q, x_k, p1, W3C, b3C, bn_w, bn_b, Wa, idx, mean, invstd = ctx.saved_tensors
        if not ctx.training:
            raise NotImplementedError("attn_w2 backward is the train-mode BatchNorm backward")
        n, C = x_q.shape
        K, G = idx.shape[1], Wa.shape[0]
        L = _lib.lib()
        dev = x_q.device
        ws = _workspace(L.cbl_attn_workspace_bytes(_i(C), _i(G)), dev)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        g_W3C, g_b3C, g_bw, g_bb, g_Wa, g_ba = e(C, 3), e(C), e(C), e(C), e(G, C), e(G)
        g_w2 = g_w2.contiguous()
        tr = _table(idx, n, C)
        if tr is not None:                                          # the x_k scatter as a gather over the transposed table: no atomics
            order, inv_start, inv_src = tr
            g_xq, g_xk, g_p1 = e(n, C), e(n, C), e(n, K, 3)
            _lib.check(L.cbl_attn_w2_backward_csr(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_q), _lib.ptr(x_k), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C),
                                                  _lib.ptr(bn_w), _lib.ptr(bn_b), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(Wa), _lib.ptr(g_w2),
                                                  _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src),
                                                  _lib.ptr(g_xq), _lib.ptr(g_xk), _lib.ptr(g_p1), _lib.ptr(g_W3C), _lib.ptr(g_b3C), _lib.ptr(g_bw), _lib.ptr(g_bb),
                                                  _lib.ptr(g_Wa), _lib.ptr(g_ba), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)),
                       "cbl_attn_w2_backward_csr")
            return g_xq, g_xk, g_p1, g_W3C, g_b3C, g_bw, g_bb, g_Wa, g_ba, None, None, None
        g_xq, g_xk, g_p1 = e(n, C), torch.zeros(n, C, dtype=torch.float32, device=dev), e(n, K, 3)
        _lib.check(L.cbl_attn_w2_backward(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_q), _lib.ptr(x_k), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C),
                                          _lib.ptr(bn_w), _lib.ptr(bn_b), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(Wa), _lib.ptr(g_w2),
                                          _lib.ptr(g_xq), _lib.ptr(g_xk), _lib.ptr(g_p1), _lib.ptr(g_W3C), _lib.ptr(g_b3C), _lib.ptr(g_bw), _lib.ptr(g_bb),
                                          _lib.ptr(g_Wa), _lib.ptr(g_ba), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)),
                   "cbl_attn_w2_backward")
        return g_xq, g_xk, g_p1, g_W3C, g_b3C, g_bw, g_bb, g_Wa, g_ba, None, None, None


class AttnAgg(Function):
    """out[i] = sum_k (x_v[idx[i,k]] + p_r[i,k]) * a[i,k]   (blocks.py:42-43) with p_r = Linear(3,C)(p1) formed on the fly.
    softmax = True: `a` are the LOGITS (n,K,G) and the softmax over K (blocks.py:41) runs inside the kernels — forward writes the weights for
    the backward pass, backward returns the gradient of the logits: no softmax launches of torch's, forward or backward."""

    @staticmethod
    def forward(ctx, x_v, p1, W3C, b3C, a, idx, softmax=False):
        n, C = x_v.shape
        K, G = idx.shape[1], a.shape[2]
        out = torch.empty((n, C), dtype=torch.float32, device=x_v.device)
        L = _lib.lib()
        if softmax:
            weights = torch.empty_like(a)
            _lib.check(L.cbl_attn_agg_softmax_forward(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_v), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C),
                                                      _lib.ptr(a), _lib.ptr(weights), _lib.ptr(out), _lib.stream_of(x_v)), "cbl_attn_agg_softmax_forward")
            a = weights
        else:
            _lib.check(L.cbl_attn_agg_forward(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_v), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C),
                                              _lib.ptr(a), _lib.ptr(out), _lib.stream_of(x_v)), "cbl_attn_agg_forward")
        ctx.save_for_backward(x_v, p1, W3C, b3C, a, idx)
        ctx.softmax = bool(softmax)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x_v, p1, W3C, b3C, a, idx = ctx.saved_tensors
        n, C = x_v.shape
        K, G = idx.shape[1], a.shape[2]
        L = _lib.lib()
        dev = x_v.device
        ws = _workspace(L.cbl_attn_workspace_bytes(_i(C), _i(G)), dev)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        g_out = g_out.contiguous()
        tr = _table(idx, n, C)
        if tr is not None:                                          # the x_v scatter as a gather over the transposed table: no atomics
            order, inv_start, inv_src = tr
            g_xv, g_p1, g_W3C, g_b3C, g_a = e(n, C), e(n, K, 3), e(C, 3), e(C), e(n, K, G)
            _lib.check(L.cbl_attn_agg_backward_csr(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_v), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C), _lib.ptr(a),
                                                   _lib.ptr(g_out), _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(g_xv), _lib.ptr(g_p1),
                                                   _lib.ptr(g_W3C), _lib.ptr(g_b3C), _lib.ptr(g_a), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                                   _i(1 if ctx.softmax else 0), _lib.stream_of(x_v)), "cbl_attn_agg_backward_csr")
            return g_xv, g_p1, g_W3C, g_b3C, g_a, None, None
        g_xv, g_p1, g_W3C, g_b3C, g_a = torch.zeros(n, C, dtype=torch.float32, device=dev), e(n, K, 3), e(C, 3), e(C), e(n, K, G)
        fn, what = (L.cbl_attn_agg_softmax_backward, "cbl_attn_agg_softmax_backward") if ctx.softmax else (L.cbl_attn_agg_backward, "cbl_attn_agg_backward")
        _lib.check(fn(_i(n), _i(K), _i(C), _i(G), _lib.ptr(x_v), _lib.ptr(idx), _lib.ptr(p1), _lib.ptr(W3C), _lib.ptr(b3C), _lib.ptr(a),
                      _lib.ptr(g_out), _lib.ptr(g_xv), _lib.ptr(g_p1), _lib.ptr(g_W3C), _lib.ptr(g_b3C), _lib.ptr(g_a), _lib.ptr(ws),
                      ctypes.c_size_t(ws.numel()), _lib.stream_of(x_v)), what)
        return g_xv, g_p1, g_W3C, g_b3C, g_a, None, None
