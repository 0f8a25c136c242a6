"""Dense layers over (n*K) neighbourhood rows with tiny feature widths (the inside of the vector attention,
/root/reference/pytorch/model/blocks.py:23-28,38-40): `linear(x, weight, bias)` == torch.nn.functional.linear, run by the streaming
kernels of csrc/skinny_linear.hip when the shape is one a GEMM library handles badly (many rows, c_in * c_out <= 4096), by torch
(rocBLAS) otherwise.  Parameters stay ordinary nn.Linear tensors."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

MIN_ROWS = 8192          # below this a library GEMM is launch bound either way


def _fits(rows, cin, cout):
    return rows >= MIN_ROWS and cin * cout <= 4096 and cin + cout <= 200


class _SkinnyLinear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        rows, cin = x.shape
        cout = weight.shape[0]
        y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().cbl_skinny_linear_forward(ctypes.c_longlong(rows), ctypes.c_int(cin), ctypes.c_int(cout), _lib.ptr(x), _lib.ptr(weight),
                                                        _lib.ptr(bias), _lib.ptr(y), _lib.stream_of(x)), "cbl_skinny_linear_forward")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        rows, cin = x.shape
        cout = weight.shape[0]
        gy = gy.contiguous()
        L = _lib.lib()
        st = _lib.stream_of(x)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.cbl_skinny_linear_backward_input(ctypes.c_longlong(rows), ctypes.c_int(cin), ctypes.c_int(cout), _lib.ptr(gy), _lib.ptr(weight),
                                                          _lib.ptr(gx), st), "cbl_skinny_linear_backward_input")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            ws = _bn_workspace(L.cbl_skinny_linear_workspace_bytes(ctypes.c_int(cin), ctypes.c_int(cout)), x.device)
            _lib.check(L.cbl_skinny_linear_backward_weight(ctypes.c_longlong(rows), ctypes.c_int(cin), ctypes.c_int(cout), _lib.ptr(x), _lib.ptr(gy),
                                                           _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), st),
                       "cbl_skinny_linear_backward_weight")
        return gx, gw, gb


TALL_ROWS = 16384        # a library GEMM's weight gradient over at least this many rows is cut into chunks (below)


class _TallLinear(Function):
    """F.linear for inputs with MANY rows and widths a GEMM library tiles (the TransitionDown layers' Linear(3 + C, C') over m * nsample grouped rows, the blocks'
    Linear layers of a multi-scene batch): forward and input gradient are the library's; the WEIGHT gradient grad_y^T x — a (c_out x c_in) result contracted
    over all rows — is computed as a batch of S partial products over row chunks and summed.  The library's heuristic answers the unsplit problem with one or
    two workgroups walking the whole contraction (measured: (128 x 327 680) . (327 680 x 67) in 1.95 ms, two such calls and one of 2.2 ms = 15 % of an 8-scene
    training step); the batched form puts S x tiles workgroups on it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        rows, cin = x.shape
        cout = weight.shape[0]
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            S = max(1, min(256, rows // 1024))
            L = rows // S
            main = S * L
            gw = torch.bmm(gy[:main].view(S, L, cout).transpose(1, 2), x[:main].view(S, L, cin)).sum(0)
            if main < rows:
                gw = gw + gy[main:].t() @ x[main:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """x (..., c_in) -> (..., c_out), same values as F.linear up to fp32 summation order"""
    cin, cout = weight.shape[1], weight.shape[0]
    rows = x.numel() // max(cin, 1)
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and _fits(rows, cin, cout)):
        if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and rows >= TALL_ROWS and torch.is_grad_enabled() and weight.requires_grad:
            y = _TallLinear.apply(x.reshape(rows, cin).contiguous(), weight, bias)
            return y.view(*x.shape[:-1], cout)
        return F.linear(x, weight, bias)
    y = _SkinnyLinear.apply(x.reshape(rows, cin).contiguous(), weight.contiguous(), None if bias is None else bias.contiguous())
    return y.view(*x.shape[:-1], cout)


class _TripleLinear(Function):
    """x_q, x_k, x_v = linear_q(x), linear_k(x), linear_v(x)  (blocks.py:33) as one launch per direction (cbl_triple_linear_*, C = 32 | 64)"""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv):
        rows, C = x.shape
        L = _lib.lib()
        ys = [torch.empty((rows, C), dtype=torch.float32, device=x.device) for _ in range(3)]
        arr = lambda ts: (ctypes.c_void_p * 3)(*[0 if t is None else t.data_ptr() for t in ts])
        _lib.check(L.cbl_triple_linear_forward(ctypes.c_longlong(rows), ctypes.c_int(C), _lib.ptr(x), arr([wq, wk, wv]), arr([bq, bk, bv]), arr(ys), _lib.stream_of(x)),
                   "cbl_triple_linear_forward")
        ctx.save_for_backward(x, wq, wk, wv)
        return tuple(ys)

    @staticmethod
    def backward(ctx, gq, gk, gv):
        x, wq, wk, wv = ctx.saved_tensors
        rows, C = x.shape
        L = _lib.lib()
        gys = [g.contiguous() for g in (gq, gk, gv)]
        gx = torch.empty_like(x)
        gws = [torch.empty_like(w) for w in (wq, wk, wv)]
        gbs = [torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(3)]
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        ws = _bn_workspace(L.cbl_triple_linear_workspace_bytes(ctypes.c_int(C)), x.device)
        _lib.check(L.cbl_triple_linear_backward(ctypes.c_longlong(rows), ctypes.c_int(C), _lib.ptr(x), arr([wq, wk, wv]), arr(gys), _lib.ptr(gx), arr(gws), arr(gbs),
                                                _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x)), "cbl_triple_linear_backward")
        return gx, gws[0], gbs[0], gws[1], gbs[1], gws[2], gbs[2]


def triple_linear(x, lq, lk, lv):
    """(lq(x), lk(x), lv(x)) for three nn.Linear(C, C) with biases: one launch per direction at C = 32 | 64 on >= MIN_ROWS rows, `linear` x 3 otherwise"""
    C = x.shape[-1]
    ok = (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and C in (32, 64) and x.shape[0] >= MIN_ROWS
          and all(l.weight.shape == (C, C) and l.bias is not None and l.weight.dtype == torch.float32 for l in (lq, lk, lv)))
    if not ok:
        return apply(lq, x), apply(lk, x), apply(lv, x)
    return _TripleLinear.apply(x.contiguous(), lq.weight.contiguous(), lq.bias.contiguous(), lk.weight.contiguous(), lk.bias.contiguous(),
                               lv.weight.contiguous(), lv.bias.contiguous())


MIN_ROWS_BN = 64            # below this torch's own kernels; up to 4096 rows csrc/bn_rows.hip runs ONE kernel per direction, above it two streaming passes


class _BnRows(Function):
    """y = [relu](bn(x) [+ residual]) (cbl_bn_rows_*_residual); the residual's gradient is the masked incoming gradient"""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum, relu):
        rows, C = x.shape
        L = _lib.lib()
        ws = _bn_workspace(L.cbl_bn_rows_workspace_bytes(ctypes.c_longlong(rows), ctypes.c_int(C)), x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        _lib.check(L.cbl_bn_rows_forward_residual(ctypes.c_longlong(rows), ctypes.c_int(C), _lib.ptr(x), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias),
                                                  ctypes.c_float(eps), ctypes.c_float(momentum), _lib.ptr(running_mean), _lib.ptr(running_var),
                                                  _lib.ptr(num_batches_tracked), ctypes.c_int(relu), _lib.ptr(mean),
                                                  _lib.ptr(invstd), _lib.ptr(y), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x)), "cbl_bn_rows_forward_residual")
        ctx.save_for_backward(x, residual, weight, bias, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, residual, weight, bias, mean, invstd = ctx.saved_tensors
        rows, C = x.shape
        gy = gy.contiguous()
        L = _lib.lib()
        ws = _bn_workspace(L.cbl_bn_rows_workspace_bytes(ctypes.c_longlong(rows), ctypes.c_int(C)), x.device)
        gx = torch.empty_like(x)
        gw = torch.empty(C, dtype=torch.float32, device=x.device) if weight is not None else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if bias is not None else None
        # without a ReLU the skip connection's gradient IS the incoming one: no copy
        want_gr = residual is not None and ctx.needs_input_grad[1]
        gr = torch.empty_like(x) if (want_gr and ctx.relu) else None
        _lib.check(L.cbl_bn_rows_backward_residual(ctypes.c_longlong(rows), ctypes.c_int(C), _lib.ptr(x), _lib.ptr(residual), _lib.ptr(gy), _lib.ptr(weight),
                                                   _lib.ptr(bias), _lib.ptr(mean), _lib.ptr(invstd), ctypes.c_int(ctx.relu), _lib.ptr(gx), _lib.ptr(gr), _lib.ptr(gw),
                                                   _lib.ptr(gb), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x)), "cbl_bn_rows_backward_residual")
        if want_gr and not ctx.relu:
            gr = gy
        return gx, gr, gw, gb, None, None, None, None, None, None


_bn_ws = {}


def _bn_workspace(nbytes, device):
    from .neighbor_state import scratch
    return scratch(_bn_ws, "bn", nbytes, device)


def batch_norm(x, bn, relu=False, residual=None):
    """`relu(bn(x))` / `bn(x)` for an nn.BatchNorm1d over the last dimension of x (..., C), every leading dimension a batch row —
    what the reference writes as bn(x.transpose(1, 2)).transpose(1, 2) for (n, K, C) tensors (blocks.py:38,40).  Train mode with
    running statistics and a float momentum goes through csrc/bn_rows.hip (2 passes forward, 2 backward, ReLU folded in); anything
    else through torch with identical semantics.  residual (same shape as x): `[relu](bn(x) + residual)`, the tail of a residual block
    (blocks.py:130-133) as the same one call."""
    C = x.shape[-1]
    rows = x.numel() // max(C, 1)
    fused = (bn.training and x.is_cuda and x.dtype == torch.float32 and rows >= MIN_ROWS_BN and bn.track_running_stats and bn.momentum is not None
             and (C % 4 == 0 and C <= 1024 or C <= 256)
             and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and residual.device == x.device)))
    if not fused:
        y = bn(x.reshape(-1, C)).view(x.shape)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    y = _BnRows.apply(x.reshape(rows, C).contiguous(), None if residual is None else residual.reshape(rows, C).contiguous(), bn.weight, bn.bias,
                      bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.eps), float(bn.momentum), int(relu))
    return y.view(x.shape)


def apply(layer, x):
    """`layer(x)` for an nn.Linear (through `linear`) or any other module"""
    return linear(x, layer.weight, layer.bias) if isinstance(layer, torch.nn.Linear) else layer(x)


def sequential(seq, x):
    """`seq(x)` for an nn.Sequential of Linear / BatchNorm1d / ReLU acting on (..., C): Linear through `linear`, BatchNorm1d (+ a directly
    following ReLU) through `batch_norm`"""
    layers = list(seq)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, torch.nn.BatchNorm1d):
            fuse = i + 1 < len(layers) and isinstance(layers[i + 1], torch.nn.ReLU)
            x = batch_norm(x, layer, relu=fuse)
            i += 2 if fuse else 1
            continue
        x = apply(layer, x)
        i += 1
    return x
