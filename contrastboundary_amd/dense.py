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
            gw = torch.zeros_like(weight)
            gb = torch.zeros(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            _lib.check(L.cbl_skinny_linear_backward_weight(ctypes.c_longlong(rows), ctypes.c_int(cin), ctypes.c_int(cout), _lib.ptr(x), _lib.ptr(gy),
                                                           _lib.ptr(gw), _lib.ptr(gb), st), "cbl_skinny_linear_backward_weight")
        return gx, gw, gb


def linear(x, weight, bias=None):
    """x (..., c_in) -> (..., c_out), same values as F.linear up to fp32 summation order"""
    cin, cout = weight.shape[1], weight.shape[0]
    rows = x.numel() // max(cin, 1)
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and _fits(rows, cin, cout)):
        return F.linear(x, weight, bias)
    y = _SkinnyLinear.apply(x.reshape(rows, cin).contiguous(), weight.contiguous(), None if bias is None else bias.contiguous())
    return y.view(*x.shape[:-1], cout)


def apply(layer, x):
    """`layer(x)` for an nn.Linear (through `linear`) or any other module"""
    return linear(x, layer.weight, layer.bias) if isinstance(layer, torch.nn.Linear) else layer(x)


def sequential(seq, x):
    """`seq(x)` for an nn.Sequential of Linear / BatchNorm1d / ReLU acting on (rows, C)"""
    for layer in seq:
        x = apply(layer, x)
    return x
