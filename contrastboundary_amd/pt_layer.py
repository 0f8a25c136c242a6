"""PointTransformerLayer's vector attention as one pass structure (csrc/pt_layer.hip, /root/reference/pytorch/model/blocks.py:34-44):
everything of the layer behind its q / k / v Linear layers — linear_p, the BatchNorm / Linear stack linear_w, the softmax over K and the
aggregation — as ONE autograd Function over the layer's own parameter tensors, for the two full-resolution shapes (C = 32 | 64 with
share_planes = 8, K = 8 | 16).  Training mode: five forward and six backward passes over the (point, neighbour) pairs, nothing of shape
(n, K, C) stored, no atomics (the x_k / x_v gradients are gathers over the transposed neighbour table), run-to-run deterministic.  Evaluation mode
under torch.no_grad(): cbl_pt_layer_forward_eval (running statistics, no statistics passes).  `supported()` says when; other shapes, and evaluation
with gradients enabled, take attention.py's kernels."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_ws = {}
MAX_POINTS = 1 << 20                                   # the transposed table's limit (neighbor_transpose.hip NT_MAX_TILES)


def _workspace(nbytes, device):
    from .neighbor_state import scratch
    return scratch(_ws, "pt_layer", nbytes, device)


def _bn_ok(bn):
    return (isinstance(bn, torch.nn.BatchNorm1d) and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32)


def _stats_ok(bn):
    """the raw running-statistics pointers the C entry takes: contiguous fp32 (int64 batch counter)"""
    return (bn.running_mean is not None and bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous()
            and bn.running_var.dtype == torch.float32 and bn.running_var.is_contiguous()
            and bn.num_batches_tracked is not None and bn.num_batches_tracked.dtype == torch.int64)


def supported(layer, x, idx=None, p=None):
    """training mode (forward + backward), or evaluation mode under torch.no_grad() (the reference's test loop, tool/test.py:217-240: running statistics,
    no backward pass); evaluation WITH gradients takes the other paths.  `idx` / `p`: the neighbour table and coordinates the caller is about to hand over —
    the C entry reads them through raw pointers (K = idx.shape[1]), so a table that is not exactly (n, layer.nsample) int32 contiguous on x's device, or
    coordinates that are not (n, 3) fp32, send the layer down attention.py's path (which validates through pointops._req) instead of into the kernels."""
    C = layer.out_planes
    mode_ok = layer.training or not torch.is_grad_enabled()
    ok = (mode_ok and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and layer.mid_planes == C and layer.share_planes == 8 and C in (32, 64)
          and int(layer.nsample) in (8, 16) and 16 <= x.shape[0] <= MAX_POINTS
          and _bn_ok(layer.linear_p[1]) and _bn_ok(layer.linear_w[0]) and _bn_ok(layer.linear_w[3])
          and _stats_ok(layer.linear_p[1]) and _stats_ok(layer.linear_w[0]) and _stats_ok(layer.linear_w[3]))
    if ok and idx is not None:
        ok = (idx.dtype == torch.int32 and idx.device == x.device and idx.dim() == 2 and idx.is_contiguous()
              and tuple(idx.shape) == (x.shape[0], int(layer.nsample)))
    if ok and p is not None:
        ok = p.dtype == torch.float32 and p.device == x.device and tuple(p.shape) == (x.shape[0], 3)
    return bool(ok)


_i, _f = ctypes.c_int, ctypes.c_float
_P = _lib.ptr


class PTAttention(Function):
    """out (n, C) = vector attention of blocks.py:34-44 given x_q / x_k / x_v (n, C), the coordinates and the layer's neighbour table.
    Parameter order: linear_p[0].weight/.bias, linear_p[1].weight/.bias, linear_p[3].weight/.bias, linear_w[0].weight/.bias,
    linear_w[2].weight/.bias, linear_w[3].weight/.bias, linear_w[5].weight/.bias."""

    @staticmethod
    def forward(ctx, p, x_q, x_k, x_v, idx, bns, *params):
        from . import pointops
        n, C = x_q.shape
        K, G = idx.shape[1], C // 8
        L = _lib.lib()
        dev = x_q.device
        x_q, x_k, x_v, p = x_q.contiguous(), x_k.contiguous(), x_v.contiguous(), p.contiguous()
        params = [t.contiguous() for t in params]
        order = pointops.spatial_order(idx)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        p_r, p0, p1, w2, a, out = e(n, K, 3), e(n, K, 3), e(n, K, 3), e(n, K, G), e(n, K, G), e(n, C)
        consts = e(L.cbl_pt_layer_consts_floats())
        ws = _workspace(L.cbl_pt_layer_workspace_bytes(_i(n), _i(K), _i(C)), dev)
        eps3 = (_f * 3)(*[float(b.eps) for b in bns])
        mom3 = (_f * 3)(*[float(b.momentum) for b in bns])
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        _lib.check(L.cbl_pt_layer_forward(_i(n), _i(K), _i(C), _P(p), _P(x_q), _P(x_k), _P(x_v), _P(idx), _P(order), *[_P(t) for t in params], eps3, mom3,
                                          arr([b.running_mean for b in bns]), arr([b.running_var for b in bns]), arr([b.num_batches_tracked for b in bns]),
                                          _P(p_r), _P(p0), _P(p1), _P(w2), _P(a), _P(out), _P(consts), _P(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)),
                   "cbl_pt_layer_forward")
        ctx.save_for_backward(x_q, x_k, x_v, idx, p_r, p0, p1, w2, a, consts, *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import pointops
        x_q, x_k, x_v, idx, p_r, p0, p1, w2, a, consts = ctx.saved_tensors[:10]
        Wp, bp, gamma_p, beta_p, W3C, b3C, gamma_c, beta_c, Wa, ba, gamma_g, beta_g, Wb, bb = params = ctx.saved_tensors[10:]
        n, C = x_q.shape
        K = idx.shape[1]
        L = _lib.lib()
        dev = x_q.device
        tr = pointops.neighbor_transpose(idx, n, build=True)
        if tr is None:
            raise _lib.CblError("pt_layer backward needs the transposed neighbour table (n <= %d)" % MAX_POINTS)
        order, inv_start, inv_src = tr
        g_out = g_out.contiguous()
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        g_xq, g_xk, g_xv = e(n, C), e(n, C), e(n, C)
        g_params = [torch.empty_like(t) for t in params]
        ws = _workspace(L.cbl_pt_layer_workspace_bytes(_i(n), _i(K), _i(C)), dev)
        _lib.check(L.cbl_pt_layer_backward(_i(n), _i(K), _i(C), _P(x_q), _P(x_k), _P(x_v), _P(idx), _P(order), _P(inv_start), _P(inv_src), _P(gamma_p), _P(W3C),
                                           _P(b3C), _P(gamma_c), _P(Wa), _P(gamma_g), _P(Wb), _P(p_r), _P(p0), _P(p1), _P(w2), _P(a), _P(consts), _P(g_out),
                                           _P(g_xq), _P(g_xk), _P(g_xv), *[_P(t) for t in g_params], _P(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)),
                   "cbl_pt_layer_backward")
        return (None, g_xq, g_xk, g_xv, None, None, *g_params)


def supported_wide(layer, x, idx=None, p=None):
    """the wide stages (C = 128 | 256 | 512, share_planes 8, K <= 64) in TRAINING mode: the whole layer behind its projections as one call each way
    (cbl_pt_layer_wide_*: the attention.hip kernels for the C-wide passes, pt_layer.hip's for the narrow ones)"""
    C = layer.out_planes
    ok = (layer.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and layer.mid_planes == C and layer.share_planes == 8 and C in (128, 256, 512)
          and 1 <= int(layer.nsample) <= 64 and 16 <= x.shape[0] <= MAX_POINTS
          and _bn_ok(layer.linear_p[1]) and _bn_ok(layer.linear_w[0]) and _bn_ok(layer.linear_w[3])
          and _stats_ok(layer.linear_p[1]) and _stats_ok(layer.linear_w[0]) and _stats_ok(layer.linear_w[3]))
    if ok and idx is not None:
        ok = (idx.dtype == torch.int32 and idx.device == x.device and idx.dim() == 2 and idx.is_contiguous()
              and tuple(idx.shape) == (x.shape[0], int(layer.nsample)))
    if ok and p is not None:
        ok = p.dtype == torch.float32 and p.device == x.device and tuple(p.shape) == (x.shape[0], 3)
    return bool(ok)


def _wide_forward(p, x_q, x_k, x_v, idx, bns, params):
    """cbl_pt_layer_wide_forward on contiguous projections; returns (out, tensors a backward pass needs)"""
    n, C = x_q.shape
    K, G = idx.shape[1], C // 8
    L = _lib.lib()
    dev = x_q.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    p_r, p0, p1, w2, a, out = e(n, K, 3), e(n, K, 3), e(n, K, 3), e(n, K, G), e(n, K, G), e(n, C)
    consts, bnc = e(L.cbl_pt_layer_wide_consts_floats()), e(2 * C)
    ws = _workspace(L.cbl_pt_layer_wide_workspace_bytes(_i(n), _i(K), _i(C)), dev)
    eps3 = (_f * 3)(*[float(b.eps) for b in bns])
    mom3 = (_f * 3)(*[float(b.momentum) for b in bns])
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    _lib.check(L.cbl_pt_layer_wide_forward(_i(n), _i(K), _i(C), _P(p), _P(x_q), _P(x_k), _P(x_v), _P(idx), *[_P(t) for t in params], eps3, mom3,
                                           arr([b.running_mean for b in bns]), arr([b.running_var for b in bns]), arr([b.num_batches_tracked for b in bns]),
                                           _P(p_r), _P(p0), _P(p1), _P(w2), _P(a), _P(out), _P(consts), _P(bnc), _P(ws), ctypes.c_size_t(ws.numel()),
                                           _lib.stream_of(x_q)), "cbl_pt_layer_wide_forward")
    return out, (p_r, p0, p1, w2, a, consts, bnc)


def _wide_backward(x_q, x_k, x_v, idx, kept, params, g_out, g_qkv):
    """cbl_pt_layer_wide_backward; g_qkv (3, n, C) receives d x_q / d x_k / d x_v (the last two adjacent: the call zeroes both scatter targets with one fill)"""
    p_r, p0, p1, w2, a, consts, bnc = kept
    Wp, bp, gamma_p, beta_p, W3C, b3C, gamma_c, beta_c, Wa, ba, gamma_g, beta_g, Wb, bb = params
    n, C = x_q.shape
    K = idx.shape[1]
    L = _lib.lib()
    g_params = [torch.empty_like(t) for t in params]
    ws = _workspace(L.cbl_pt_layer_wide_workspace_bytes(_i(n), _i(K), _i(C)), x_q.device)
    _lib.check(L.cbl_pt_layer_wide_backward(_i(n), _i(K), _i(C), _P(x_q), _P(x_k), _P(x_v), _P(idx), _P(gamma_p), _P(W3C), _P(b3C), _P(gamma_c), _P(beta_c),
                                            _P(Wa), _P(gamma_g), _P(Wb), _P(p_r), _P(p0), _P(p1), _P(w2), _P(a), _P(consts), _P(bnc), _P(g_out.contiguous()),
                                            _P(g_qkv[0]), _P(g_qkv[1]), _P(g_qkv[2]), *[_P(t) for t in g_params], _P(ws), ctypes.c_size_t(ws.numel()),
                                            _lib.stream_of(x_q)), "cbl_pt_layer_wide_backward")
    return g_params


class PTAttentionWide(Function):
    """PTAttention for the wide stages (cbl_pt_layer_wide_forward / _backward); same argument and parameter order"""

    @staticmethod
    def forward(ctx, p, x_q, x_k, x_v, idx, bns, *params):
        x_q, x_k, x_v, p = x_q.contiguous(), x_k.contiguous(), x_v.contiguous(), p.contiguous()
        params = [t.contiguous() for t in params]
        out, kept = _wide_forward(p, x_q, x_k, x_v, idx, bns, params)
        ctx.save_for_backward(x_q, x_k, x_v, idx, *kept, *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x_q, x_k, x_v, idx = ctx.saved_tensors[:4]
        g_qkv = torch.empty((3,) + tuple(x_q.shape), dtype=torch.float32, device=x_q.device)
        g_params = _wide_backward(x_q, x_k, x_v, idx, ctx.saved_tensors[4:11], ctx.saved_tensors[11:], g_out, g_qkv)
        return (None, g_qkv[0], g_qkv[1], g_qkv[2], None, None, *g_params)


def _adjacent(ts):
    """do the three tensors already lie one after the other in one storage?"""
    a, b, c = ts
    step = a.numel()
    return (a.is_contiguous() and b.is_contiguous() and c.is_contiguous() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + step and c.storage_offset() == a.storage_offset() + 2 * step)


def _stacked(ts):
    """the three tensors as one (3, ...) tensor: a view when they already lie one after the other in memory (`adjoin_qkv`), a copy otherwise"""
    a, b, c = ts
    if _adjacent(ts):
        return a.as_strided((3,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))
    return torch.stack((a, b, c))


def adjoin_qkv(module):
    """lay the q / k / v projections' weights (and biases) of every PointTransformerLayer under `module` out one after the other, so that `_stacked` is a view.
    Values, Parameter objects and state_dict entries stay what they were; call it after the module reached its device and before an optimizer state / flat state
    takes views of the parameters (distributed.FlatState keeps such groups adjacent itself).  Idempotent: a triple that is adjacent already keeps its storage —
    a captured hipGraph (an earlier GraphedTrainStep on the same model) holds the addresses of that storage, and moving the weights away would leave it reading
    and updating memory that went back to the allocator."""
    from .blocks import PointTransformerLayer
    for m in module.modules():
        if isinstance(m, PointTransformerLayer) and m.linear_q.weight.shape == m.linear_k.weight.shape == m.linear_v.weight.shape:
            for name in ("weight", "bias"):
                ps = [getattr(l, name) for l in (m.linear_q, m.linear_k, m.linear_v)]
                if any(t is None for t in ps) or _adjacent([t.data for t in ps]):
                    continue
                with torch.no_grad():
                    whole = torch.stack([t.data for t in ps])
                    for i, t in enumerate(ps):
                        t.data = whole[i]
    return module


class PTAttentionWideProjected(Function):
    """the q / k / v projections (blocks.py:33) and PTAttentionWide as one node: the three Linear(C, C) as ONE batched product each way (forward 1 launch
    instead of 3; backward d x, d W, d b in 4 instead of 11), their outputs and gradients as slices of one (3, n, C) tensor"""

    @staticmethod
    def forward(ctx, p, x, idx, bns, wq, bq, wk, bk, wv, bv, *params):
        x, p = x.contiguous(), p.contiguous()
        params = [t.contiguous() for t in params]
        W3, b3 = _stacked((wq, wk, wv)), _stacked((bq, bk, bv))
        n, C = x.shape
        qkv = torch.baddbmm(b3.unsqueeze(1), x.unsqueeze(0).expand(3, n, C), W3.transpose(1, 2))        # (3, n, C): x W^T + b per projection
        out, kept = _wide_forward(p, qkv[0], qkv[1], qkv[2], idx, bns, params)
        ctx.save_for_backward(x, W3, qkv, idx, *kept, *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, W3, qkv, idx = ctx.saved_tensors[:4]
        n, C = x.shape
        g_qkv = torch.empty_like(qkv)
        g_params = _wide_backward(qkv[0], qkv[1], qkv[2], idx, ctx.saved_tensors[4:11], ctx.saved_tensors[11:], g_out, g_qkv)
        g_x = torch.bmm(g_qkv, W3).sum(0) if ctx.needs_input_grad[1] else None
        g_W3 = torch.bmm(g_qkv.transpose(1, 2), x.unsqueeze(0).expand(3, n, C))
        g_b3 = g_qkv.sum(1)
        return (None, g_x, None, None, g_W3[0], g_b3[0], g_W3[1], g_b3[1], g_W3[2], g_b3[2], *g_params)


def attention_wide_projected(layer, p, x, idx):
    """a wide-stage `layer` on its input features, projections included (training mode)"""
    lp, lw = layer.linear_p, layer.linear_w
    q, k, v = layer.linear_q, layer.linear_k, layer.linear_v
    return PTAttentionWideProjected.apply(p, x, idx, (lp[1], lw[0], lw[3]), q.weight, q.bias, k.weight, k.bias, v.weight, v.bias,
                                          lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias, lp[3].weight, lp[3].bias,
                                          lw[0].weight, lw[0].bias, lw[2].weight, lw[2].bias, lw[3].weight, lw[3].bias, lw[5].weight, lw[5].bias)


def attention_wide(layer, p, x_q, x_k, x_v, idx):
    """the fused part of a wide-stage `layer` on its q / k / v projections (training mode)"""
    lp, lw = layer.linear_p, layer.linear_w
    return PTAttentionWide.apply(p, x_q, x_k, x_v, idx, (lp[1], lw[0], lw[3]), lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias, lp[3].weight, lp[3].bias,
                                 lw[0].weight, lw[0].bias, lw[2].weight, lw[2].bias, lw[3].weight, lw[3].bias, lw[5].weight, lw[5].bias)


def _forward_eval(p, x_q, x_k, x_v, idx, bns, params):
    """evaluation mode: cbl_pt_layer_forward_eval (running statistics, nothing kept for a backward pass)"""
    from . import pointops
    n, C = x_q.shape
    K, G = idx.shape[1], C // 8
    L = _lib.lib()
    dev = x_q.device
    x_q, x_k, x_v, p = x_q.contiguous(), x_k.contiguous(), x_v.contiguous(), p.contiguous()
    params = [t.contiguous() for t in params]
    order = pointops.spatial_order(idx)
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    p_r, p0, p1, w2, a, out = e(n, K, 3), e(n, K, 3), e(n, K, 3), e(n, K, G), e(n, K, G), e(n, C)
    consts = e(L.cbl_pt_layer_consts_floats())
    ws = _workspace(L.cbl_pt_layer_workspace_bytes(_i(n), _i(K), _i(C)), dev)
    eps3 = (_f * 3)(*[float(b.eps) for b in bns])
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    _lib.check(L.cbl_pt_layer_forward_eval(_i(n), _i(K), _i(C), _P(p), _P(x_q), _P(x_k), _P(x_v), _P(idx), _P(order), *[_P(t) for t in params], eps3,
                                           arr([b.running_mean for b in bns]), arr([b.running_var for b in bns]),
                                           _P(p_r), _P(p0), _P(p1), _P(w2), _P(a), _P(out), _P(consts), _P(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(x_q)),
               "cbl_pt_layer_forward_eval")
    return out


def attention(layer, p, x_q, x_k, x_v, idx):
    """the fused part of `layer` (a blocks.PointTransformerLayer) on its q / k / v projections"""
    lp, lw = layer.linear_p, layer.linear_w
    bns = (lp[1], lw[0], lw[3])
    if not layer.training:
        return _forward_eval(p, x_q, x_k, x_v, idx, bns, [lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias, lp[3].weight, lp[3].bias, lw[0].weight, lw[0].bias,
                                                          lw[2].weight, lw[2].bias, lw[3].weight, lw[3].bias, lw[5].weight, lw[5].bias])
    return PTAttention.apply(p, x_q, x_k, x_v, idx, bns, lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias, lp[3].weight, lp[3].bias,
                             lw[0].weight, lw[0].bias, lw[2].weight, lw[2].bias, lw[3].weight, lw[3].bias, lw[5].weight, lw[5].bias)
